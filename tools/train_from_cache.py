#!/usr/bin/env python
"""End-to-end example of the accelerated path: embedding cache on disk (reference layout) -> PrefetchLoader -> fused LoRA
training step (forward, criterion, backward, bucketed all-reduce, clip+AdamW) -> LR schedule -> checkpoint.

  python tools/train_from_cache.py --cache /path/to/cache --transformer /path/to/Qwen-Image-Edit/transformer --steps 100
  python tools/train_from_cache.py --synthetic 16 --layers 2 --steps 6          # self-contained smoke run (random weights)
  python -m torch.distributed.run --nproc-per-node 8 tools/train_from_cache.py ...   # one process per GPU (RCCL)
"""
import argparse, glob, json, os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.data import CachedEmbeddingDataset, PrefetchLoader, convert_img_shapes_to_latent_space, write_cache_sample
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep, get_scheduler
from qflux_amd.trainer.qwen_step import init_distributed_from_env

ap = argparse.ArgumentParser()
ap.add_argument("--cache"); ap.add_argument("--transformer"); ap.add_argument("--synthetic", type=int, default=0)
ap.add_argument("--layers", type=int, default=60); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--rank", type=int, default=16); ap.add_argument("--lr", type=float, default=1e-4); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--out", default=None)
a = ap.parse_args()
rank, local, world = init_distributed_from_env()
dev = torch.device("cuda", local)
torch.manual_seed(1234)
if a.synthetic:
    a.cache = a.cache or tempfile.mkdtemp(prefix="qfx_cache_")
    if rank == 0 and not glob.glob(os.path.join(a.cache, "metadata", "*.json")):
        g = torch.Generator().manual_seed(0)
        for i in range(a.synthetic):
            write_cache_sample(a.cache, f"{i:032x}", dict(image_latents=torch.randn(1024, 64, generator=g), control_latents=torch.randn(1024, 64, generator=g),
                                                        prompt_embeds=torch.randn(384, 3584, generator=g) * 4, prompt_embeds_mask=torch.ones(384)),
                               img_shapes=[(3, 512, 512), (3, 512, 512)])
    if world > 1:
        torch.distributed.barrier()
cfg = {}
if a.transformer:
    cfg = {k: v for k, v in json.load(open(os.path.join(a.transformer, "config.json"))).items()
           if k in ("patch_size", "in_channels", "out_channels", "num_layers", "attention_head_dim", "num_attention_heads", "joint_attention_dim", "axes_dims_rope")}
else:
    cfg = dict(num_layers=a.layers)
with torch.device(dev):
    dit = QwenImageTransformer2DModel(**cfg)
if a.transformer:
    from safetensors.torch import load_file
    for f in sorted(glob.glob(os.path.join(a.transformer, "*.safetensors"))):
        dit.load_state_dict(load_file(f, device=str(dev)), strict=False)
else:
    with torch.no_grad():
        for n, p in dit.named_parameters():
            p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=a.rank, lora_alpha=a.rank), "lora_edit", generator=torch.Generator().manual_seed(1234))
step = QwenLoraTrainStep(dit, lr=a.lr)
sched = get_scheduler("constant_with_warmup", num_warmup_steps=a.warmup)
loader = PrefetchLoader(CachedEmbeddingDataset(a.cache), batch_size=a.batch, device=dev, rank=rank, world=world)
done, epoch, t0 = 0, 0, None
while done < a.steps:
    loader.set_epoch(epoch)
    for b in loader:
        emb = dict(image_latents=b["image_latents"], control_latents=b["control_latents"], prompt_embeds=b["prompt_embeds"],
                   prompt_embeds_mask=b["prompt_embeds_mask"].long(), img_shapes=convert_img_shapes_to_latent_space(b["img_shapes"]))
        step.lr = a.lr * sched(step.global_step)
        loss = step.gather_loss(step.train_step(emb))
        done += 1
        if done == 2:
            torch.cuda.synchronize(); t0 = time.time(); n0 = done
        if rank == 0 and (done % max(1, a.steps // 5) == 0 or done == a.steps):
            print(f"step {done}: loss {loss.item():.4f} lr {step.lr:.2e}", flush=True)
        if done >= a.steps:
            break
    epoch += 1
torch.cuda.synchronize()
if rank == 0:
    if t0 is not None and done > n0:
        print(f"{(done - n0) * a.batch * world / (time.time() - t0):.2f} images/s over {world} GPU(s)")
    if a.out:
        step.save_checkpoint(a.out, extra_state={"epoch": epoch})
        print("checkpoint:", sorted(os.listdir(a.out)))
