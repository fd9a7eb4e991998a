#!/usr/bin/env python
"""Where does the host-fed step lose time against the resident one?  Variants of feeding the same fused step (60 blocks, B=1, 512^2)."""
import os, sys, tempfile, threading, time, shutil, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.data import CachedEmbeddingDataset, PrefetchLoader, convert_img_shapes_to_latent_space, write_cache_sample
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
dev = torch.device("cuda", 0); torch.manual_seed(0)
L = int(os.environ.get("LAYERS", "60"))
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=L)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
step = QwenLoraTrainStep(dit, lr=1e-4)
S_t, T, Jd = 1024, 384, 3584
shapes = [[(1, 32, 32), (1, 32, 32)]]
host = dict(image_latents=torch.randn(1, S_t, 64).half(), control_latents=torch.randn(1, S_t, 64).half(), prompt_embeds=(torch.randn(1, T, Jd) * 4).half(),
            prompt_embeds_mask=torch.ones(1, T))
res = {k: v.to(dev) for k, v in host.items()}
N = 12


def timed(fn, n=N):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


e_a = dict(res, prompt_embeds_mask=None, img_shapes=shapes)
print("A resident, mask None           %.2f ms" % timed(lambda: step.train_step(e_a)), flush=True)
e_b = dict(res, img_shapes=shapes); e_b["prompt_embeds_mask"] = res["prompt_embeds_mask"].long()
print("B resident, device mask         %.2f ms" % timed(lambda: step.train_step(e_b)), flush=True)
pin = {k: v.pin_memory() for k, v in host.items()}
def c():
    e = {k: v.to(dev, non_blocking=True) for k, v in pin.items()}; e["img_shapes"] = shapes; step.train_step(e)
print("C pinned host -> .to() in line  %.2f ms" % timed(c), flush=True)
stop = False
root = tempfile.mkdtemp(prefix="qfx_probe_")
g = torch.Generator().manual_seed(0)
for i in range(16):
    write_cache_sample(root, f"{i:032x}", dict(image_latents=torch.randn(S_t, 64, generator=g), control_latents=torch.randn(S_t, 64, generator=g),
                                              prompt_embeds=torch.randn(T, Jd, generator=g) * 4, prompt_embeds_mask=torch.ones(T)),
                       img_shapes=[(3, 512, 512), (3, 512, 512)])
ds = CachedEmbeddingDataset(root)
def bg():
    i = 0
    while not stop:
        ds[i % 16]; i += 1
        time.sleep(0.02)
th = threading.Thread(target=bg, daemon=True); th.start()
print("F resident + a thread reading the cache (50 samples/s)  %.2f ms" % timed(lambda: step.train_step(e_a)), flush=True)
stop = True; th.join()
for workers, prefetch in ((1, 2), (4, 2), (2, 3), (4, 6), (2, 3)):
    loader = PrefetchLoader(ds, batch_size=1, device=dev, workers=workers, prefetch=prefetch)
    done, t0 = 0, None
    while done < N + 4:
        for b in loader:
            e = dict(image_latents=b["image_latents"], control_latents=b["control_latents"], prompt_embeds=b["prompt_embeds"],
                     prompt_embeds_mask=b["prompt_embeds_mask"].long(), img_shapes=convert_img_shapes_to_latent_space(b["img_shapes"]))
            step.train_step(e); done += 1
            if done == 4:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            if done >= N + 4: break
    torch.cuda.synchronize()
    print(f"D PrefetchLoader workers={workers} prefetch={prefetch}  {(time.perf_counter() - t0) / N * 1e3:.2f} ms   "
          f"(last epoch: consumer waited {loader.stats['wait_s'] * 1e3:.1f} ms, uploads took {loader.stats['upload_s'] * 1e3:.1f} ms of host time)", flush=True)
shutil.rmtree(root, ignore_errors=True)
