#!/usr/bin/env python
"""Round-2 sweep of the non-headline configurations on one MI355X (one JSON line per case into gpurun_out/configs_r02.jsonl):
cfg #2 at B=2/4, cfg #3 (3 images, r=32), cfg #4 shape (1024^2), all-linear adapters, the MX-FP8 trunk mode, a 150-step soak."""
import gc, json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
dev = torch.device("cuda", 0)
out_path = os.path.join(ROOT, "gpurun_out", "configs_r02.jsonl")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
open(out_path, "w").close()


def emit(rec):
    print(json.dumps(rec), flush=True)
    with open(out_path, "a") as f:
        f.write(json.dumps(rec) + "\n")


def model(r=16, targets=None, layers=60):
    torch.manual_seed(1234)
    with torch.device(dev):
        dit = QwenImageTransformer2DModel(num_layers=layers)
    with torch.no_grad():
        for n, p in dit.named_parameters():
            p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
    kw = dict(target_modules=targets) if targets is not None else {}
    dit.add_adapter(LoraConfig(r=r, lora_alpha=r, **kw), "default", generator=torch.Generator().manual_seed(0))
    return dit


def embeddings(B, side, T, n_ctrl=1):
    S_t = side * side
    return dict(image_latents=torch.randn(B, S_t, 64).half().to(dev), control_latents=torch.randn(B, n_ctrl * S_t, 64).half().to(dev),
                prompt_embeds=(torch.randn(B, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None,
                img_shapes=[[(1, side, side)] * (1 + n_ctrl)] * B)


def timeit(step, emb, warm=3, n=8):
    for _ in range(warm):
        step.train_step(emb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        loss = step.train_step(emb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, loss.item()


dit = model()
step = QwenLoraTrainStep(dit, lr=1e-4)
for B in (1, 2, 4):
    dt, loss = timeit(step, embeddings(B, 32, 384))
    emit({"case": f"cfg#2 512^2 r=16 B={B}", "ms_per_step": round(dt * 1e3, 1), "images_per_s": round(B / dt, 2), "loss": round(loss, 4)})
    dit._plans.clear(); gc.collect(); torch.cuda.empty_cache()
dt, loss = timeit(step, embeddings(1, 64, 384), warm=2, n=4)
emit({"case": "cfg#4 shape 1024^2 (S=8576) r=16 B=1", "ms_per_step": round(dt * 1e3, 1), "images_per_s": round(1 / dt, 3), "loss": round(loss, 4),
      "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)})
dit._plans.clear(); gc.collect(); torch.cuda.empty_cache()
# MX-FP8 trunk mode at the headline shape (forward GEMMs in fp8; first-stage kernel, expected slower than bf16)
dit.quantize_trunk("mxfp8")
dt, loss = timeit(step, embeddings(1, 32, 384))
emit({"case": "cfg#2 with quantize_trunk('mxfp8') (forward GEMMs MX-FP8)", "ms_per_step": round(dt * 1e3, 1), "images_per_s": round(1 / dt, 2), "loss": round(loss, 4)})
dit.quantize_trunk(None)
# soak: 150 steps on fresh batches
losses, times = [], []
for i in range(150):
    emb = embeddings(1, 32, 384)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = step.train_step(emb)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0); losses.append(l.item())
ts = sorted(times[10:])
emit({"case": "soak 150 steps cfg#2", "all_finite": all(x == x and abs(x) < 1e6 for x in losses), "loss_first10": round(sum(losses[:10]) / 10, 4),
      "loss_last10": round(sum(losses[-10:]) / 10, 4), "ms_median": round(ts[len(ts) // 2] * 1e3, 2), "ms_p95": round(ts[int(len(ts) * 0.95)] * 1e3, 2),
      "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)})
del step, dit; gc.collect(); torch.cuda.empty_cache()
dit = model(r=32)
step = QwenLoraTrainStep(dit, lr=1e-4)
dt, loss = timeit(step, embeddings(1, 32, 512, n_ctrl=2))
emit({"case": "cfg#3 3 images (S_i=3072) T=512 r=32", "ms_per_step": round(dt * 1e3, 1), "images_per_s": round(1 / dt, 2), "loss": round(loss, 4)})
del step, dit; gc.collect(); torch.cuda.empty_cache()
dit = model(targets="all-linear")
step = QwenLoraTrainStep(dit, lr=1e-4)
dt, loss = timeit(step, embeddings(1, 32, 384))
emit({"case": "cfg#2 shape, target_modules='all-linear' (every Linear adapted, conditioning head through cond_hip.py)", "ms_per_step": round(dt * 1e3, 1),
      "images_per_s": round(1 / dt, 2), "loss": round(loss, 4), "lora_params_M": round(sum(p.numel() for p in dit.lora_parameters()) / 1e6, 1)})
