#!/usr/bin/env python
"""Stability soak at the headline config: N steps on fresh synthetic batches; reports loss trend, step-time spread and peak memory
(growth would show as a rising max_memory_allocated between the two halves)."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0); torch.manual_seed(1234)
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=60)
with torch.no_grad():
    for nm, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in nm else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
step = QwenLoraTrainStep(dit, lr=1e-4)
losses, times, mem = [], [], []
for i in range(n):
    emb = dict(image_latents=torch.randn(1, 1024, 64).half().to(dev), control_latents=torch.randn(1, 1024, 64).half().to(dev),
               prompt_embeds=(torch.randn(1, 384, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = step.train_step(emb)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    losses.append(l.item())
    if i in (n // 2 - 1, n - 1): mem.append(torch.cuda.max_memory_allocated() / 2**30)
ts = sorted(times[10:])
print(json.dumps({"steps": n, "loss_first10": round(sum(losses[:10]) / 10, 4), "loss_last10": round(sum(losses[-10:]) / 10, 4),
                  "all_finite": all(x == x and abs(x) < 1e6 for x in losses), "ms_median": round(ts[len(ts) // 2] * 1e3, 2),
                  "ms_p95": round(ts[int(len(ts) * 0.95)] * 1e3, 2), "mem_GB_half_full": [round(m, 2) for m in mem]}))
