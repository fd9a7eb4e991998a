#!/usr/bin/env python
"""Does running the step on a HIGH-priority stream (side-stream gradient launches stay at normal priority) reduce the interference
of the side launches with the persistent GEMM grids?  A/B/A/B of ms/step at the headline shape."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=60)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
S_t, T = 1024, 384
emb = dict(image_latents=torch.randn(1, S_t, 64).half().to(dev), control_latents=torch.randn(1, S_t, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
step = QwenLoraTrainStep(dit)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
hi = torch.cuda.Stream(device=dev, priority=-1)


def run(n, stream):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(6):
            step.train_step(emb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            step.train_step(emb)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {"default": [], "high": []}
for _ in range(2):
    res["default"].append(round(run(25, None), 2))
    res["high"].append(round(run(25, hi), 2))
print(json.dumps(res))
