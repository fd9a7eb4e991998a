#!/usr/bin/env python
"""BASELINE.json config #3 at full size on one MI355X: Qwen-Image-Edit-2509, 512^2 target + TWO 512^2 controls (S_i=3072, frame
index 0/1/2), T=512, LoRA r=32.  Checks that the step runs, the loss is finite and decreasing over a few steps on a fixed batch."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 60
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=layers)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=32, lora_alpha=32), "default", generator=torch.Generator().manual_seed(0))
step = QwenLoraTrainStep(dit, lr=5e-4)
S_t, T = 1024, 512
emb = dict(image_latents=torch.randn(1, S_t, 64).half().to(dev), control_latents=torch.randn(1, 2 * S_t, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None,
           img_shapes=[[(1, 32, 32), (1, 32, 32), (1, 32, 32)]])
noise = torch.randn(1, S_t, 64); u = torch.tensor([0.4])
losses = []
for i in range(8):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    losses.append(step.train_step(emb, noise=noise, u=u).item())
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(json.dumps({"config": "cfg #3: 3 images (S_i=3072), T=512, r=32", "ms_per_step": round(dt * 1e3, 1), "images_per_s": round(1 / dt, 2),
                  "losses": [round(x, 4) for x in losses], "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
assert all(map(lambda x: x == x and x < 1e4, losses)) and losses[-1] < losses[0]
