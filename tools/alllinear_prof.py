#!/usr/bin/env python
"""All-linear LoRA step (target_modules="all-linear", cfg #2 shape) for profiling: python tools/alllinear_prof.py [steps] [layers]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qwen-image-finetune_amd"))
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0); torch.manual_seed(1234)
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=layers)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16, target_modules="all-linear"), "default", generator=torch.Generator().manual_seed(0))
step = QwenLoraTrainStep(dit, lr=1e-4)
emb = dict(image_latents=torch.randn(1, 1024, 64).half().to(dev), control_latents=torch.randn(1, 1024, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, 384, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
for _ in range(2): step.train_step(emb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step.train_step(emb)
torch.cuda.synchronize()
print(f"all-linear: {(time.perf_counter() - t0) / steps * 1e3:.1f} ms/step")
