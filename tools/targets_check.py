#!/usr/bin/env python
"""Headline shape (cfg #2: 512^2 target + 512^2 control, T=384, r=16, B=1) with adapters beyond the reference's default four:
attention projections + the feed-forward linears of both streams (8 sites per block, K or N = 12288 for the feed-forward ones),
once with AdamW and once with the fused Prodigy step.  Checks finiteness / progress on a fixed batch and records the step time."""
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
targets = ["to_k", "to_q", "to_v", "to_out.0", "img_mlp.net.0.proj", "img_mlp.net.2", "txt_mlp.net.0.proj", "txt_mlp.net.2"]
names = dit.add_adapter(LoraConfig(r=16, lora_alpha=16, target_modules=targets), "default", generator=torch.Generator().manual_seed(0))
S_t, T = 1024, 384
emb = dict(image_latents=torch.randn(1, S_t, 64).half().to(dev), control_latents=torch.randn(1, S_t, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
noise = torch.randn(1, S_t, 64); u = torch.tensor([0.4])
out = {"config": "cfg #2 shape, r=16, adapters on attention + feed-forward linears", "adapters": len(names),
       "lora_params_M": round(sum(p.numel() for p in dit.lora_parameters()) / 1e6, 2)}
for opt, kw in (("adamw", dict(lr=5e-4)), ("prodigy", dict(lr=1.0, weight_decay=0.01, optimizer="prodigy",
                                                             optimizer_args=dict(use_bias_correction=True, safeguard_warmup=True)))):
    step = QwenLoraTrainStep(dit, **kw)
    losses = []
    for i in range(10):
        if i == 4:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        losses.append(step.train_step(emb, noise=noise, u=u).item())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    out[opt] = {"ms_per_step": round(dt * 1e3, 1), "losses": [round(x, 4) for x in losses]}
    assert all(x == x and x < 1e4 for x in losses), losses
out["mem_GB"] = round(torch.cuda.max_memory_allocated() / 2**30, 1)
print(json.dumps(out))
assert out["adamw"]["losses"][-1] < out["adamw"]["losses"][0]
