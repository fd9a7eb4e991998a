#!/usr/bin/env python
"""Full-width race check of the side-stream gradient launches: the same forward+backward at the headline shape with the lora_grad
batches on the side stream (default) and inline (QFX_SIDE_GRADS=0) must give the same flat LoRA gradient up to the order of the
fp32 atomics (run-to-run noise of the inline path is measured next to it)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 12
targets = sys.argv[2] if len(sys.argv) > 2 else None      # e.g. "all-linear": feed-forward adapters put dh / v^T(fc) on the parity scratch too
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=layers)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16, **({"target_modules": targets} if targets else {})), "default",
                generator=torch.Generator().manual_seed(0))
with torch.no_grad():
    for n, p in dit.named_parameters():
        if "lora_B" in n:
            p.normal_(0.0, 0.01)
S_t, T = 1024, 384
emb = dict(image_latents=torch.randn(1, S_t, 64).half().to(dev), control_latents=torch.randn(1, S_t, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
noise = torch.randn(1, S_t, 64); u = torch.tensor([0.4])


def grads(mode, reps):
    os.environ["QFX_SIDE_GRADS"] = mode
    dit._invalidate()
    step = QwenLoraTrainStep(dit)
    out = []
    for _ in range(reps):
        step.zero_grad()
        step.forward_backward(emb, noise=noise, u=u)
        torch.cuda.synchronize()
        out.append(dit.lora_store.gflat.detach().clone())
    plan = list(dit._plans.values())[0]
    assert plan.side_grads == (mode == "1")
    return out


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


inline = grads("0", 3)
side = grads("1", 4)
res = {"layers": layers, "inline_run_to_run": max(rel(inline[i], inline[0]) for i in (1, 2)),
       "side_vs_inline": max(rel(s, inline[0]) for s in side), "side_run_to_run": max(rel(s, side[0]) for s in side[1:]),
       "grad_absmax": inline[0].abs().max().item()}
print(json.dumps(res))
assert res["side_vs_inline"] < max(10 * res["inline_run_to_run"], 1e-5), res
