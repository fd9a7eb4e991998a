#!/usr/bin/env python
"""Full-size check of the quantising GEMM epilogues (MX-FP8 trunk): one forward + backward of the 60-block model with fixed noise,
stand-alone quantisation passes (QFX_FP8_FUSED_QUANT=0) vs quantising epilogues; the MX-FP8 images are bit-identical, so the
prediction must be EQUAL (the loss up to its atomic summation order) and the LoRA gradients equal up to the fp32-atomic summation order of the gradient kernels."""
import os, sys, torch
sys.path.insert(0, "qwen-image-finetune_amd")
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
L_ = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0); torch.manual_seed(1234)
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=L_)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
with torch.no_grad():
    for n, p in dit.named_parameters():
        if "lora_B" in n:
            p.normal_(0.0, 0.01)
step = QwenLoraTrainStep(dit, lr=1e-4)
g = torch.Generator().manual_seed(3)
emb = dict(image_latents=torch.randn(1, 1024, 64, generator=g).half().to(dev), control_latents=torch.randn(1, 1024, 64, generator=g).half().to(dev),
           prompt_embeds=(torch.randn(1, 384, 3584, generator=g) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
noise = torch.randn(1, 1024, 64, generator=g)
u = torch.tensor([0.37])
res = {}
for tag, env in (("separate", "0"), ("fused", "1"), ("separate2", "0")):
    os.environ["QFX_FP8_FUSED_QUANT"] = env
    dit.quantize_trunk("mxfp8-fb")
    step.zero_grad()
    loss = step.forward_backward(emb, noise=noise, u=u)
    torch.cuda.synchronize()
    plan = list(dit._plans.values())[0]
    nq = sum(1 for c in plan.fwd.calls + plan.bwd.calls if c[0] is not None and c[0].__name__ == "qfx_quant_mxfp8")
    res[tag] = (loss.item(), plan.A["out"].float().clone(), dit.lora_store.gflat.clone(), nq)
    print(tag, "loss", loss.item(), "stand-alone quantisation launches", nq, flush=True)
a, b, c = res["separate"], res["fused"], res["separate2"]
rel = lambda x, y: ((x - y).abs().max() / y.abs().max()).item()
print("pred  fused vs separate:", rel(b[1], a[1]), "   separate vs separate:", rel(c[1], a[1]))
print("grads fused vs separate:", rel(b[2], a[2]), "   separate vs separate:", rel(c[2], a[2]))
assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0]) and torch.equal(a[1], b[1]), "quantising epilogue changed the forward"   # (the loss sum is an fp32 atomic reduction)
assert rel(b[2], a[2]) <= 10 * max(rel(c[2], a[2]), 1e-7) + 1e-6
print("OK")
