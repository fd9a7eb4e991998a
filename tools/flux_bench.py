#!/usr/bin/env python
"""FLUX.1-Kontext-dev sized LoRA training step on one MI355X (NOT the driver's bench: the headline metric is the Qwen config).
19 double + 38 single blocks, D=3072, T=512, 512x512 target + 512x512 control (S_i=2048), r=16 on to_q/to_k/to_v/to_out.0."""
import argparse, json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.models import FluxTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import FluxKontextTrainStep
from qflux_amd.trainer.flux_step import prepare_latent_image_ids

ap = argparse.ArgumentParser()
ap.add_argument("--double", type=int, default=19); ap.add_argument("--single", type=int, default=38)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--res", type=int, default=512); ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--quant", default="", help='MX-FP8 trunk mode: "mxfp8" | "mxfp8-fb"')
ap.add_argument("--targets", default="", help='"all-linear", or the regex of configs/face_seg_flux_kontext_fp16.yaml:11 with "regex"')
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
with torch.device(dev):
    dit = FluxTransformer2DModel(num_layers=a.double, num_single_layers=a.single, guidance_embeds=True)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n and p.ndim == 1 and "linear" not in n else p.normal_(0.0, 0.02))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REGEX = None
if a.targets == "regex":      # the reference's shipped broad regex is test data: tests/test_flux_gpu.py::_REFERENCE_REGEX
    import importlib.util
    spec = importlib.util.spec_from_file_location("tfg", os.path.join(ROOT, "tests", "test_flux_gpu.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); REGEX = m._REFERENCE_REGEX
kw = {} if not a.targets else dict(target_modules=("all-linear" if a.targets == "all-linear" else REGEX))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16, **kw), "default", generator=torch.Generator().manual_seed(0))
if a.quant:
    dit.quantize_trunk(a.quant)
step = FluxKontextTrainStep(dit)
side = a.res // 16; S_t = side * side; T = 512; B = a.batch
ctl = prepare_latent_image_ids(side, side); ctl[:, 0] = 1
emb = dict(image_latents=torch.randn(B, S_t, 64).half().to(dev), control_latents=torch.randn(B, S_t, 64).half().to(dev),
           prompt_embeds=(torch.randn(B, T, 4096)).half().to(dev), pooled_prompt_embeds=torch.randn(B, 768).half().to(dev),
           text_ids=torch.zeros(T, 3), control_ids=ctl, latent_hw=(side, side))
for _ in range(a.warmup): step.train_step(emb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): loss = step.train_step(emb)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
S = T + 2 * S_t; D = 3072
f_lin = 2 * (a.double * S * 12 * D * D + a.single * S * (3 * D * D + 4 * D * D + 5 * D * D))
f_attn = 2 * (a.double + a.single) * 2 * S * S * D
tf = (2 * f_lin + 3.5 * f_attn) * B / 1e12
print(json.dumps({"model": f"FLUX-Kontext-sized DiT {a.double}+{a.single} blocks", "targets": a.targets or "default", "trunk": a.quant or "bf16", "images_per_s": round(B / dt, 3), "ms_per_step": round(dt * 1e3, 2),
                  "step_tflop_algorithmic": round(tf, 1), "tflops": round(tf / dt, 1), "loss": float(loss.item()),
                  "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
