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
ap.add_argument("--multires", default="", help='cfg #5: ragged batch of buckets, e.g. "20x20,40x40" (token grids; one control of the same size each)')
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
def run():
    if not a.multires:
        return step.train_step(emb)
    loss = step.forward_backward_multires(samples, txt)
    step.optimizer_step(grad_scale=step.allreduce_grads())
    step.zero_grad()
    return loss


pad = None
if a.multires:      # the ragged two-bucket batch of tests/test_fullsize_cfgs_gpu.py (flux_kontext_trainer.py:579-796)
    g = torch.Generator().manual_seed(53)
    grids = [tuple(int(v) for v in s_.split("x")) for s_ in a.multires.split(",")]
    samples = []
    for (h, w) in grids:
        samples.append(dict(image_latents=torch.randn(h * w, 64, generator=g).half().to(dev), control_latents=torch.randn(h * w, 64, generator=g).half().to(dev),
                            hw=(h, w), control_hw=[(h, w)]))
    B = len(grids)
    txt = dict(text_ids=torch.zeros(T, 3), pooled_prompt_embeds=torch.randn(B, 768, generator=g).half().to(dev),
               prompt_embeds=torch.randn(B, T, 4096, generator=g).half().to(dev))
    toks = [2 * h * w for h, w in grids]
    pad = 1.0 - sum(toks) / (len(toks) * max(toks))
for _ in range(a.warmup): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): loss = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
D = 3072
if a.multires:
    Ss = [T + n for n in toks]
    f_lin = sum(2 * (a.double * S * 12 * D * D + a.single * S * (3 * D * D + 4 * D * D + 5 * D * D)) for S in Ss)
    f_attn = sum(2 * (a.double + a.single) * 2 * S * S * D for S in Ss)
    tf = (2 * f_lin + 3.5 * f_attn) / 1e12
else:
    S = T + 2 * S_t
    f_lin = 2 * (a.double * S * 12 * D * D + a.single * S * (3 * D * D + 4 * D * D + 5 * D * D))
    f_attn = 2 * (a.double + a.single) * 2 * S * S * D
    tf = (2 * f_lin + 3.5 * f_attn) * B / 1e12
# GEMM share of the replayed programs (HIP events around every GEMM launch of one extra step)
from qflux_amd import _lib as L
sys.path.insert(0, ROOT)
gemm_frac = None
try:
    import bench as _b
    plan = [p_ for k_, p_ in dit._plans.items() if (("multires" in str(k_)) == bool(a.multires))][-1]
    fns = (L.lib.qfx_gemm_bf16, L.lib.qfx_gemm_grouped)
    ev = _b.run_profiled(plan.fwd, fns, {}) + _b.run_profiled(plan.bwd, fns, {})
    torch.cuda.synchronize()
    gms = sum(x.elapsed_time(y) for x, y in ev)
    gf = _b.gemm_flops_of(plan.fwd)[0] + _b.gemm_flops_of(plan.bwd)[0]
    gemm_frac = {"gemm_ms": round(gms, 2), "gemm_share_of_step": round(gms / (dt * 1e3), 3), "gemm_frac_of_peak": round(gf / (gms * 1e-3) / 1e12 / 2500.0, 4)}
except Exception as e:  # noqa: BLE001
    gemm_frac = {"error": repr(e)[:200]}
print(json.dumps({"model": f"FLUX-Kontext-sized DiT {a.double}+{a.single} blocks", "targets": a.targets or "default", "trunk": a.quant or "bf16",
                  "batch": B, "multires": a.multires or None, "padded_token_fraction": None if pad is None else round(pad, 4),
                  "images_per_s": round(B / dt, 3), "ms_per_step": round(dt * 1e3, 2),
                  "step_tflop_algorithmic": round(tf, 1), "tflops": round(tf / dt, 1), "loss": float(loss.item()), "gemm": gemm_frac,
                  "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
