#!/usr/bin/env python
"""how much of the FLUX tiny-step gradient bar (tests/parity_util.py FLUX_BARS[2] = 4e-2, observed maximum 3.75e-2) is the
bf16 graph's OWN noise?  The same bf16 oracle step (FLUX_TINY, r = 4, same weights / inputs / draws) is evaluated twice with different
reduction orders -- on the host (torch CPU kernels) and on the GPU (rocBLAS / MIOpen kernels) -- and the HIP path once; per adapter
tensor: relmax(oracle_gpu, oracle_cpu) = the eager graph against itself, relmax(hip, oracle_cpu) = what the parity test measures.
Several weight seeds.  -> gpurun_out/r06_flux_bar_noise.json"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "qwen-image-finetune_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from common import FLUX_TINY, fill_weights
from oracle import flux_dit as FO
from oracle import qwen_dit as O
from parity_util import BF, relmax, _grad_tol_factor
from qflux_amd.models import FluxTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import FluxKontextTrainStep

DEV = "cuda:0"
rows = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    cfg = dict(FLUX_TINY); cfg["guidance_embeds"] = True
    if cfg["joint_attention_dim"] % 64:
        cfg["joint_attention_dim"] = 64
    def make():
        o = FO.OracleFluxDiT(**cfg)
        O.add_lora(o, r=4, lora_alpha=8, adapter_name="lora_edit", target_modules=("to_k", "to_q", "to_v", "to_out.0"))
        fill_weights(o, seed=5 + seed)
        for n, p in o.named_parameters():
            if "lora" not in n:
                p.data = p.data.to(BF)
        return o
    oc, og = make(), make().to(DEV)
    g = torch.Generator().manual_seed(31 + seed)
    h, w, T, B = 4, 6, 7, 2
    S_t = h * w
    ids = FO.prepare_latent_image_ids(h, w); ids[:, 0] = 1
    emb = dict(image_latents=torch.randn(B, S_t, 64, generator=g).half(), control_latents=torch.randn(B, S_t, 64, generator=g).half(), control_ids=ids,
               text_ids=torch.zeros(T, 3), latent_hw=(h, w), pooled_prompt_embeds=torch.randn(B, cfg["pooled_projection_dim"], generator=g).half(),
               prompt_embeds=torch.randn(B, T, cfg["joint_attention_dim"], generator=g).half())
    noise = torch.randn(B, S_t, 64, generator=g).to(BF); t = torch.tensor([0.7109, 0.1611]).to(BF)
    eo = dict(emb, control_latents=emb["control_latents"].to(BF))
    FO.flux_compute_loss(oc, eo, noise, t, BF).float().backward()
    eg = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in eo.items()}
    FO.flux_compute_loss(og, eg, noise.to(DEV), t.to(DEV), BF).float().backward()
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**cfg)
    hip.add_adapter(LoraConfig(r=4, lora_alpha=8, target_modules=["to_k", "to_q", "to_v", "to_out.0"]), "lora_edit")
    hip.load_state_dict(oc.state_dict(), strict=True)
    FluxKontextTrainStep(hip).forward_backward(emb, noise=noise, t=t)
    torch.cuda.synchronize()
    gc = {n: p.grad for n, p in oc.named_parameters() if "lora" in n and p.grad is not None}
    gg = {n: p.grad for n, p in og.named_parameters() if "lora" in n and p.grad is not None}
    gh = {n: p.grad for n, p in hip.named_parameters() if "lora" in n}
    self_noise = {n: relmax(gg[n], gc[n]) / _grad_tol_factor(n) for n in gc}
    hip_err = {n: relmax(gh[n], gc[n]) / _grad_tol_factor(n) for n in gc}
    hip_err_g = {n: relmax(gh[n], gg[n]) / _grad_tol_factor(n) for n in gc}
    wn = max(self_noise, key=self_noise.get); wh = max(hip_err, key=hip_err.get)
    rows.append(dict(seed=seed, oracle_gpu_vs_oracle_cpu_worst=self_noise[wn], oracle_worst_tensor=wn, hip_vs_oracle_cpu_worst=hip_err[wh], hip_worst_tensor=wh,
                     hip_vs_oracle_gpu_worst=max(hip_err_g.values()), oracle_self_noise_on_hip_worst_tensor=self_noise[wh]))
    print(rows[-1], flush=True)
out = dict(rows=rows, max_oracle_self_noise=max(r["oracle_gpu_vs_oracle_cpu_worst"] for r in rows), max_hip_vs_oracle=max(r["hip_vs_oracle_cpu_worst"] for r in rows),
           note="relmax = max |a - b| / max |b| per adapter gradient tensor, worst tensor per seed; both oracle evaluations are the SAME bf16 eager graph")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_flux_bar_noise.json"), "w"), indent=1)
print({k: v for k, v in out.items() if k != "rows"})
