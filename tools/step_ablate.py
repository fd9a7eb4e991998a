#!/usr/bin/env python
"""What does each class of small launches COST the real training step?  (upper bound of what fusing it away could return)
    python tools/step_ablate.py [--steps 30] [--rounds 3] [--out profiles/r05_step_ablation.json]
One model, one plan; per variant the named C calls are removed from the launch programs (results are garbage, only time is read),
whole steps back to back (the step is package-power-limited), rounds alternate.  `side_main` moves the side-stream launches to the
main stream instead of dropping them."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))

VARIANTS = {
    "full": (),
    "no_lora_grad": ("qfx_lora_grad_batch", "qfx_lora_grad"),
    "no_head_reduce": ("qfx_lora_head_reduce",),
    "no_lora_down": ("qfx_lora_down", "qfx_lora_down_batch"),
    "no_qk_norm_rope": ("qfx_qk_norm_rope_fwd",),
    "no_ln_bwd": ("qfx_ln_modulate_bwd", "qfx_ln_modulate_bwd_batch"),
    "no_small": ("qfx_lora_grad_batch", "qfx_lora_grad", "qfx_lora_head_reduce", "qfx_lora_down", "qfx_lora_down_batch", "qfx_qk_norm_rope_fwd"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="full,no_lora_grad,side_main,no_head_reduce,no_lora_down,no_qk_norm_rope,no_small")
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    with torch.device(dev):
        dit = QwenImageTransformer2DModel(num_layers=args.layers)
    with torch.no_grad():
        for n, p in dit.named_parameters():
            if "norm" in n and p.ndim == 1:
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02)
    dit.add_adapter(LoraConfig(r=16, lora_alpha=16, init_lora_weights="gaussian"), "default", generator=torch.Generator().manual_seed(1))
    step = QwenLoraTrainStep(dit, lr=1e-4)
    B, side, T = 1, 32, 384
    emb = dict(image_latents=torch.randn(B, side * side, 64).half().to(dev), control_latents=torch.randn(B, side * side, 64).half().to(dev),
               prompt_embeds=(torch.randn(B, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, side, side)] * 2] * B)
    for _ in range(3):
        step.train_step(emb)
    torch.cuda.synchronize()
    plan = list(dit._plans.values())[0]
    full = {"fwd": list(plan.fwd.calls), "bwd": list(plan.bwd.calls)}
    names_seen = sorted({c[0].__name__ for k in full for c in full[k] if c[0] is not None})
    counts = {n: sum(1 for k in full for c in full[k] if c[0] is not None and c[0].__name__ == n) for n in names_seen}
    print("calls per step:", counts)
    names = args.variants.split(",")

    env_keys = set()

    def install(v):
        # a variant of the form KEY=VAL[+KEY=VAL...] runs the full programs under those environment levers (read per launch by the library)
        for k in env_keys:
            os.environ.pop(k, None)
        if "=" in v:
            for kv in v.split("+"):
                k, _, val = kv.partition("=")
                os.environ[k] = val
                env_keys.add(k)
        drop = VARIANTS.get(v, ())
        for k, prog in (("fwd", plan.fwd), ("bwd", plan.bwd)):
            calls = []
            for c in full[k]:
                if c[0] is not None and c[0].__name__ in drop:
                    continue
                if v == "side_main" and c[0] is not None and len(c) > 2:
                    c = c[:2]
                calls.append(c)
            prog.calls[:] = calls

    def timed():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step.train_step(emb)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    res = {n: [] for n in names}
    for r in range(args.rounds + 1):
        for n in names:
            install(n)
            ms = timed()
            if r:
                res[n].append(ms)
    install("full")
    out = {"unit": "ms per step", "steps": args.steps, "calls_per_step": counts, "variants": {}}
    base = sorted(res[names[0]])[len(res[names[0]]) // 2]
    for n in names:
        med = sorted(res[n])[len(res[n]) // 2]
        out["variants"][n] = {"median_ms": round(med, 3), "delta_ms": round(med - base, 3), "all": [round(x, 3) for x in res[n]], "dropped": list(VARIANTS.get(n, ()))}
        print(f"{n:16s} {med:8.2f} ms  ({med - base:+.2f})   {['%.2f' % x for x in res[n]]}")
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
