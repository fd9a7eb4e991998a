#!/usr/bin/env python
"""What is a launch class worth on the critical path?  Time the real step (side-stream gradient launches and all), then time it
again with every launch of one class REMOVED from the launch programs (the numbers the step computes are then wrong -- this is a
timing experiment only), same process, same plan, alternating.  The difference is the most a perfect kernel for that class
could return; compare with the class's summed kernel time to see how much of it is hidden / returns as interference.

    python tools/step_whatif.py --drop qfx_lora_grad_batch,qfx_lora_down_batch+qfx_lora_down,qfx_qk_norm_rope_fwd
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--drop", default="qfx_lora_grad_batch")
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--steps", type=int, default=15)
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
    names = sorted({c[0].__name__ for c in full["fwd"] + full["bwd"] if c[0] is not None})
    print("launch classes:", ", ".join(names))

    def timed():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step.train_step(emb)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    def install(drop):
        for k, prog in (("fwd", plan.fwd), ("bwd", plan.bwd)):
            assert not prog.marks or k == "bwd"
            prog.calls[:] = [c for c in full[k] if c[0] is None or c[0].__name__ not in drop]

    variants = [("full", set())] + [(d, set(d.split("+"))) for d in args.drop.split(",") if d]
    res = {n: [] for n, _ in variants}
    counts = {}
    for _ in range(args.rounds):
        for n, drop in variants:
            install(drop)
            counts[n] = sum(1 for k in full for c in full[k] if c[0] is not None and c[0].__name__ in drop)
            timed() if not res[n] else None
            res[n].append(timed())
    install(set())
    out = {"unit": "ms per step (median of rounds)", "layers": args.layers, "steps": args.steps, "variants": {}}
    base = sorted(res["full"])[len(res["full"]) // 2]
    for n, _ in variants:
        med = sorted(res[n])[len(res[n]) // 2]
        out["variants"][n] = {"ms": med, "saved_vs_full_ms": base - med, "launches_dropped": counts[n], "all": res[n]}
        print(f"{n:60s} {med:8.2f} ms   saved {base - med:6.2f} ms   ({counts[n]} launches dropped)")
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
