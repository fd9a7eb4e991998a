#!/usr/bin/env python
"""Sustained-load A/B of kernel-variant libraries on the REAL training step (60 blocks, side-stream gradient launches, optimizer):
    python tools/step_lib_ab.py base,<variant>[,...] [--steps 30] [--rounds 3]
Unlike tools/step_ab.py (per-launch events, bursts: the chip never reaches its power cap there) this runs whole steps back to
back, so a variant that only saves ENERGY shows up as time -- the step is package-power-limited (profiles/HISTORY.md).  One model,
one plan; the launch programs' C calls are re-bound to the same symbol of each variant library, rounds alternate."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants")
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--gflat-repro", action="store_true", help="per variant: two forward+backward passes on identical draws, is the flat LoRA gradient bit-identical?")
    args = ap.parse_args()
    from step_ab import load_variant
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
    names = args.variants.split(",")
    libs = {n: load_variant(n) for n in names}

    def install(lib):
        for k, prog in (("fwd", plan.fwd), ("bwd", plan.bwd)):
            prog.calls[:] = [c if c[0] is None else (getattr(lib, c[0].__name__),) + tuple(c[1:]) for c in full[k]]

    def timed():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step.train_step(emb)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3, float(loss)

    repro = {}
    if args.gflat_repro:
        gen = torch.Generator().manual_seed(3)
        noise = torch.randn(emb["image_latents"].shape, generator=gen)
        u = torch.rand(B, generator=gen)
        for n in names:
            install(libs[n])
            gs = []
            for _ in range(3):
                step.zero_grad()
                step.forward_backward(emb, noise=noise, u=u)
                torch.cuda.synchronize()
                gs.append(dit.lora_store.gflat.clone())
            step.zero_grad()
            repro[n] = {"bit_identical": bool(all(torch.equal(gs[0].view(torch.int32), g.view(torch.int32)) for g in gs[1:])),
                        "max_rel_diff": float(max(((g - gs[0]).abs().max() / gs[0].abs().max()).item() for g in gs[1:]))}
            print("gflat repro", n, repro[n])
    res = {n: [] for n in names}
    losses = {}
    for r in range(args.rounds + 1):
        for n in names:
            install(libs[n])
            ms, loss = timed()
            losses.setdefault(n, loss)
            if r:
                res[n].append(ms)       # round 0 warms every variant's code objects
    out = {"unit": "ms per step", "steps": args.steps, "variants": {}, "gflat_repro": repro}
    for n in names:
        med = sorted(res[n])[len(res[n]) // 2]
        out["variants"][n] = {"median_ms": med, "all": res[n], "first_loss": losses[n]}
        print(f"{n:12s} {med:8.2f} ms   {['%.2f' % x for x in res[n]]}   loss {losses[n]:.5f}")
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
