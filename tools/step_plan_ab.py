#!/usr/bin/env python
"""A/B of PLAN-BUILD-TIME environment levers on the real training step (60 blocks, side-stream gradient launches, optimizer):
    python tools/step_plan_ab.py base,QFX_ATTN_BWD_CONC=1[,KEY=VAL+KEY=VAL...] [--steps 20] [--rounds 3] [--out f.json]
tools/step_ablate.py switches levers the LIBRARY reads per launch; levers read while the launch programs are emitted (python side) need
their own plan.  One model; per variant the plan cache is cleared and rebuilt under the variant's environment, then the timing rounds
swap the cached plans in and out (whole steps back to back, rounds alternate: the step is package-power-limited)."""
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
    ap.add_argument("variants")
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--side", type=int, default=32, help="latent side of the target (and control) image: 32 = 512^2 (S_i = 2048), 64 = 1024^2 (cfg #4 shape)")
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
    B, side, T = args.batch, args.side, 384
    emb = dict(image_latents=torch.randn(B, side * side, 64).half().to(dev), control_latents=torch.randn(B, side * side, 64).half().to(dev),
               prompt_embeds=(torch.randn(B, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, side, side)] * 2] * B)
    names = args.variants.split(",")
    plans, env_keys, first_loss = {}, set(), {}

    def set_env(v):
        for k in env_keys:
            os.environ.pop(k, None)
        if "=" in v:
            for kv in v.split("+"):
                k, _, val = kv.partition("=")
                os.environ[k] = val
                env_keys.add(k)

    for n in names:
        set_env(n)
        dit._plans.clear()
        dit._plans.sizes.clear()
        torch.manual_seed(7)
        losses = [float(step.train_step(emb)) for _ in range(3)]
        torch.cuda.synchronize()
        plans[n] = dict(dit._plans)
        first_loss[n] = losses[0]
        print(f"built {n}: first loss {losses[0]:.5f}", flush=True)

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
            set_env(n)
            dit._plans.clear()
            dit._plans.update(plans[n])
            ms = timed()
            if r:
                res[n].append(ms)
    out = {"unit": "ms per step", "steps": args.steps, "batch": B, "side": side, "variants": {}}
    base = sorted(res[names[0]])[len(res[names[0]]) // 2]
    for n in names:
        med = sorted(res[n])[len(res[n]) // 2]
        out["variants"][n] = {"median_ms": round(med, 3), "delta_ms": round(med - base, 3), "all": [round(x, 3) for x in res[n]], "first_loss": first_loss[n]}
        print(f"{n:40s} {med:8.2f} ms  ({med - base:+.2f})   {['%.2f' % x for x in res[n]]}")
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
