#!/usr/bin/env python
"""Same-box A/B of the persistent GEMM's tile-geometry policies (qfx_gemm_tune, round 4) on the HEADLINE step.

    python tools/tiles_ab.py [--layers 60] [--steps 12] [--rounds 3] [--policies legacy,n160,...]

One process, one model (BASELINE configs[1]: 60 blocks, S_i = 2048, T = 384, r = 16, B = 1): per policy (a) whole training steps
back to back, the policies interleaved round-robin (sustained clocks: the step runs at the package power limit, burst timings hide
energy effects), (b) one profiled replay with a HIP-event pair around every GEMM launch, summed per launch class, (c) loss and flat
LoRA gradient of one fixed batch (the geometry must not change results: same K order per output element).  Prints one JSON line and
writes gpurun_out/tiles_ab.json."""
from __future__ import annotations

import argparse
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
sys.path.insert(0, ROOT)

POLICIES = {
    "legacy": "256x128,256x256",           # rounds 1-3
    "n160": "256x128,160x192,256x256",     # all three: the cost model decides
    "n160only": "160x192,256x256",       # (the launcher itself keeps 256x128 for gate + residual when it is enabled)
    "all": "all",                          # the launcher's own cost model over all five
}


def klass(g, n):
    return f"N={g.N} K={g.K1}+{g.K2} epi={g.epi} groups={n} M={g.M}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--policies", default="legacy,n160,all")
    ap.add_argument("--eff", default="", help="three efficiencies for the 'all' policy (qfx_gemm_tune)")
    args = ap.parse_args()
    import bench as Bn
    from qflux_amd import _lib as L
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    with torch.device(dev):
        dit = QwenImageTransformer2DModel(num_layers=args.layers)
    with torch.no_grad():
        for n, p in dit.named_parameters():
            if p.ndim == 2:
                p.normal_(0.0, 0.02)
            elif "norm" in n:
                p.fill_(1.0)
            elif ".img_mod." in n or ".txt_mod." in n or "norm_out" in n:
                p.normal_(0.0, 0.02)
            else:
                p.zero_()
    dit.add_adapter(LoraConfig(r=16, lora_alpha=16, init_lora_weights="gaussian"), "default", generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        for n, p in dit.named_parameters():
            if "lora_B" in n:
                p.normal_(0.0, 1e-2)
    step = QwenLoraTrainStep(dit, lr=0.0, weight_decay=0.0, max_grad_norm=1.0)      # lr 0: every policy sees the same adapters
    B, side, T = args.batch, 32, 384
    S_t = side * side
    emb = dict(image_latents=torch.randn(B, S_t, 64).half().to(dev), control_latents=torch.randn(B, S_t, 64).half().to(dev),
               prompt_embeds=(torch.randn(B, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None,
               img_shapes=[[(1, side, side), (1, side, side)]] * B)
    noise = torch.randn(B, S_t, 64)
    u = torch.full((B,), 0.37)
    names = [p for p in args.policies.split(",") if p]

    def tune(name):
        rc = L.lib.qfx_gemm_tune(POLICIES[name].encode(), (args.eff.encode() if (name == "all" and args.eff) else None))
        assert rc == 0, (name, rc)

    for _ in range(4):
        step.train_step(emb)
    torch.cuda.synchronize()
    res = {n: dict(ms=[]) for n in names}
    # (c) results must not depend on the geometry
    ref = None
    for n in names:
        tune(n)
        step.zero_grad()
        loss = step.forward_backward(emb, noise=noise, u=u).item()
        g = dit.lora_store.gflat.clone()
        step.zero_grad()
        if ref is None:
            ref = (loss, g)
        res[n]["loss"] = loss
        res[n]["grad_rel_vs_first"] = ((g - ref[1]).abs().max() / ref[1].abs().max()).item()
        plan = list(dit._plans.values())[0]
        res[n]["pred_equal_first"] = None
        pred = plan.A["out"].clone()
        if n == names[0]:
            pred0 = pred
        res[n]["pred_equal_first"] = bool(torch.equal(pred, pred0))
    # (a) sustained steps, interleaved
    for r in range(args.rounds):
        for n in names:
            tune(n)
            step.train_step(emb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step.train_step(emb)
            torch.cuda.synchronize()
            res[n]["ms"].append((time.perf_counter() - t0) / args.steps * 1e3)
    # (b) per-class GEMM launch times
    plan = list(dit._plans.values())[0]
    gemm_fns = (L.lib.qfx_gemm_bf16, L.lib.qfx_gemm_grouped)
    for n in names:
        tune(n)
        dit.refresh_lora_operands()
        per = collections.OrderedDict()
        tot = 0.0
        for rep in range(2):
            for prog in (plan.fwd, plan.bwd):
                ev = Bn.run_profiled(prog, gemm_fns)
                torch.cuda.synchronize()
                k = 0
                for ent in prog.calls:
                    fn, a = ent[0], ent[1]
                    if fn in gemm_fns:
                        obj = getattr(a[0], "_obj", a[0])
                        gs = [obj] if isinstance(obj, L.GemmArgs) else [obj[i] for i in range(a[1])]
                        key = klass(gs[0], len(gs))
                        us = ev[k][0].elapsed_time(ev[k][1]) * 1e3
                        k += 1
                        if rep == 1:
                            c = per.setdefault(key, [0, 0.0])
                            c[0] += 1; c[1] += us
                            tot += us
            step.zero_grad()
        res[n]["gemm_ms_per_step"] = tot / 1e3
        res[n]["classes_us"] = {k: dict(launches=v[0], avg_us=round(v[1] / v[0], 1)) for k, v in per.items() if v[0] >= 10}
    for n in names:
        ms = res[n]["ms"]
        res[n]["ms_median"] = sorted(ms)[len(ms) // 2]
    out = dict(config=f"Qwen {args.layers} blocks B={B} 512^2 r=16", steps_per_round=args.steps, policies={n: POLICIES[n] for n in names}, results=res)
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tiles_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
