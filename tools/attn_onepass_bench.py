#!/usr/bin/env python
"""Round 6: the one-pass attention backward (qfx_attn_bwd_fused = dsum prep + attn_bwd1_kernel + dQ finish) against the two-pass pair
(qfx_attn_bwd_dq + qfx_attn_bwd_dkv) on one box, interleaved, with the step's fused epilogues on (QK-norm + RoPE backward, rank-r
projections of the q / k / v adapters): us per layer-launch, best and median of N rounds, and output agreement.
    python tools/attn_onepass_bench.py [--S 2432,8576,2432:24:2] [--rounds 8] [--plain]"""
import argparse, ctypes as C, json, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from qflux_amd import ops, _lib as L
import test_attention_onepass_gpu as T

ap = argparse.ArgumentParser(); ap.add_argument("--S", default="2432,8576"); ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--libs", default="", help="variant libraries (tools/build_variants.py) whose one-pass entry is timed too (outputs not compared)")
ap.add_argument("--only", default="", help="time only this entry (e.g. one_pass@a31): target of counter passes")
ap.add_argument("--plain", action="store_true", help="without the fused epilogues"); ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
out = {}
st = ops.stream_ptr()
for spec in args.S.split(","):
    S, H, Bn = (int(x) for x in (spec.split(":") + ["24", "1"])[:3])
    a, t, keep = T._setup(S, H, Bn, 0, 0 if args.plain else 16, fused=not args.plain, seed=3)
    ws = ops.attn_bwd_fused_workspace(a)
    fns = {"dq": [L.lib.qfx_attn_bwd_dq], "dkv": [L.lib.qfx_attn_bwd_dkv], "two_pass": [L.lib.qfx_attn_bwd_dq, L.lib.qfx_attn_bwd_dkv],
           "one_pass": [L.lib.qfx_attn_bwd_fused]}
    if args.libs:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from step_ab import load_variant
        for nm in args.libs.split(","):
            fns["one_pass@" + nm] = [load_variant(nm).qfx_attn_bwd_fused]
    if args.only:
        fns = {k: v for k, v in fns.items() if k == args.only}
    res = {k: [] for k in fns}
    for rnd in range(args.rounds):
        for k, fl in fns.items():
            for f in fl:
                assert f(C.byref(a), st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                for f in fl:
                    f(C.byref(a), st)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / args.iters * 1e3)
    if args.only:
        print(spec, {k: min(v) for k, v in res.items()}); continue
    two = T._run_pair(a, t, "two"); one = T._run_pair(a, t, "fused")
    D = t["D"]
    r = {k: {"us_best": round(min(v), 1), "us_median": round(sorted(v)[len(v) // 2], 1)} for k, v in res.items()}
    r["one_over_two"] = round(min(res["one_pass"]) / min(res["two_pass"]), 4)
    r["rel_diff_dq"] = T._rel(one[0][:, :, :D], two[0][:, :, :D]); r["rel_diff_dkdv"] = T._rel(one[0][:, :, D:], two[0][:, :, D:])
    r["algorithmic_tflops_one_pass"] = round(10 * S * S * 128 * H * Bn / min(res["one_pass"]) / 1e6, 1)
    out[f"S{S}_H{H}_B{Bn}"] = r
    print(f"S={S} H={H} B={Bn}", json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"unit": "us per launch group (one layer's attention backward)", "epilogues": "plain" if args.plain else "qk-norm/rope backward + rank-16 projections fused", "results": out},
          open(os.path.join(ROOT, "gpurun_out", "r06_attn_onepass.json"), "w"), indent=1)
