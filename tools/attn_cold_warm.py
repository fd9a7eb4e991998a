#!/usr/bin/env python
"""Attention kernels on WARM buffers (one operand set launched back to back: everything sits in L2 / Infinity Cache) vs COLD buffers
(NSETS operand sets in rotation, > 256 MB in total: every launch streams its operands from HBM, as the backward of a 60-block step does).
    python tools/attn_cold_warm.py [--S 2432] [--sets 8]"""
import argparse, ctypes as C, json, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops, _lib as L
ap = argparse.ArgumentParser(); ap.add_argument("--S", type=int, default=2432); ap.add_argument("--sets", type=int, default=8); ap.add_argument("--reps", type=int, default=24)
ap.add_argument("--out", default="")
args = ap.parse_args()
BF, DEV = torch.bfloat16, "cuda:0"
S, H, Bn, dh = args.S, 24, 1, 128; D = H * dh; S_pad = (S + 63) // 64 * 64; ld = 3 * D
lib = L.lib
sets = []
for i in range(args.sets):
    torch.manual_seed(i)
    qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF); dO = torch.randn(Bn, S, D, device=DEV).to(BF)
    O = torch.zeros(Bn, S, D, dtype=BF, device=DEV); lse2 = torch.zeros(Bn, H, S_pad, device=DEV); dsum = torch.zeros(Bn, H, S_pad, device=DEV)
    dqkv = torch.zeros_like(qkv)
    a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=qkv[:, :, :D], K=qkv[:, :, D:2 * D], V=qkv[:, :, 2 * D:], ldq=ld, ldk=ld, ldv=ld,
                      O=O, ldo=D, lse2=lse2, dsum=dsum, dO=dO, lddo=D, dQ=dqkv[:, :, :D], dK=dqkv[:, :, D:2 * D], dV=dqkv[:, :, 2 * D:], lddq=ld, lddk=ld, lddv=ld)
    sets.append((a, (qkv, dO, O, lse2, dsum, dqkv)))
st = torch.cuda.current_stream().cuda_stream
for a, _ in sets:
    for f in ("qfx_attn_fwd", "qfx_attn_bwd_dq", "qfx_attn_bwd_dkv"): assert getattr(lib, f)(C.byref(a), st) == 0
torch.cuda.synchronize()
res = {}
for f in ("qfx_attn_fwd", "qfx_attn_bwd_dq", "qfx_attn_bwd_dkv"):
    fn = getattr(lib, f)
    for mode in ("warm", "cold"):
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.reps):
                a = sets[i % args.sets if mode == "cold" else 0][0]
                fn(C.byref(a), st)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.reps * 1e3)
        res.setdefault(f, {})[mode] = round(sorted(ts)[1], 1)
    print(f, res[f], flush=True)
if args.out: json.dump({"S": S, "sets": args.sets, "us": res}, open(args.out, "w"), indent=1)
