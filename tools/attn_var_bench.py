#!/usr/bin/env python
"""Attention kernels of variant libraries (tools/build_variants.py) against the product library on one box: interleaved timing of
qfx_attn_fwd / _bwd_dq / _bwd_dkv at S = 2432 and 8576 (24 heads x 128) and bit-comparison of every output with the product's.
    python tools/attn_var_bench.py base,pp1,pp2 [--entry fwd,dq,dkv]"""
import argparse, ctypes as C, json, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from step_ab import load_variant
from qflux_amd import ops, _lib as L
ap = argparse.ArgumentParser(); ap.add_argument("variants"); ap.add_argument("--entry", default="fwd,dq,dkv"); ap.add_argument("--S", default="2432,8576")
ap.add_argument("--hl", type=int, default=0, help="rank of a fused out-projection down projection (qfx_head_lora slot 0, forward only); names ending in '-' run without it")
ap.add_argument("--qk", action="store_true", help="fused QK-norm / RoPE backward epilogues (qk_saved) in dq / dkv; names ending in '-' run without it")
args = ap.parse_args()
names = args.variants.split(",")
libs = {n: load_variant(n.rstrip("-")) for n in names}
BF, DEV = torch.bfloat16, "cuda:0"
out = {}
for Sspec in args.S.split(","):      # "S" or "S:H" (heads; 24 by default)
    S, H = (int(x) for x in (Sspec.split(":") + ["24"])[:2])
    Bn, dh = 1, 128
    D = H * dh; S_pad = (S + 63) // 64 * 64
    torch.manual_seed(S)
    qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF)
    ld = 3 * D
    dO = torch.randn(Bn, S, D, device=DEV).to(BF)
    bufs = {}
    for n in names:
        O = torch.zeros(Bn, S, D, dtype=BF, device=DEV); lse2 = torch.zeros(Bn, H, S_pad, device=DEV); dsum = torch.zeros(Bn, H, S_pad, device=DEV)
        dqkv = torch.zeros_like(qkv)
        a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=qkv[:, :, :D], K=qkv[:, :, D:2 * D], V=qkv[:, :, 2 * D:], ldq=ld, ldk=ld, ldv=ld,
                          O=O, ldo=D, lse2=lse2, dsum=dsum, dO=dO, lddo=D, dQ=dqkv[:, :, :D], dK=dqkv[:, :, D:2 * D], dV=dqkv[:, :, 2 * D:],
                          lddq=ld, lddk=ld, lddv=ld)
        if args.hl and not n.endswith("-"):
            R = args.hl
            wts = [(torch.randn(R, D, device=DEV) * 0.1).to(BF) for _ in range(4)]
            part = torch.zeros(H, Bn * S, R, device=DEV)
            a.T = 384
            hl = a.hl[0]
            hl.part, hl.part_hstride, hl.ld_part, hl.c0, hl.R = part.data_ptr(), Bn * S * R, R, 0, R
            wts = [L.head_fragment_image(wts[0], wts[1], dh), L.head_fragment_image(wts[2], wts[3], dh)]
            hl.w_pk[0], hl.w_pk[1] = (t.data_ptr() for t in wts)
            a._keep = (wts, part)
        if args.qk and not n.endswith("-"):
            sqk = torch.randn(Bn, S, 2 * D, device=DEV).to(BF)
            ang = torch.rand(S, dh // 2, device=DEV) * 6.28
            rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
            ws = [(1 + 0.1 * torch.randn(dh, device=DEV)).to(BF) for _ in range(4)]
            a.qk_saved, a.ld_saved, a.rope, a.rope_bstride = sqk.data_ptr(), 2 * D, rope.data_ptr(), 0
            a.wq_txt, a.wk_txt, a.wq_img, a.wk_img = (t.data_ptr() for t in ws)
            a.T, a.norm_flags, a.norm_eps = 384, 0, 1e-6
            a._keep2 = (sqk, rope, ws)
        bufs[n] = (a, O, lse2, dsum, dqkv)
    st = torch.cuda.current_stream().cuda_stream
    for ent in args.entry.split(","):
        sym = {"fwd": "qfx_attn_fwd", "dq": "qfx_attn_bwd_dq", "dkv": "qfx_attn_bwd_dkv"}[ent]
        res = {n: [] for n in names}
        for rnd in range(6):
            for n in names:
                fn = getattr(libs[n], sym); a = bufs[n][0]
                for _ in range(2): assert fn(C.byref(a), st) == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): fn(C.byref(a), st)
                e1.record(); torch.cuda.synchronize()
                if rnd: res[n].append(e0.elapsed_time(e1) / 10 * 1e3)
        fl = {"fwd": 4.0, "dq": 6.0, "dkv": 8.0}[ent] * S * S * D
        for n in names:
            med = sorted(res[n])[len(res[n]) // 2]
            eq = all(torch.equal(x, y) for x, y in zip(bufs[n][1:], bufs[names[0]][1:])) if ent == "dkv" or len(args.entry.split(",")) == 1 else None
            out[f"S{S}_{ent}_{n}"] = dict(us=round(med, 1), tflops=round(fl / med / 1e6, 0))
    for n in names[1:]:
        out[f"S{S}_equal_{n}"] = {k: bool(torch.equal(x, y)) for k, x, y in zip(("O", "lse2", "dsum", "dqkv"), bufs[n][1:], bufs[names[0]][1:])}
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "attn_var_bench.json"), "w"), indent=1)
