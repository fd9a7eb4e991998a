#!/usr/bin/env python
"""qfx_attn_bwd_dq: the 64-query kernel (QFX_ATTN_DQ64=1) against the 32-query kernel (=0) on one box -- outputs compared (dQ, dsum, fused
rank-r partial sums; plain and with the fused QK-norm / RoPE backward), both against an fp32 autograd reference of SDPA in the plain mode,
kernels timed interleaved.   python tools/attn64_dq_check.py [--S 2432,8576,333:2:2,...]"""
import argparse, ctypes as C, json, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops, _lib as L
ap = argparse.ArgumentParser(); ap.add_argument("--S", default="2432,8576,333:2:2,200:2:3,64:1:1,1000:4:1"); ap.add_argument("--hl", type=int, default=16)
ap.add_argument("--mask", type=int, default=0); ap.add_argument("--time", type=int, default=1); ap.add_argument("--out", default="attn64_dq_check.json")
args = ap.parse_args()
BF, DEV = torch.bfloat16, "cuda:0"
out = {}
def rel(x, y): return ((x.float() - y.float()).abs().max() / (y.float().abs().max() + 1e-12)).item()
for spec in args.S.split(","):
    S, H, Bn = (int(x) for x in (spec.split(":") + ["24", "1"])[:3])
    dh = 128; D = H * dh; S_pad = (S + 63) // 64 * 64
    torch.manual_seed(S)
    qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF); ld = 3 * D
    dO = torch.randn(Bn, S, D, device=DEV).to(BF)
    T = 48 if S > 64 else 16
    R = args.hl
    kmask = None
    if args.mask:
        kmask = torch.zeros(Bn, S, device=DEV); kmask[:, S - S // 5:] = -1e4 if args.mask == 1 else float("-inf")
    sqk = torch.randn(Bn, S, 2 * D, device=DEV).to(BF)
    ang = torch.rand(S, dh // 2, device=DEV) * 6.28
    rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
    ws = [(1 + 0.1 * torch.randn(dh, device=DEV)).to(BF) for _ in range(4)]
    wts = [(torch.randn(R, D, device=DEV) * 0.1).to(BF) for _ in range(2)] if R else None
    wpk = [L.head_fragment_image(wts[0], wts[1], dh), L.head_fragment_image(wts[1], wts[0], dh)] if R else None
    res = {}
    st = torch.cuda.current_stream().cuda_stream
    for fused in (0, 1):
        for mode in ("0", "1"):
            os.environ["QFX_ATTN_DQ64"] = mode
            O = torch.zeros(Bn, S, D, dtype=BF, device=DEV); lse2 = torch.zeros(Bn, H, S_pad, device=DEV); dsum = torch.zeros(Bn, H, S_pad, device=DEV)
            dqkv = torch.zeros_like(qkv); part = torch.zeros(H, Bn * S, max(R, 1), device=DEV)
            a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=qkv[:, :, :D], K=qkv[:, :, D:2 * D], V=qkv[:, :, 2 * D:], ldq=ld, ldk=ld, ldv=ld,
                              O=O, ldo=D, lse2=lse2, dsum=dsum, dO=dO, lddo=D, dQ=dqkv[:, :, :D], dK=dqkv[:, :, D:2 * D], dV=dqkv[:, :, 2 * D:],
                              lddq=ld, lddk=ld, lddv=ld)
            if kmask is not None: a.key_mask = kmask.data_ptr()
            a.T = T
            L.check(L.lib.qfx_attn_fwd(C.byref(a), st), "fwd")
            if fused:
                a.qk_saved, a.ld_saved, a.rope, a.rope_bstride = sqk.data_ptr(), 2 * D, rope.data_ptr(), 0
                a.wq_txt, a.wk_txt, a.wq_img, a.wk_img = (t.data_ptr() for t in ws)
                a.norm_flags, a.norm_eps = 0, 1e-6
                if R:
                    hl = a.hl[1]
                    hl.part, hl.part_hstride, hl.ld_part, hl.c0, hl.R = part.data_ptr(), Bn * S * R, R, 0, R
                    hl.w_pk[0], hl.w_pk[1] = wpk[0].data_ptr(), wpk[1].data_ptr()
            L.check(L.lib.qfx_attn_bwd_dq(C.byref(a), st), "dq"); torch.cuda.synchronize()
            us = None
            if args.time and fused:
                ts = []
                for rnd in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10): L.lib.qfx_attn_bwd_dq(C.byref(a), st)
                    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
                us = sorted(ts)[len(ts) // 2]
            res[(fused, mode)] = (dqkv[:, :, :D].float().clone(), dsum[:, :, :S].clone(), part.clone(), us)
    # fp32 reference of the plain mode
    q, k, v = (qkv[:, :, i * D:(i + 1) * D].float().view(Bn, S, H, dh).transpose(1, 2).detach().requires_grad_(i == 0) for i in range(3))
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if kmask is not None: sc = sc + kmask[:, None, None, :]
    o = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(Bn, S, D)
    o.backward(dO.float())
    dq_ref = q.grad.transpose(1, 2).reshape(Bn, S, D)
    r = dict(us_old=res[(1, "0")][3], us_new=res[(1, "1")][3],
             plain_dq_new_vs_ref=rel(res[(0, "1")][0], dq_ref), plain_dq_old_vs_ref=rel(res[(0, "0")][0], dq_ref),
             plain_dsum_new_vs_old=rel(res[(0, "1")][1], res[(0, "0")][1]),
             fused_dq_new_vs_old=rel(res[(1, "1")][0], res[(1, "0")][0]), fused_part_new_vs_old=rel(res[(1, "1")][2], res[(1, "0")][2]),
             finite=bool(torch.isfinite(res[(1, "1")][0]).all() and torch.isfinite(res[(0, "1")][0]).all()))
    out[spec] = r
    print(spec, json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", args.out), "w"), indent=1)
