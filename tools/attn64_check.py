#!/usr/bin/env python
"""qfx_attn_fwd: the 64-query / 32x32x16 kernel (QFX_ATTN_FWD64=1) against the 32-query kernels (=0) and an fp32 reference on one box:
outputs compared (O, lse2, fused rank-r partial sums), kernels timed interleaved.   python tools/attn64_check.py [--S 2432,8576,...]"""
import argparse, ctypes as C, json, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops, _lib as L
ap = argparse.ArgumentParser(); ap.add_argument("--S", default="2432,8576,333:2:2,200:2:3,64:1:1,1000:4:1"); ap.add_argument("--hl", type=int, default=16)
ap.add_argument("--mask", type=int, default=0); ap.add_argument("--time", type=int, default=1); ap.add_argument("--new", default="1", help="QFX_ATTN_FWD64 value of the new kernel: 1 = sub-tile pipeline, 1s = skewed query blocks")
args = ap.parse_args()
BF, DEV = torch.bfloat16, "cuda:0"
out = {}
for spec in args.S.split(","):
    S, H, Bn = (int(x) for x in (spec.split(":") + ["24", "1"])[:3])
    dh = 128; D = H * dh; S_pad = (S + 63) // 64 * 64
    torch.manual_seed(S)
    qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF); ld = 3 * D
    T = 48 if S > 64 else 16
    R = args.hl
    wts = [(torch.randn(R, D, device=DEV) * 0.1).to(BF) for _ in range(4)]
    wpk = [L.head_fragment_image(wts[0], wts[1], dh), L.head_fragment_image(wts[2], wts[3], dh)]
    kmask = None
    if args.mask:
        kmask = torch.zeros(Bn, S, device=DEV); kmask[:, S - S // 5:] = -1e4 if args.mask == 1 else float("-inf")
    res = {}
    for mode in ("0", "1"):
        os.environ["QFX_ATTN_FWD64"] = mode if mode == "0" else args.new
        O = torch.zeros(Bn, S, D, dtype=BF, device=DEV); lse2 = torch.zeros(Bn, H, S_pad, device=DEV)
        part = torch.zeros(H, Bn * S, R, device=DEV)
        a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=qkv[:, :, :D], K=qkv[:, :, D:2 * D], V=qkv[:, :, 2 * D:], ldq=ld, ldk=ld, ldv=ld,
                          O=O, ldo=D, lse2=lse2)
        if kmask is not None: a.key_mask = kmask.data_ptr()
        a.T = T
        hl = a.hl[0]
        hl.part, hl.part_hstride, hl.ld_part, hl.c0, hl.R = part.data_ptr(), Bn * S * R, R, 0, R
        hl.w_pk[0], hl.w_pk[1] = wpk[0].data_ptr(), wpk[1].data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        L.check(L.lib.qfx_attn_fwd(C.byref(a), st), "fwd"); torch.cuda.synchronize()
        us = None
        if args.time:
            ts = []
            for rnd in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): L.lib.qfx_attn_fwd(C.byref(a), st)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
            us = sorted(ts)[len(ts) // 2]
        res[mode] = (O.float(), lse2[:, :, :S].clone(), part.clone(), us)
    q, k, v = (qkv[:, :, i * D:(i + 1) * D].float().view(Bn, S, H, dh).transpose(1, 2) for i in range(3))
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if kmask is not None: sc = sc + kmask[:, None, None, :]
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(Bn, S, D)
    lse_ref = torch.logsumexp(sc, -1) * 1.4426950408889634
    def rel(x, y): return ((x - y).abs().max() / (y.abs().max() + 1e-12)).item()
    finite = bool(torch.isfinite(res["1"][0]).all())
    out[spec] = dict(us_old=res["0"][3], us_new=res["1"][3], O_new_vs_ref=rel(res["1"][0], ref), O_old_vs_ref=rel(res["0"][0], ref),
                     lse_new_vs_ref=rel(res["1"][1], lse_ref), lse_old_vs_ref=rel(res["0"][1], lse_ref),
                     part_new_vs_old=rel(res["1"][2], res["0"][2]), O_new_vs_old=rel(res["1"][0], res["0"][0]), finite=finite)
    print(spec, json.dumps(out[spec]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "attn64_check.json"), "w"), indent=1)
