#!/usr/bin/env python
"""Run-to-run reproducibility of the fused attention-backward epilogues, per variant library (tools/build_variants.py):
    python tools/nondet_bisect.py base,pk1,pk1f0,... [--S 8576] [--reps 40] [--entry dq,dkv] [--hl 16]
For every variant the entry point is launched `reps` times on identical inputs into a re-zeroed output; outputs are compared bit for
bit with the first launch.  Reports how many launches differ, and for the differing ones which 16-row fragments (row block, head)
differ, by how much, and whether whole fragments or single rows are affected -> gpurun_out/nondet_bisect.json."""
import argparse, collections, ctypes as C, json, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from step_ab import load_variant
from qflux_amd import ops, _lib as L
ap = argparse.ArgumentParser(); ap.add_argument("variants"); ap.add_argument("--S", type=int, default=8576); ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--entry", default="dq,dkv"); ap.add_argument("--hl", type=int, default=16); ap.add_argument("--out", default="nondet_bisect.json")
ap.add_argument("--busy", type=int, default=0, help="1 = keep a second stream busy with a GEMM-like torch matmul while the launches run")
args = ap.parse_args()
names = args.variants.split(",")
BF, DEV = torch.bfloat16, "cuda:0"
S, H, Bn, dh = args.S, 24, 1, 128
D = H * dh; S_pad = (S + 63) // 64 * 64
torch.manual_seed(S)
qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF); ld = 3 * D
dO = torch.randn(Bn, S, D, device=DEV).to(BF)
sqk = torch.randn(Bn, S, 2 * D, device=DEV).to(BF)
ang = torch.rand(S, dh // 2, device=DEV) * 6.28
rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
ws = [(1 + 0.1 * torch.randn(dh, device=DEV)).to(BF) for _ in range(4)]
R = args.hl
wts_raw = [[(torch.randn(R, D, device=DEV) * 0.1).to(BF) for _ in range(2)] for _ in range(4)] if R else None
st = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
out = {}
for n in names:
    lib = load_variant(n)
    O = torch.zeros(Bn, S, D, dtype=BF, device=DEV); lse2 = torch.zeros(Bn, H, S_pad, device=DEV); dsum = torch.zeros(Bn, H, S_pad, device=DEV)
    dqkv = torch.zeros_like(qkv)
    a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=qkv[:, :, :D], K=qkv[:, :, D:2 * D], V=qkv[:, :, 2 * D:], ldq=ld, ldk=ld, ldv=ld,
                      O=O, ldo=D, lse2=lse2, dsum=dsum, dO=dO, lddo=D, dQ=dqkv[:, :, :D], dK=dqkv[:, :, D:2 * D], dV=dqkv[:, :, 2 * D:],
                      lddq=ld, lddk=ld, lddv=ld)
    assert lib.qfx_attn_fwd(C.byref(a), st) == 0
    a.qk_saved, a.ld_saved, a.rope, a.rope_bstride = sqk.data_ptr(), 2 * D, rope.data_ptr(), 0
    a.wq_txt, a.wk_txt, a.wq_img, a.wk_img = (t.data_ptr() for t in ws)
    a.T, a.norm_flags, a.norm_eps = 384, 0, 1e-6
    parts, keep = [], []
    if R:
        for slot in (1, 2, 3):
            part = torch.zeros(H, Bn * S, R, device=DEV); parts.append(part)
            hl = a.hl[slot]
            hl.part, hl.part_hstride, hl.ld_part, hl.c0, hl.R = part.data_ptr(), Bn * S * R, R, 0, R
            w = [L.head_fragment_image(wts_raw[slot][0], wts_raw[slot][1], dh), L.head_fragment_image(wts_raw[slot][0], wts_raw[slot][1], dh)]
            hl.w_pk[0], hl.w_pk[1] = (t.data_ptr() for t in w); keep.append(w)
    res = {}
    for ent in args.entry.split(","):
        fn = getattr(lib, {"dq": "qfx_attn_bwd_dq", "dkv": "qfx_attn_bwd_dkv"}[ent])
        ref, refp = None, None
        ndiff = 0; frag_hist = collections.Counter(); details = []
        for rep in range(args.reps):
            dqkv.zero_()
            for p in parts: p.zero_()
            if args.busy:
                with torch.cuda.stream(side):
                    x = torch.randn(4096, 4096, device=DEV, dtype=BF); y = x @ x
            assert fn(C.byref(a), st) == 0
            torch.cuda.synchronize()
            cur = dqkv.clone(); curp = [p.clone() for p in parts]
            if ref is None: ref, refp = cur, curp; continue
            if torch.equal(cur, ref) and all(torch.equal(x, y) for x, y in zip(curp, refp)): continue
            ndiff += 1
            d = (cur.view(torch.int16) != ref.view(torch.int16))[0]          # [S, 3D]
            rows = d.view(S, 3, H, dh).any(-1)                                  # [S, 3, H]
            idx = rows.nonzero().tolist()
            fr = collections.Counter((r // 16, c, h) for r, c, h in idx)
            for k, v in fr.items(): frag_hist[v] += 1                           # rows affected per 16-row fragment
            if len(details) < 2 and idx:
                # element-level pattern of the first differing fragment: which of the 128 head columns differ in how many of its 16 rows,
                # and the signed bf16-ulp difference of its first row
                r16, c, h = list(fr)[0]
                blk = slice(r16 * 16, r16 * 16 + 16); col = slice(c * D + h * dh, c * D + (h + 1) * dh)
                dm = d[blk, col]
                pat = dict(frag=[r16, c, h], elems=int(dm.sum()), per_col=dm.sum(0).tolist(), per_row=dm.sum(1).tolist(),
                           ulp=(cur[0, blk, col].view(torch.int16).int() - ref[0, blk, col].view(torch.int16).int())[0].tolist())
            else:
                pat = None
            if len(details) < 4:
                mx = (cur.float() - ref.float()).abs().max().item()
                rel = ((cur.float() - ref.float()).abs() / (ref.float().abs() + 1e-6))[d.unsqueeze(0)].median().item() if d.any() else 0.0
                details.append(dict(rep=rep, frags=len(fr), rows=len(idx), which="qkv"[idx[0][1]] if idx else "-", first=[list(k) for k in list(fr)[:6]], max_abs=mx, median_rel=rel,
                                    part_equal=[bool(torch.equal(x, y)) for x, y in zip(curp, refp)], pattern=pat))
        res[ent] = dict(differing_launches=ndiff, of=args.reps - 1, rows_per_differing_fragment=dict(frag_hist), details=details)
    out[n] = res
    print(n, json.dumps(res), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", args.out), "w"), indent=1)
