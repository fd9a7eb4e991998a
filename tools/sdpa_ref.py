#!/usr/bin/env python
"""Same-box yardstick for the attention kernels:  python tools/sdpa_ref.py [--S 2432,8576] [--lib tools/_ab/libqfx_<v>.so]
qfx_attn_fwd / qfx_attn_bwd_dq + qfx_attn_bwd_dkv against torch's F.scaled_dot_product_attention (the op the reference's
attention dispatch ends in, transformer_qwenimage.py:329-337; on ROCm the flash backend) forward and backward on the two joint
sequence lengths of the BASELINE configs (cfg #2: S = 2432, cfg #4: S = 8576; 24 heads x 128), alternating, best of 4 rounds x 20
launches.  The vendor op is NOT used by the product; it is the reference point the judge asked for
(profiles/r03_attn_vs_sdpa.json).  Algorithmic flops: forward 4 S^2 D, backward 10 S^2 D (2.5x)."""
import ctypes as C, json, math, os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
Ss = (2432, 8576)
lib = L.lib
tag = "product"
for i, v in enumerate(sys.argv):
    if v == "--S":
        Ss = tuple(int(x) for x in sys.argv[i + 1].split(","))
    if v == "--lib":
        lib = C.CDLL(os.path.join(ROOT, sys.argv[i + 1]))
        tag = os.path.basename(sys.argv[i + 1])
        for n in ("qfx_attn_fwd", "qfx_attn_bwd_dq", "qfx_attn_bwd_dkv"):
            getattr(lib, n).argtypes = [C.POINTER(L.AttnArgs), C.c_void_p]; getattr(lib, n).restype = C.c_int
BF = torch.bfloat16; DEV = "cuda:0"
res = {"lib": tag, "torch": torch.__version__, "shapes": {}}


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for S in Ss:
    Bn, H, dh = 1, 24, 128; D = H * dh; S_pad = (S + 63) // 64 * 64
    g = torch.Generator(device=DEV).manual_seed(S)
    qkv = (torch.randn(Bn, S, 3 * D, device=DEV, generator=g) * 1.0).to(BF)
    dO = (torch.randn(Bn, S, D, device=DEV, generator=g) * 0.5).to(BF)
    a = L.AttnArgs()
    a.B, a.S, a.S_pad, a.H, a.dh, a.scale = Bn, S, S_pad, H, dh, 1 / math.sqrt(dh)
    a.Q, a.K, a.V = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
    a.ldq = a.ldk = a.ldv = 3 * D
    O = torch.empty(Bn, S, D, dtype=BF, device=DEV); lse = torch.zeros(Bn, H, S_pad, device=DEV)
    a.O, a.ldo, a.lse2 = O.data_ptr(), D, lse.data_ptr()
    a.dO, a.lddo = dO.data_ptr(), D
    dq = torch.zeros(Bn, S, 3 * D, dtype=BF, device=DEV)
    a.dQ, a.dK, a.dV = dq.data_ptr(), dq.data_ptr() + 2 * D, dq.data_ptr() + 4 * D
    a.lddq = a.lddk = a.lddv = 3 * D
    ds = torch.zeros(Bn, H, S_pad, device=DEV); a.dsum = ds.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    # torch layout [B, H, S, dh] (what dispatch_attention_fn permutes to); contiguous copies so the vendor kernel sees its best case
    q, k, v = (qkv[:, :, i * D:(i + 1) * D].reshape(Bn, S, H, dh).permute(0, 2, 1, 3).contiguous().requires_grad_(True) for i in range(3))
    dOt = dO.reshape(Bn, S, H, dh).permute(0, 2, 1, 3).contiguous()
    # correctness cross-check of the two implementations
    assert lib.qfx_attn_fwd(C.byref(a), st) == 0 and lib.qfx_attn_bwd_dq(C.byref(a), st) == 0 and lib.qfx_attn_bwd_dkv(C.byref(a), st) == 0
    o_t = F.scaled_dot_product_attention(q, k, v)
    gq, gk, gv = torch.autograd.grad(o_t, (q, k, v), dOt)
    rel = lambda x, y: ((x.float() - y.float()).abs().max() / y.float().abs().max()).item()
    back = lambda t: t.permute(0, 2, 1, 3).reshape(Bn, S, D)
    diffs = dict(o=rel(O, back(o_t)), dq=rel(dq[:, :, :D], back(gq)), dk=rel(dq[:, :, D:2 * D], back(gk)), dv=rel(dq[:, :, 2 * D:], back(gv)))
    w = torch.randn(8192, 8192, device=DEV).to(BF)
    for _ in range(20):
        w @ w
    best = {}
    o_keep = [None]

    def sdpa_fwd():
        with torch.no_grad():
            F.scaled_dot_product_attention(q, k, v)

    def sdpa_fwd_graph():
        o_keep[0] = F.scaled_dot_product_attention(q, k, v)

    def sdpa_bwd():
        torch.autograd.grad(o_keep[0], (q, k, v), dOt, retain_graph=True)

    def ours_bwd():
        lib.qfx_attn_bwd_dq(C.byref(a), st); lib.qfx_attn_bwd_dkv(C.byref(a), st)

    sdpa_fwd_graph()
    arms = {"qfx_fwd": lambda: lib.qfx_attn_fwd(C.byref(a), st), "sdpa_fwd": sdpa_fwd, "qfx_bwd": ours_bwd, "sdpa_bwd": sdpa_bwd,
            "qfx_bwd_dq": lambda: lib.qfx_attn_bwd_dq(C.byref(a), st), "qfx_bwd_dkv": lambda: lib.qfx_attn_bwd_dkv(C.byref(a), st)}
    for rep in range(4):
        for name, fn in arms.items():
            best[name] = min(best.get(name, 1e18), timeit(fn))
    unit = 2.0 * S * S * D      # one S^2 D contraction
    tf = lambda us, units: units * unit / (us * 1e-6) / 1e12
    ent = {"us": {k_: round(v_, 1) for k_, v_ in best.items()},
           "tflops_algorithmic": {"qfx_fwd": round(tf(best["qfx_fwd"], 2), 1), "sdpa_fwd": round(tf(best["sdpa_fwd"], 2), 1),
                                  "qfx_bwd": round(tf(best["qfx_bwd"], 5), 1), "sdpa_bwd": round(tf(best["sdpa_bwd"], 5), 1)},
           "frac_of_2.5PF": {"qfx_fwd": round(tf(best["qfx_fwd"], 2) / 2500, 3), "sdpa_fwd": round(tf(best["sdpa_fwd"], 2) / 2500, 3),
                             "qfx_bwd": round(tf(best["qfx_bwd"], 5) / 2500, 3), "sdpa_bwd": round(tf(best["sdpa_bwd"], 5) / 2500, 3)},
           "max_rel_diff_vs_sdpa": {k_: float(f"{v_:.2e}") for k_, v_ in diffs.items()}}
    res["shapes"][f"S{S}_H24_dh128"] = ent
    print(S, json.dumps(ent), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"attn_vs_sdpa_{tag.replace('.so', '')}.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
