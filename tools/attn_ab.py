#!/usr/bin/env python
"""A/B of attention kernel variants on one box:  python tools/attn_ab.py <variant> [kernels] [--S 2432,8576]
`variant` = tools/_ab/libqfx_<variant>.so (tools/build_variants.py), compared with the product library, alternating, best of 4
rounds x 30 launches per kernel; outputs are compared too."""
import ctypes as C, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
KERNELS = ("qfx_attn_fwd", "qfx_attn_bwd_dq", "qfx_attn_bwd_dkv")
var = C.CDLL(os.path.join(ROOT, "tools", "_ab", f"libqfx_{sys.argv[1]}.so"))
for _n in KERNELS:
    getattr(var, _n).argtypes = [C.POINTER(L.AttnArgs), C.c_void_p]; getattr(var, _n).restype = C.c_int
which = [k for k in KERNELS if len(sys.argv) < 3 or sys.argv[2].startswith("--") or k.split("qfx_attn_")[1] in sys.argv[2].split(",")]
Ss = (2432, 8576)
for i, v in enumerate(sys.argv):
    if v == "--S":
        Ss = tuple(int(x) for x in sys.argv[i + 1].split(","))
BF = torch.bfloat16; DEV = "cuda:0"
for S in Ss:
    Bn, H, dh = 1, 24, 128; D = H * dh; S_pad = (S + 63) // 64 * 64
    qkv = (torch.randn(Bn, S, 3 * D, device=DEV) * 0.5).to(BF)
    dO = (torch.randn(Bn, S, D, device=DEV) * 0.5).to(BF)
    def args():
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
        return a, (O, lse, dq, ds)
    (a1, k1), (a2, k2) = args(), args()
    st = torch.cuda.current_stream().cuda_stream
    for lib, a_ in ((L.lib, a1), (var, a2)):
        for n in KERNELS:
            assert getattr(lib, n)(C.byref(a_), st) == 0
    torch.cuda.synchronize()
    rel = lambda x, y: ((x.float() - y.float()).abs().max() / x.float().abs().max()).item()
    print(f"S={S}: O rel diff {rel(k1[0], k2[0]):.2e}  dqkv rel diff {rel(k1[2], k2[2]):.2e}", flush=True)
    w = torch.randn(8192, 8192, device=DEV).to(BF)
    for _ in range(20): w @ w
    for n in which:
        best = {}
        for rep in range(4):
            for k, (lib, a) in {"base": (L.lib, a1), "var": (var, a2)}.items():
                fn = getattr(lib, n)
                for _ in range(5): assert fn(C.byref(a), st) == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30): fn(C.byref(a), st)
                e1.record(); torch.cuda.synchronize()
                best[k] = min(best.get(k, 1e9), e0.elapsed_time(e1) / 30 * 1e3)
        print(f"  {n:20s} base {best['base']:8.1f} us   {sys.argv[1]} {best['var']:8.1f} us   ({(best['var'] / best['base'] - 1) * 100:+.1f} %)", flush=True)
