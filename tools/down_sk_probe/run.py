#!/usr/bin/env python
"""Split-K down-projection prototype (probe.hip) vs the product kernel: time and result.  Build: hipcc --offload-arch=gfx950 -O3
-shared -fPIC -o tools/down_sk_probe/probe.so tools/down_sk_probe/probe.hip"""
import ctypes as C, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
pl = C.CDLL(os.path.join(here, "probe.so"))
pl.run_down_sk.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
pl.run_down_sk.restype = C.c_int
DEV, BF = "cuda:0", torch.bfloat16
M = 2432
for K, R in ((3072, 16), (3072, 48), (12288, 16), (12288, 48)):
    nring = max(2, int(300e6 // (M * K * 2)) + 1)
    xs = [torch.randn(M, K, device=DEV).to(BF) for _ in range(nring)]
    A = torch.randn(R, K, device=DEV) * 0.05
    hi = A.to(BF); lo = (A - hi.float()).to(BF)
    U = torch.zeros(M, R, device=DEV); U2 = torch.zeros(M, R, device=DEV)
    ext = torch.zeros(M, 3 * R, dtype=BF, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    def prod(i):
        a = L.LoraDownArgs()
        a.X, a.ldx, a.M, a.K = xs[i % nring].data_ptr(), K, M, K
        a.W_hi, a.W_lo, a.ldw, a.R = hi.data_ptr(), lo.data_ptr(), K, R
        a.U, a.ldu, a.ext, a.ld_ext = U.data_ptr(), R, ext.data_ptr(), 3 * R
        a.group_R, a.group_stride, a.rows_per_batch = R, 3 * R, M
        assert L.lib.qfx_lora_down(C.byref(a), st) == 0
    def sk(i):
        assert pl.run_down_sk(xs[i % nring].data_ptr(), K, M, K, hi.data_ptr(), lo.data_ptr(), K, R, U2.data_ptr(), R, st) == 0
    res = {}
    for name, fn in (("product", prod), ("split-K", sk)):
        for i in range(4): fn(i)
        best = 1e9
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(4 * nring): fn(i)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / (4 * nring) * 1e3)
        res[name] = best
    U.zero_(); U2.zero_(); prod(0); sk(0); torch.cuda.synchronize()
    ref = xs[0].float() @ (hi.float() + lo.float()).t()
    e1 = ((U - ref).abs().max() / ref.abs().max()).item(); e2 = ((U2 - ref).abs().max() / ref.abs().max()).item()
    print(f"K={K} R={R}: product {res['product']:.1f} us, split-K main kernel {res['split-K']:.1f} us  (rel err {e1:.1e} / {e2:.1e})", flush=True)
