#!/usr/bin/env python
"""Rank-r down projection over long rows (K = 12288: feed-forward adapters) vs the product library of a variant build:
python tools/down_k_bench.py [variant]  -- single-adapter launches, M = 2432, r = 16, inputs rotated through > 256 MB."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
DEV, BF = "cuda:0", torch.bfloat16
libs = {"product": L.lib}
if len(sys.argv) > 1:
    v = C.CDLL(os.path.join(ROOT, "tools", "_ab", f"libqfx_{sys.argv[1]}.so"))
    v.qfx_lora_down.argtypes = L.lib.qfx_lora_down.argtypes; v.qfx_lora_down.restype = C.c_int
    libs[sys.argv[1]] = v
for M, K, R in ((2432, 3072, 16), (2432, 3072, 48), (2432, 12288, 16)):
    nring = max(2, int(300e6 // (M * K * 2)) + 1)
    xs = [torch.randn(M, K, device=DEV).to(BF) for _ in range(nring)]
    A = torch.randn(R, K, device=DEV) * 0.05
    hi = A.to(BF); lo = (A - hi.float()).to(BF)
    U = torch.zeros(M, R, device=DEV)
    ext = torch.zeros(M, 3 * R, dtype=BF, device=DEV)
    def args(i):
        a = L.LoraDownArgs()
        a.X, a.ldx, a.M, a.K = xs[i % nring].data_ptr(), K, M, K
        a.W_hi, a.W_lo, a.ldw, a.R = hi.data_ptr(), lo.data_ptr(), K, R
        a.U, a.ldu, a.ext, a.ld_ext = U.data_ptr(), R, ext.data_ptr(), 3 * R
        a.group_R, a.group_stride, a.rows_per_batch = R, 3 * R, M
        return a
    al = [args(i) for i in range(nring)]
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for name, lib in libs.items():
        for i in range(4): assert lib.qfx_lora_down(C.byref(al[i % nring]), st) == 0
        best = 1e9
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(4 * nring): lib.qfx_lora_down(C.byref(al[i % nring]), st)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / (4 * nring) * 1e3)
        U.zero_(); lib.qfx_lora_down(C.byref(al[0]), st); torch.cuda.synchronize(); outs[name] = U.clone()
        print(f"M={M} K={K} R={R} {name:8s} {best:7.1f} us  ({M * K * 2 / best / 1e6:.2f} TB/s)", flush=True)
    ref = xs[0].float() @ (hi.float() + lo.float()).t()
    for name, u in outs.items():
        print(f"   {name}: max rel err vs fp32 {((u - ref).abs().max() / ref.abs().max()).item():.2e}")
