#!/usr/bin/env python
"""MX-FP8 GEMM vs the bf16 production GEMM on the DiT shapes (cold weights ring, warm chip)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops
BF = torch.bfloat16
dev = "cuda"
w = torch.randn(8192, 8192, device=dev).to(BF)
for _ in range(40): w @ w
torch.cuda.synchronize()
def t_of(fn, n=20):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
out = {}
for (M, N, K) in ((2432, 12288, 3072), (2432, 3072, 12288), (2432, 3072, 3072), (2432, 9216, 3072), (8192, 8192, 8192)):
    nW = max(2, min(6, int(400e6 / (N * K * 2)) + 1))
    Ws = [(torch.randn(N, K, device=dev) * 0.02).to(BF) for _ in range(nW)]
    Wq = [ops.quant_mxfp8(x) for x in Ws]
    x = torch.randn(M, K, device=dev).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev)
    xq, xs = ops.quant_mxfp8(x)
    k = [0]
    def fb():
        ops.gemm(x, Ws[k[0] % nW], out=y); k[0] += 1
    def f8():
        q, s = Wq[k[0] % nW]; k[0] += 1
        ops.gemm_mxfp8(xq, xs, q, s, out=y)
    def fq():
        ops.quant_mxfp8(x, out=(xq, xs))
    tb, t8, tq = t_of(fb), t_of(f8), t_of(fq)
    fl = 2 * M * N * K
    out[f"{M}x{N}x{K}"] = {"bf16_us": round(tb, 1), "bf16_tflops": round(fl / tb / 1e6, 0), "mxfp8_us": round(t8, 1), "mxfp8_tflops": round(fl / t8 / 1e6, 0),
                           "quant_A_us": round(tq, 1)}
    del Ws, Wq
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fp8_bench.json"), "w"), indent=1)
