#!/usr/bin/env python
"""The production GEMM against the vendor library (torch.matmul = hipBLASLt / rocBLAS bf16) on the shapes of the headline step,
same operands, alternating on one box, best of 5 rounds x 20 launches.  Plain C = A B^T, no epilogue / LoRA segment (the library
has neither); M = 2432 = the joint rows of one sample (the step launches image and text rows as two grouped problems)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops
DEV, BF = "cuda:0", torch.bfloat16
SHAPES = [(2048, 3072, 3072), (2432, 3072, 3072), (2432, 9216, 3072), (2432, 12288, 3072), (2432, 3072, 12288), (2432, 3072, 9216),
          (4864, 3072, 3072), (4864, 12288, 3072), (8192, 8192, 8192)]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


res = {}
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=DEV).to(BF)
    b = torch.randn(N, K, device=DEV).to(BF)
    o1 = torch.empty(M, N, dtype=BF, device=DEV)
    o2 = torch.empty(M, N, dtype=BF, device=DEV)
    bt = b.t()
    best = {"qfx": 1e9, "lib": 1e9}
    for _ in range(5):
        best["qfx"] = min(best["qfx"], bench(lambda: ops.gemm(a, b, out=o1)))
        best["lib"] = min(best["lib"], bench(lambda: torch.matmul(a, bt, out=o2)))
    fl = 2.0 * M * N * K
    d = ((o1.float() - o2.float()).abs().max() / o2.float().abs().max()).item()
    res[f"{M}x{N}x{K}"] = {"qfx_us": best["qfx"], "lib_us": best["lib"], "qfx_TFs": fl / best["qfx"] / 1e6, "lib_TFs": fl / best["lib"] / 1e6,
                           "qfx_over_lib": best["lib"] / best["qfx"], "rel_diff": d}
    print(f"{M}x{N}x{K}: qfx {best['qfx']:.1f} us ({fl / best['qfx'] / 1e6:.0f} TF/s)   library {best['lib']:.1f} us ({fl / best['lib'] / 1e6:.0f} TF/s)   "
          f"speed-up {best['lib'] / best['qfx']:.3f}   rel diff {d:.1e}", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"unit": "us per launch, best of 5 x 20; TF/s = 2MNK / time", "shapes_MxNxK": res}, open(os.path.join(ROOT, "gpurun_out", "gemm_vs_hipblaslt.json"), "w"), indent=1)
