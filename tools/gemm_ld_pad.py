#!/usr/bin/env python
"""Does the row stride of the GEMM operands matter (L2 / fabric channel mapping)?  The product GEMM on the DiT shapes with A and / or B
stored at a padded leading dimension (K + pad elements), cold weights (4 copies cycled), interleaved, medians.
    python tools/gemm_ld_pad.py -> gpurun_out/gemm_ld_pad.json"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops
BF = torch.bfloat16; dev = "cuda"
M = 2432
out = {}
for (N, K) in ((3072, 12288), (12288, 3072), (3072, 3072), (3072, 9216)):
    pads = (0, 64, 128, 192, 256)
    A = {p: torch.randn(M, K + p, device=dev).to(BF)[:, :K] for p in pads}
    Bs = {p: [(torch.randn(N, K + p, device=dev) * 0.02).to(BF)[:, :K] for _ in range(4)] for p in pads}
    C = torch.empty(M, N, dtype=BF, device=dev)
    cfgs = [("A0_B0", 0, 0)] + [(f"A{p}_B0", p, 0) for p in pads[1:3]] + [(f"A0_B{p}", 0, p) for p in pads[1:]] + [(f"A{p}_B{p}", p, p) for p in pads[1:3]]
    res = {n: [] for n, _, _ in cfgs}
    k = 0
    for rnd in range(7):
        for n, pa, pb in cfgs:
            for _ in range(2):
                ops.gemm(A[pa], Bs[pb][k % 4], out=C); k += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm(A[pa], Bs[pb][k % 4], out=C); k += 1
            e1.record(); torch.cuda.synchronize()
            if rnd: res[n].append(e0.elapsed_time(e1) / 8 * 1e3)
    out[f"N{N}_K{K}"] = {n: round(sorted(v)[len(v) // 2], 1) for n, v in res.items()}
    print(f"N{N}_K{K}", out[f"N{N}_K{K}"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_ld_pad.json"), "w"), indent=1)
