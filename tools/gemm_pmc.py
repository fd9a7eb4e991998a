#!/usr/bin/env python
"""Runs the persistent GEMM on the DiT's narrow / wide shapes with cold weights -- target command of rocprofv3 counter passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops, _lib as L
BF, dev = torch.bfloat16, "cuda"
M = 2432
for N, K, epi in ((3072, 3072, 0), (3072, 12288, 0), (12288, 3072, 0)):
    Ws = [(torch.randn(N, K, device=dev) * 0.02).to(BF) for _ in range(6)]
    x = torch.randn(M, K, device=dev).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev)
    for i in range(12):
        ops.gemm(x, Ws[i % 6], out=y, epi=epi)
torch.cuda.synchronize(); print("ok")
