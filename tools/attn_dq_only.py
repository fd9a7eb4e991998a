#!/usr/bin/env python
"""qfx_attn_bwd_dq N times at S (target of rocprofv3 counter passes; QFX_ATTN_DQ64 selects the kernel)."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops
BF, DEV = torch.bfloat16, "cuda:0"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2432
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Bn, H, dh = 1, 24, 128
D = H * dh; S_pad = (S + 63) // 64 * 64
qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF); ld = 3 * D
O = torch.empty(Bn, S, D, dtype=BF, device=DEV); lse2 = torch.zeros(Bn, H, S_pad, device=DEV); dsum = torch.zeros(Bn, H, S_pad, device=DEV)
dO = torch.randn(Bn, S, D, device=DEV).to(BF); dqkv = torch.empty_like(qkv)
a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=qkv[:, :, :D], K=qkv[:, :, D:2 * D], V=qkv[:, :, 2 * D:], ldq=ld, ldk=ld, ldv=ld, O=O, ldo=D, lse2=lse2,
                  dsum=dsum, dO=dO, lddo=D, dQ=dqkv[:, :, :D], dK=dqkv[:, :, D:2 * D], dV=dqkv[:, :, 2 * D:], lddq=ld, lddk=ld, lddv=ld)
ops.attn_call("qfx_attn_fwd", a)
for _ in range(N): ops.attn_call("qfx_attn_bwd_dq", a)
torch.cuda.synchronize(); print("ok")
