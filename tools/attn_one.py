import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops
BF = torch.bfloat16; DEV = "cuda:0"
which = sys.argv[1]; S = int(sys.argv[2])
Bn, H, dh = 1, 24, 128; D = H * dh; S_pad = (S + 63) // 64 * 64
qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF)
Q, K_, V = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
ld = 3 * D
Vt = ops.transpose_heads(V, ld, Bn, S, S_pad, H, dh); Qt = ops.transpose_heads(Q, ld, Bn, S, S_pad, H, dh); Kt = ops.transpose_heads(K_, ld, Bn, S, S_pad, H, dh)
O = torch.empty(Bn, S, D, dtype=BF, device=DEV); dO = torch.randn(Bn, S, D, device=DEV).to(BF)
dOt = ops.transpose_heads(dO, D, Bn, S, S_pad, H, dh)
lse2 = torch.zeros(Bn, H, S_pad, device=DEV); dsum = torch.zeros(Bn, H, S_pad, device=DEV); dqkv = torch.empty_like(qkv)
a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=Q, K=K_, V=V, ldq=ld, ldk=ld, ldv=ld, Vt=Vt, Qt=Qt, Kt=Kt, O=O, ldo=D,
                  lse2=lse2, dsum=dsum, dO=dO, lddo=D, dOt=dOt, dQ=dqkv[:, :, :D], dK=dqkv[:, :, D:2 * D], dV=dqkv[:, :, 2 * D:],
                  lddq=ld, lddk=ld, lddv=ld)
ops.attn_call("qfx_attn_fwd", a); ops.attn_call("qfx_attn_bwd_prep", a)
for _ in range(3):
    ops.attn_call(which, a)
torch.cuda.synchronize()
