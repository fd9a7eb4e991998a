#!/usr/bin/env python
"""Phase timing of the forward attention kernel (s_memtime at the phase boundaries, -DQFX_ATTN_TIMING build in tools/_ab):
cycles per tile spent in QK^T, softmax + PV, and the tile barrier, per wave of the first blocks."""
import ctypes as C, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
var = C.CDLL(os.path.join(ROOT, "tools", "_ab", "libqfx_timing.so"))
var.qfx_attn_fwd.argtypes = [C.POINTER(L.AttnArgs), C.c_void_p]; var.qfx_attn_fwd.restype = C.c_int
BF = torch.bfloat16; DEV = "cuda:0"
for S in (2432, 8576):
    Bn, H, dh = 1, 24, 128; D = H * dh; S_pad = (S + 63) // 64 * 64
    qkv = (torch.randn(Bn, S, 3 * D, device=DEV) * 0.5).to(BF)
    a = L.AttnArgs()
    a.B, a.S, a.S_pad, a.H, a.dh, a.scale = Bn, S, S_pad, H, dh, 1 / math.sqrt(dh)
    a.Q, a.K, a.V = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
    a.ldq = a.ldk = a.ldv = 3 * D
    O = torch.empty(Bn, S, D, dtype=BF, device=DEV); lse = torch.zeros(Bn * H * S_pad + 16 * 8 * 4, device=DEV)
    a.O, a.ldo, a.lse2 = O.data_ptr(), D, lse.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert var.qfx_attn_fwd(C.byref(a), st) == 0
    torch.cuda.synchronize()
    d = lse[Bn * H * S_pad:].view(16, 8, 4).cpu()
    nt = d[0, 0, 3].item()
    print(f"S={S} tiles={nt}: per-tile cycles (mean over waves of 16 blocks)  QK {d[:, :, 0].mean().item() / nt:.0f}  softmax+PV {d[:, :, 1].mean().item() / nt:.0f}"
          f"  barrier {d[:, :, 2].mean().item() / nt:.0f}   per wave QK {[round(x / nt) for x in d[0, :, 0].tolist()]} PV {[round(x / nt) for x in d[0, :, 1].tolist()]} bar {[round(x / nt) for x in d[0, :, 2].tolist()]}")
