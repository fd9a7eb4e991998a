#!/usr/bin/env python
"""Phase timing of the 64-query forward (s_memtime at the phase boundaries, -DQFX_A64_TIMING build tools/_ab/libqfx_a64t.so):
cycles per tile and wave in the tile barrier (+ DMA wait), T1 (QK^T qb 0), T2 (QK^T qb 1 | softmax 0), T3 (PV 0 | softmax 1), T4 (PV 1)."""
import ctypes as C, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
var = C.CDLL(os.path.join(ROOT, "tools", "_ab", os.environ.get("A64T_LIB", "libqfx_a64t.so")))
var.qfx_attn_fwd.argtypes = [C.POINTER(L.AttnArgs), C.c_void_p]; var.qfx_attn_fwd.restype = C.c_int
var.qfx_attn_bwd_dq.argtypes = [C.POINTER(L.AttnArgs), C.c_void_p]; var.qfx_attn_bwd_dq.restype = C.c_int
os.environ["QFX_ATTN_FWD64"] = "1"; os.environ["QFX_ATTN_DQ64"] = "1"
BF = torch.bfloat16; DEV = "cuda:0"
for S in (2432, 8576):
    Bn, H, dh = 1, 24, 128; D = H * dh; S_pad = (S + 63) // 64 * 64
    qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF)
    a = L.AttnArgs()
    a.B, a.S, a.S_pad, a.H, a.dh, a.scale = Bn, S, S_pad, H, dh, 1 / math.sqrt(dh)
    a.Q, a.K, a.V = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
    a.ldq = a.ldk = a.ldv = 3 * D
    O = torch.empty(Bn, S, D, dtype=BF, device=DEV); lse = torch.zeros(Bn * H * S_pad + 16 * 4 * 8, device=DEV)
    a.O, a.ldo, a.lse2 = O.data_ptr(), D, lse.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): assert var.qfx_attn_fwd(C.byref(a), st) == 0
    torch.cuda.synchronize()
    d = lse[Bn * H * S_pad:].view(16, 4, 8).cpu()
    nt = d[0, 0, 6].item()
    names = ["barrier+dma", "T1", "T2", "T3", "T4", "loop rest"]
    live = d[:, :2]          # waves 0, 1 are live in every block
    print(f"S={S} tiles={nt:.0f}: cycles per tile and wave (mean over waves 0-1 of 16 blocks): " + "  ".join(f"{n} {live[:, :, i].mean().item() / nt:.0f}" for i, n in enumerate(names)) +
          f"   total {live[:, :, :6].sum(-1).mean().item() / nt:.0f}")
    print("   block 0 per wave:", [[round(x / nt) for x in d[0, w, :6].tolist()] for w in range(4)])
    # dQ
    dO = torch.randn(Bn, S, D, device=DEV).to(BF); dsum = torch.zeros(Bn * H * S_pad + 16 * 4 * 8, device=DEV); dqkv = torch.zeros_like(qkv)
    a.dO, a.lddo, a.dsum = dO.data_ptr(), D, dsum.data_ptr()
    a.dQ, a.dK, a.dV = dqkv.data_ptr(), dqkv.data_ptr() + 2 * D, dqkv.data_ptr() + 4 * D
    a.lddq = a.lddk = a.lddv = 3 * D
    for fused in (0, 1):
      if fused:      # the fused QK-norm + RoPE backward of the step's launches (epilogue cost)
        sqk = torch.randn(Bn, S, 2 * D, device=DEV).to(BF)
        ang = torch.rand(S, dh // 2, device=DEV) * 6.28
        rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
        ws = [(1 + 0.1 * torch.randn(dh, device=DEV)).to(BF) for _ in range(4)]
        a.qk_saved, a.ld_saved, a.rope, a.rope_bstride = sqk.data_ptr(), 2 * D, rope.data_ptr(), 0
        a.wq_txt, a.wk_txt, a.wq_img, a.wk_img = (t.data_ptr() for t in ws)
        a.T, a.norm_flags, a.norm_eps = 384, 0, 1e-6
      for _ in range(3): assert var.qfx_attn_bwd_dq(C.byref(a), st) == 0
      torch.cuda.synchronize()
      d = dsum[Bn * H * S_pad:].view(16, 4, 8).cpu()
      live = d[:, :2]
      print(f"S={S} dQ64 fused_qknorm={fused}: cycles per tile and wave: " + "  ".join(f"{n} {live[:, :, i].mean().item() / nt:.0f}" for i, n in enumerate(names[:5])) + f"   loop total {live[:, :, :5].sum(-1).mean().item() / nt:.0f}"
            f"   per wave: prologue {live[:, :, 7].mean().item():.0f}  loop {live[:, :, :5].sum(-1).mean().item():.0f}  epilogue {live[:, :, 5].mean().item():.0f} cycles")
