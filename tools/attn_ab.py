import ctypes as C, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
old = C.CDLL(os.path.join(ROOT, "tools", "attn_old.so"))
for _n in ("qfx_attn_fwd", "qfx_attn_bwd_dq", "qfx_attn_bwd_dkv"):
    getattr(old, _n).argtypes = [C.POINTER(L.AttnArgs), C.c_void_p]; getattr(old, _n).restype = C.c_int
WHICH = sys.argv[1] if len(sys.argv) > 1 else "qfx_attn_fwd"
BF = torch.bfloat16; DEV = "cuda:0"
for S in (2432, 8576):
    Bn, H, dh = 1, 24, 128; D = H * dh; S_pad = (S + 63) // 64 * 64
    qkv = (torch.randn(Bn, S, 3 * D, device=DEV) * 0.5).to(BF)
    O1 = torch.empty(Bn, S, D, dtype=BF, device=DEV); O2 = torch.empty_like(O1)
    l1 = torch.zeros(Bn, H, S_pad, device=DEV); l2 = torch.zeros_like(l1)
    dO = (torch.randn(Bn, S, D, device=DEV) * 0.5).to(BF); keep = []
    def args(O, lse):
        a = L.AttnArgs()
        a.B, a.S, a.S_pad, a.H, a.dh, a.scale = Bn, S, S_pad, H, dh, 1 / math.sqrt(dh)
        a.Q, a.K, a.V = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
        a.ldq = a.ldk = a.ldv = 3 * D
        a.O, a.ldo, a.lse2 = O.data_ptr(), D, lse.data_ptr()
        a.dO, a.lddo = dO.data_ptr(), D
        dq = torch.zeros(Bn, S, 3 * D, dtype=BF, device=DEV); keep.append(dq)
        a.dQ, a.dK, a.dV = dq.data_ptr(), dq.data_ptr() + 2 * D, dq.data_ptr() + 4 * D
        a.lddq = a.lddk = a.lddv = 3 * D
        ds = torch.zeros(Bn, H, S_pad, device=DEV); keep.append(ds); a.dsum = ds.data_ptr()
        return a
    a1, a2 = args(O1, l1), args(O2, l2)
    st = torch.cuda.current_stream().cuda_stream
    for a_ in (a1, a2):   # forward + dQ first: the backward kernels need lse / dsum
        assert L.lib.qfx_attn_fwd(C.byref(a_), st) == 0 and L.lib.qfx_attn_bwd_dq(C.byref(a_), st) == 0
    fns = {"old": (getattr(old, WHICH), a1), "new": (getattr(L.lib, WHICH), a2)}
    w = torch.randn(8192, 8192, device=DEV).to(BF)
    for _ in range(20): w @ w
    best = {}
    for rep in range(4):
        for k, (fn, a) in fns.items():
            for _ in range(5): assert fn(C.byref(a), st) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): fn(C.byref(a), st)
            e1.record(); torch.cuda.synchronize()
            best[k] = min(best.get(k, 1e9), e0.elapsed_time(e1) / 30 * 1e3)
    d = ((O1.float() - O2.float()).abs().max() / O1.float().abs().max()).item()
    dg = ((keep[0].float() - keep[2].float()).abs().max() / keep[0].float().abs().max()).item()
    print(f"{WHICH} S={S}: old {best['old']:.1f} us  new {best['new']:.1f} us   O rel diff {d:.2e}  dqkv rel diff {dg:.2e}")
