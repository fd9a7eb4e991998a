import ctypes as C, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
old = C.CDLL(os.path.join(ROOT, "tools", "attn_old.so"))
old.qfx_attn_fwd.argtypes = [C.POINTER(L.AttnArgs), C.c_void_p]; old.qfx_attn_fwd.restype = C.c_int
BF = torch.bfloat16; DEV = "cuda:0"
for S in (2432, 8576):
    Bn, H, dh = 1, 24, 128; D = H * dh; S_pad = (S + 63) // 64 * 64
    qkv = (torch.randn(Bn, S, 3 * D, device=DEV) * 0.5).to(BF)
    O1 = torch.empty(Bn, S, D, dtype=BF, device=DEV); O2 = torch.empty_like(O1)
    l1 = torch.zeros(Bn, H, S_pad, device=DEV); l2 = torch.zeros_like(l1)
    def args(O, lse):
        a = L.AttnArgs()
        a.B, a.S, a.S_pad, a.H, a.dh, a.scale = Bn, S, S_pad, H, dh, 1 / math.sqrt(dh)
        a.Q, a.K, a.V = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
        a.ldq = a.ldk = a.ldv = 3 * D
        a.O, a.ldo, a.lse2 = O.data_ptr(), D, lse.data_ptr()
        return a
    a1, a2 = args(O1, l1), args(O2, l2)
    st = torch.cuda.current_stream().cuda_stream
    fns = {"old": (old.qfx_attn_fwd, a1), "new": (L.lib.qfx_attn_fwd, a2)}
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
    dl = (l1 - l2).abs().max().item()
    print(f"S={S}: old {best['old']:.1f} us  new {best['new']:.1f} us   O rel diff {d:.2e}  lse diff {dl:.2e}")
