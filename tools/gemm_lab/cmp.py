"""Paired comparison of lab kernels: python cmp.py lib:var[:grid],... [shapes]  (warm GPU, round-robin, best of 3)."""
import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
libs = {}
def get(so):
    if so not in libs:
        l = ctypes.CDLL(os.path.join(here, so + ".so"))
        l.lab_gemm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        libs[so] = l
    return libs[so]
BF = torch.bfloat16
vs = []
for x in sys.argv[1].split(','):
    p = x.split(':'); vs.append((p[0], int(p[1]), int(p[2]) if len(p) > 2 else 0))
shapes = [(2432, 12288, 3072), (2432, 3072, 12288), (2432, 3072, 3072), (2432, 9216, 3072), (8192, 8192, 8192)]
if len(sys.argv) > 2: shapes = [tuple(int(v) for v in s.split('x')) for s in sys.argv[2].split(',')]
w = torch.randn(8192, 8192, device="cuda").to(BF)
for _ in range(40): w @ w
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(BF); b = torch.randn(N, K, device="cuda").to(BF)
    ref = (a @ b.t()).float()
    c = torch.zeros(M, N, dtype=BF, device="cuda")
    best = {}; bad = {}
    for rep in range(3):
        for v in vs:
            lib = get(v[0])
            if rep == 0:
                c.zero_()
                rc = lib.lab_gemm(v[1], a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, v[2], st)
                torch.cuda.synchronize()
                bad[v] = rc != 0 or ((c.float() - ref).abs().max() / ref.abs().max()).item() > 2e-2
            for _ in range(5): lib.lab_gemm(v[1], a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, v[2], st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): lib.lab_gemm(v[1], a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, v[2], st)
            e1.record(); torch.cuda.synchronize()
            tf = 2 * M * N * K / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12
            best[v] = max(best.get(v, 0), tf)
    print(f"{M}x{N}x{K}: " + "  ".join(f"{v[0]}{v[1]}/{v[2]}:{best[v]:5.0f}{'!' if bad[v] else ''}" for v in vs), flush=True)
