"""Sustained-clock check: same launch repeated for ~1 s, TF/s per 100-launch window.  python sustain.py lib:var[:grid] MxNxK [rotate]
rotate=1 cycles through 8 different weight matrices (cold B, like the training step)."""
import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
p = sys.argv[1].split(':')
lib = ctypes.CDLL(os.path.join(here, p[0] + ".so"))
lib.lab_gemm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
var, grid = int(p[1]), (int(p[2]) if len(p) > 2 else 0)
M, N, K = (int(v) for v in sys.argv[2].split('x'))
rot = int(sys.argv[3]) if len(sys.argv) > 3 else 0
BF = torch.bfloat16
a = torch.randn(M, K, device="cuda").to(BF)
bs = [torch.randn(N, K, device="cuda").to(BF) for _ in range(8 if rot else 1)]
c = torch.zeros(M, N, dtype=BF, device="cuda")
st = torch.cuda.current_stream().cuda_stream
out = []
for win in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(400): lib.lab_gemm(var, a.data_ptr(), bs[i % len(bs)].data_ptr(), c.data_ptr(), M, N, K, grid, st)
    e1.record(); torch.cuda.synchronize()
    out.append(2 * M * N * K / (e0.elapsed_time(e1) / 400 * 1e-3) / 1e12)
print(sys.argv[1], sys.argv[2], "rot" if rot else "same", " ".join(f"{x:5.0f}" for x in out), flush=True)
