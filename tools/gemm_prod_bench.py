"""Production-GEMM micro-benchmark through the C ABI: plain vs grouped vs epilogues (warm, paired, best of 3)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops, _lib as L
BF = torch.bfloat16
dev = "cuda"
w = torch.randn(8192, 8192, device=dev).to(BF)
for _ in range(40): w @ w
torch.cuda.synchronize()
def bench(fn, flops, n=30):
    best = 0
    for _ in range(3):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = max(best, flops / (e0.elapsed_time(e1) / n * 1e-3) / 1e12)
    return best
Mi, Mt, D, H = 2048, 384, 3072, 12288
xi = torch.randn(Mi + Mt, D, device=dev).to(BF)
W1 = [torch.randn(H, D, device=dev).to(BF) * 0.02 for _ in range(4)]
W2 = [torch.randn(D, H, device=dev).to(BF) * 0.02 for _ in range(4)]
b1 = torch.randn(H, device=dev).to(BF)
h = torch.empty(Mi + Mt, H, dtype=BF, device=dev); gh = torch.empty_like(h)
y = torch.empty(Mi + Mt, D, dtype=BF, device=dev)
gate = torch.randn(1, D, device=dev).to(BF)
F1 = 2 * (Mi + Mt) * D * H
print("fc1 plain single 2432 rows       %.0f" % bench(lambda: ops.gemm(xi, W1[0], out=h), F1))
print("fc1 +bias                        %.0f" % bench(lambda: ops.gemm(xi, W1[0], bias=b1, out=h), F1))
print("fc1 +bias +gelu(dual)            %.0f" % bench(lambda: ops.gemm(xi, W1[0], bias=b1, out=h, out2=gh, epi=L.EPI_GELU), F1))
def grouped(epi, cold):
    k = [0]
    def f():
        i = k[0] % 4 if cold else 0; k[0] += 1
        gs = []
        for (r0, r1, Wm) in ((0, Mi, W1[i]), (Mi, Mi + Mt, W1[(i + 1) % 4])):
            g = L.GemmArgs()
            a = xi[r0:r1]
            g.A1, g.B1, g.lda1, g.ldb1, g.K1 = a.data_ptr(), Wm.data_ptr(), D, D, D
            g.M, g.N = r1 - r0, H
            g.bias = b1.data_ptr()
            g.C, g.ldc = h[r0:r1].data_ptr(), H
            if epi == L.EPI_GELU: g.C2, g.ldc2 = gh[r0:r1].data_ptr(), H
            g.rows_per_batch = r1 - r0
            g.epi = epi
            gs.append(g)
        arr = (L.GemmArgs * 2)(*gs)
        L.check(ops.lib.qfx_gemm_grouped(arr, 2, ops.stream_ptr()), "g")
    return f
print("fc1 grouped img+txt plain        %.0f" % bench(grouped(L.EPI_NONE, False), F1))
print("fc1 grouped img+txt gelu         %.0f" % bench(grouped(L.EPI_GELU, False), F1))
print("fc1 grouped img+txt gelu coldW   %.0f" % bench(grouped(L.EPI_GELU, True), F1))
print("fc2 plain single                 %.0f" % bench(lambda: ops.gemm(h, W2[0], out=y), F1))
print("fc2 gate_res                     %.0f" % bench(lambda: ops.gemm(h, W2[0], out=y, aux=xi, gate=gate, epi=L.EPI_GATE_RES), F1))
print("dgelu (M x H, K=D)               %.0f" % bench(lambda: ops.gemm(xi, W1[0], out=gh, aux=h, epi=L.EPI_DGELU), F1))
