"""Reference point only (NOT used by the product): rocBLAS/hipBLASLt bf16 GEMM through torch.matmul on the DiT shapes."""
import torch
BF = torch.bfloat16
w = torch.randn(8192, 8192, device="cuda").to(BF)
for _ in range(40): w @ w
torch.cuda.synchronize()
for (M, N, K) in [(2432, 12288, 3072), (2432, 3072, 12288), (2432, 3072, 3072), (2432, 9216, 3072), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda").to(BF); b = torch.randn(N, K, device="cuda").to(BF)
    best = 0
    for _ in range(3):
        for _ in range(5): torch.matmul(a, b.t())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): torch.matmul(a, b.t())
        e1.record(); torch.cuda.synchronize()
        best = max(best, 2 * M * N * K / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12)
    print(f"{M}x{N}x{K}: torch.matmul (hipBLASLt) {best:6.0f} TF/s", flush=True)
