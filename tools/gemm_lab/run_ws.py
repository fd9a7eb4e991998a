import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = sys.argv[2] if len(sys.argv) > 2 else "ws.so"
lib = ctypes.CDLL(os.path.join(here, so))
lib.lab_gemm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
BF = torch.bfloat16
variants = [tuple(int(y) for y in x.split(':')) if ':' in x else (int(x), 0) for x in sys.argv[1].split(',')]
shapes = [(2432, 12288, 3072), (2432, 3072, 12288), (2432, 3072, 3072), (2432, 9216, 3072), (8192, 8192, 8192)]
if len(sys.argv) > 3: shapes = [tuple(int(v) for v in s.split('x')) for s in sys.argv[3].split(',')]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(BF); b = torch.randn(N, K, device="cuda").to(BF)
    ref = (a @ b.t()).float()
    line = f"{M}x{N}x{K}:"
    for (v, grid) in variants:
        c = torch.zeros(M, N, dtype=BF, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.lab_gemm(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, grid, st)
        torch.cuda.synchronize()
        err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
        for _ in range(3): lib.lab_gemm(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, grid, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.lab_gemm(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, grid, st)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        line += f"  v{v}/{grid}: {2*M*N*K/t/1e12:6.0f}{'!' if (err > 2e-2 or rc) else ''}"
    print(line, flush=True)
