import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "lab.so"))
lib.lab_gemm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
BF = torch.bfloat16
variants = [tuple(int(y) for y in x.split(':')) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0:0','1:0','6:0','1:8','6:8','6:4'])]
shapes = [(2432, 3072, 3072), (2432, 12288, 3072), (2432, 3072, 12288), (4096, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(BF); b = torch.randn(N, K, device="cuda").to(BF)
    ref = (a @ b.t()).float()
    line = f"{M}x{N}x{K}:"
    for (v, gm) in variants:
        c = torch.zeros(M, N, dtype=BF, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.lab_gemm(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, gm, st)
        torch.cuda.synchronize()
        err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
        for _ in range(3): lib.lab_gemm(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, gm, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.lab_gemm(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, gm, st)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        line += f"  v{v}g{gm}: {2*M*N*K/t/1e12:6.0f}TF{'!' if err > 2e-2 or rc else ''}"
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    for _ in range(3): torch.matmul(a, b.t())
    t0.record()
    for _ in range(20): torch.matmul(a, b.t())
    t1.record(); torch.cuda.synchronize()
    t = t0.elapsed_time(t1) / 20 * 1e-3
    line += f"  hipblaslt: {2*M*N*K/t/1e12:6.0f}TF"
    print(line, flush=True)
