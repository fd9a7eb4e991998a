import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "lab.so"))
lib.lab_gemm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
v, M, N, K, gm = (int(x) for x in sys.argv[1:6])
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
c = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(4):
    lib.lab_gemm(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, gm, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()

