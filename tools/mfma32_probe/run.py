#!/usr/bin/env python
"""Lane mapping of v_mfma_f32_32x32x16_bf16 (gfx950), checked against a host product:
   A operand: lane l holds row l%32, k = 8*(l/32)..+7;  B operand: lane l holds column l%32, the same k;
   D: lane l holds column l%32, register i -> row 8*(i/4) + 4*(l/32) + i%4."""
import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "probe.so"))
lib.run_probe.argtypes = [C.c_void_p] * 4; lib.run_probe.restype = C.c_int
A = torch.randn(32, 16).to(torch.bfloat16).cuda(); B = torch.randn(32, 16).to(torch.bfloat16).cuda()
out = torch.zeros(64, 16, device="cuda")
assert lib.run_probe(A.data_ptr(), B.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
ref = A.float() @ B.float().t()          # [row][col]
got = torch.zeros(32, 32)
o = out.cpu()
for l in range(64):
    for i in range(16):
        got[8 * (i // 4) + 4 * (l // 32) + i % 4, l % 32] = o[l, i]
err = (got - ref.cpu()).abs().max().item()
print("mfma 32x32x16 bf16 lane-map check: max |diff| =", err)
assert err < 1e-3
