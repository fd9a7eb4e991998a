#!/usr/bin/env python
"""Per-kernel micro-benchmark at the headline shapes (Qwen-Image-Edit 512^2: S_i=2048, T=384, D=3072).
Writes gpurun_out/kbench.json.  TF/s = algorithmic flops / measured time (torch.cuda events on the
current stream, which is the stream the C ABI launches on)."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops  # noqa: E402

BF = torch.bfloat16
DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    res = {}
    torch.manual_seed(0)
    for (M, N, K) in [(2048, 3072, 3072), (2432, 3072, 3072), (2048, 12288, 3072), (2048, 3072, 12288), (2048, 9216, 3072),
                      (4096, 4096, 4096), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=DEV).to(BF)
        b = torch.randn(N, K, device=DEV).to(BF)
        out = torch.empty(M, N, dtype=BF, device=DEV)
        t = timeit(lambda: ops.gemm(a, b, out=out))
        res[f"gemm_{M}x{N}x{K}"] = dict(ms=t * 1e3, tflops=2 * M * N * K / t / 1e12)
        if (M, N, K) == (2048, 3072, 3072):
            t2 = timeit(lambda: torch.matmul(a, b.t()))
            res["hipblaslt_2048x3072x3072"] = dict(ms=t2 * 1e3, tflops=2 * M * N * K / t2 / 1e12)
        print(f"gemm {M}x{N}x{K}: {t*1e6:.1f} us  {2*M*N*K/t/1e12:.1f} TF/s", flush=True)
    # attention
    for S in (2432, 8576):
        Bn, H, dh = 1, 24, 128
        D = H * dh
        S_pad = (S + 63) // 64 * 64
        qkv = torch.randn(Bn, S, 3 * D, device=DEV).to(BF)
        Q, K_, V = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
        ld = 3 * D
        Vt = ops.transpose_heads(V, ld, Bn, S, S_pad, H, dh)
        Qt = ops.transpose_heads(Q, ld, Bn, S, S_pad, H, dh)
        Kt = ops.transpose_heads(K_, ld, Bn, S, S_pad, H, dh)
        O = torch.empty(Bn, S, D, dtype=BF, device=DEV)
        dO = torch.randn(Bn, S, D, device=DEV).to(BF)
        dOt = ops.transpose_heads(dO, D, Bn, S, S_pad, H, dh)
        lse2 = torch.zeros(Bn, H, S_pad, device=DEV)
        dsum = torch.zeros(Bn, H, S_pad, device=DEV)
        dqkv = torch.empty_like(qkv)
        a = ops.attn_args(Bn, S, S_pad, H, dh, 1 / math.sqrt(dh), Q=Q, K=K_, V=V, ldq=ld, ldk=ld, ldv=ld, Vt=Vt, Qt=Qt, Kt=Kt,
                          O=O, ldo=D, lse2=lse2, dsum=dsum, dO=dO, lddo=D, dOt=dOt,
                          dQ=dqkv[:, :, :D], dK=dqkv[:, :, D:2 * D], dV=dqkv[:, :, 2 * D:], lddq=ld, lddk=ld, lddv=ld)
        fl = 4.0 * S * S * D * Bn
        t = timeit(lambda: ops.attn_call("qfx_attn_fwd", a), iters=10)
        res[f"attn_fwd_S{S}"] = dict(ms=t * 1e3, tflops=fl / t / 1e12)
        ops.attn_call("qfx_attn_bwd_prep", a)
        t1 = timeit(lambda: ops.attn_call("qfx_attn_bwd_dq", a), iters=10)
        t2 = timeit(lambda: ops.attn_call("qfx_attn_bwd_dkv", a), iters=10)
        res[f"attn_bwd_dq_S{S}"] = dict(ms=t1 * 1e3, tflops=1.5 * fl / t1 / 1e12)
        res[f"attn_bwd_dkv_S{S}"] = dict(ms=t2 * 1e3, tflops=2.0 * fl / t2 / 1e12)
        tt = timeit(lambda: ops.transpose_heads(V, ld, Bn, S, S_pad, H, dh, out=Vt))
        res[f"transpose_S{S}"] = dict(ms=tt * 1e3, gbps=2 * S * D * 2 / tt / 1e9)
        print(f"attn S={S}: fwd {t*1e3:.2f} ms ({fl/t/1e12:.0f} TF/s) dq {t1*1e3:.2f} ms dkv {t2*1e3:.2f} ms", flush=True)
    # row kernels
    rows, D = 2432, 3072
    x = torch.randn(rows, D, device=DEV).to(BF)
    mod = torch.randn(1, 6 * D, device=DEV).to(BF)
    y = torch.empty_like(x)
    t = timeit(lambda: ops.ln_modulate_fwd(x, mod[:, :D], mod[:, D:2 * D], rows, out=y))
    res["ln_mod_fwd"] = dict(us=t * 1e6, gbps=2 * rows * D * 2 / t / 1e9)
    t = timeit(lambda: ops.ln_modulate_bwd(x, x, mod[:, D:2 * D], rows, dres=x, gate=mod[:, :D], want_dyg=True))
    res["ln_mod_bwd"] = dict(us=t * 1e6, gbps=5 * rows * D * 2 / t / 1e9)
    temb = torch.randn(1, D, device=DEV).to(BF)
    Ws = [torch.randn(6 * D, D, device=DEV).to(BF) for _ in range(8)]
    bs = [torch.randn(6 * D, device=DEV).to(BF) for _ in range(8)]
    t = timeit(lambda: ops.mod_gemv(temb, Ws, bs))
    res["mod_gemv_8x"] = dict(us=t * 1e6, gbps=8 * 6 * D * D * 2 / t / 1e9)
    # LoRA skinny
    M, K, R = 2048, 3072, 48
    xx = torch.randn(M, K, device=DEV).to(BF)
    wh = torch.randn(R, K, device=DEV).to(BF); wl = torch.randn(R, K, device=DEV).to(BF)
    U = torch.empty(M, R, device=DEV); ext = torch.zeros(M, 192, dtype=BF, device=DEV)
    t = timeit(lambda: ops.lora_down(xx, wh, wl, U=U, ext=ext, group_R=16, group_stride=64))
    res["lora_down_R48"] = dict(us=t * 1e6, gbps=M * K * 2 / t / 1e9)
    G = torch.zeros(R, K, device=DEV)
    Mp = (M + 127) // 128 * 128
    Vt = (torch.randn(R, Mp, device=DEV).to(BF), torch.randn(R, Mp, device=DEV).to(BF))
    t = timeit(lambda: ops.lora_grad(Vt, xx, G, K, 1, M=M))
    res["lora_grad_R48"] = dict(us=t * 1e6, gbps=M * K * 2 / t / 1e9)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "kbench.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
