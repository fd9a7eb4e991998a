#!/usr/bin/env python
"""Per-launch FIXED cost of the production GEMM: time = a + b*K fitted over K for the DiT's two output widths and the four
epilogues, with COLD weights (a ring of distinct B operands larger than the Infinity Cache) and a dependent-launch chain like the
step's (each launch consumes the previous one's output row block).  Prints one JSON line."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops, _lib as L
BF = torch.bfloat16
dev = "cuda"
M = 2432
w = torch.randn(8192, 8192, device=dev).to(BF)
for _ in range(40): w @ w
torch.cuda.synchronize()
out = {}
def t_of(fn, n=24):
    for _ in range(6): fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
null = t_of(lambda: ops.gemm(w[:128, :64], w[:128, :64]), 40)
out["tiny_128x128x64_launch_us"] = round(null, 2)
for N in (3072, 12288):
    for epi, name in ((L.EPI_NONE, "none"), (L.EPI_GATE_RES, "gate_res"), (L.EPI_GELU, "gelu"), (L.EPI_DGELU, "dgelu")):
        if N == 12288 and epi == L.EPI_GATE_RES: continue
        pts = []
        for K in (64, 512, 1024, 3072, 6144, 12288):
            if N == 12288 and K > 3072: continue
            nW = max(2, min(8, int(400e6 / (N * K * 2)) + 1))     # > 256 MB of distinct weights where they are big enough to matter
            Ws = [(torch.randn(N, K, device=dev) * 0.02).to(BF) for _ in range(nW)]
            x = torch.randn(M, K, device=dev).to(BF)
            y = torch.empty(M, N, dtype=BF, device=dev); y2 = torch.empty_like(y)
            aux = torch.randn(M, N, device=dev).to(BF); gate = torch.randn(1, N, device=dev).to(BF)
            bias = torch.randn(N, device=dev).to(BF)
            k = [0]
            def f():
                Wm = Ws[k[0] % nW]; k[0] += 1
                kw = dict(bias=bias, out=y, epi=epi)
                if epi == L.EPI_GELU: kw["out2"] = y2
                if epi in (L.EPI_GATE_RES, L.EPI_DGELU): kw["aux"] = aux
                if epi == L.EPI_GATE_RES: kw["gate"] = gate
                ops.gemm(x, Wm, **kw)
            pts.append((K, t_of(f)))
            del Ws
        # least squares a + b K over the points with K >= 512
        xs = [p[0] for p in pts if p[0] >= 512]; ys = [p[1] for p in pts if p[0] >= 512]
        n = len(xs); sx, sy = sum(xs), sum(ys); sxx = sum(v * v for v in xs); sxy = sum(a * b for a, b in zip(xs, ys))
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        flops_per_k = 2 * M * N
        out[f"N{N}_{name}"] = {"points_us": {str(k_): round(t, 1) for k_, t in pts}, "fixed_us": round(a, 1), "us_per_ktile64": round(b * 64, 3),
                               "mainloop_tflops": round(flops_per_k / (b * 1e-6) / 1e12, 0)}
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_fixed.json"), "w"), indent=1)
