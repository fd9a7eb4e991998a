#!/usr/bin/env python
"""Round 6 (VERDICT r5 #2a): the production GEMM against the vendor library (torch.matmul = hipBLASLt bf16) AT HEAD, alternating on
one box, on the nine shapes of profiles/r02_gemm_vs_hipblaslt.json plus the step's six-problem q/k/v launch (image 2048 + text 384
rows x {q, k, v}, N = K = 3072, one qfx_gemm_grouped launch; the library runs it as ONE 2432 x 9216 x 3072 matmul and as six).
Uniform random normal operands (cdna_hip_programming.md 5.4 rule 25), best AND median of 7 rounds x 20 launches, interleaved."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops
DEV, BF = "cuda:0", torch.bfloat16
SHAPES = [(2048, 3072, 3072), (2432, 3072, 3072), (2432, 9216, 3072), (2432, 12288, 3072), (2432, 3072, 12288), (2432, 3072, 9216),
          (4864, 3072, 3072), (4864, 12288, 3072), (8192, 8192, 8192)]
ROUNDS = 7


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def ab(fns):
    t = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            t[k].append(bench(f))
    return {k: (min(v), sorted(v)[len(v) // 2]) for k, v in t.items()}


res = {}
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=DEV).to(BF)
    b = torch.randn(N, K, device=DEV).to(BF)
    o1 = torch.empty(M, N, dtype=BF, device=DEV)
    o2 = torch.empty(M, N, dtype=BF, device=DEV)
    bt = b.t()
    r = ab({"qfx": lambda: ops.gemm(a, b, out=o1), "lib": lambda: torch.matmul(a, bt, out=o2)})
    fl = 2.0 * M * N * K
    d = ((o1.float() - o2.float()).abs().max() / o2.float().abs().max()).item()
    res[f"{M}x{N}x{K}"] = {"qfx_us": r["qfx"][0], "lib_us": r["lib"][0], "qfx_us_median": r["qfx"][1], "lib_us_median": r["lib"][1],
                           "qfx_TFs": fl / r["qfx"][0] / 1e6, "lib_TFs": fl / r["lib"][0] / 1e6, "qfx_over_lib": r["lib"][0] / r["qfx"][0],
                           "rel_diff": d}
    print(f"{M}x{N}x{K}: qfx {r['qfx'][0]:.1f} us ({fl / r['qfx'][0] / 1e6:.0f} TF/s)   library {r['lib'][0]:.1f} us ({fl / r['lib'][0] / 1e6:.0f} TF/s)   "
          f"speed-up {r['lib'][0] / r['qfx'][0]:.3f}   rel diff {d:.1e}", flush=True)

# ---- the six-problem q/k/v launch of the step (grouped: image rows + text rows, three weight sections)
D = 3072
xi = torch.randn(2048, D, device=DEV).to(BF)
xt = torch.randn(384, D, device=DEV).to(BF)
W = [torch.randn(D, D, device=DEV).to(BF) for _ in range(6)]          # q, k, v of the image stream, then of the text stream
out = torch.empty(2432, 3 * D, dtype=BF, device=DEV)
groups, libs = [], []
for s_, (x, r0) in enumerate(((xi, 384), (xt, 0))):
    for sec in range(3):
        o = out[r0:r0 + x.shape[0], sec * D:(sec + 1) * D]
        groups.append((x, W[3 * s_ + sec], o, {}))
        libs.append((x, W[3 * s_ + sec].t(), torch.empty(x.shape[0], D, dtype=BF, device=DEV)))
xall = torch.cat([xt, xi], 0)
Wi = torch.cat(W[0:3], 0)
o_lib1 = torch.empty(2432, 3 * D, dtype=BF, device=DEV)


def lib6():
    for x, wt, o in libs:
        torch.matmul(x, wt, out=o)


r = ab({"qfx_grouped6": lambda: ops.gemm_grouped(groups), "lib_one_2432x9216": lambda: torch.matmul(xall, Wi.t(), out=o_lib1), "lib_six_calls": lib6})
fl = 2.0 * 2432 * 9216 * 3072
res["qkv_six_problem_launch"] = {k: {"us": v[0], "us_median": v[1], "TFs": fl / v[0] / 1e6} for k, v in r.items()}
print("qkv six-problem:", {k: (round(v[0], 1), round(fl / v[0] / 1e6)) for k, v in r.items()}, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"unit": "us per launch, best (and median) of 7 interleaved rounds x 20; TF/s = 2MNK / best time; operands randn", "shapes_MxNxK": res},
          open(os.path.join(ROOT, "gpurun_out", "r06_gemm_vs_hipblaslt.json"), "w"), indent=1)
