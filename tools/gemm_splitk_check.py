#!/usr/bin/env python
"""Same-XCD split-K (256x256 tiles, two work items per tile) against the unsplit launch and an fp32 reference, on the step's launch
shapes: image + text groups of one grouped launch, every epilogue, with and without the LoRA K extension; timed interleaved.
    python tools/gemm_splitk_check.py [--out gpurun_out/gemm_splitk.json]"""
import argparse, ctypes as C, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops, _lib as L
ap = argparse.ArgumentParser(); ap.add_argument("--out", default=""); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--bias", default=""); ap.add_argument("--lib", default="base"); ap.add_argument("--cases", type=int, default=99)
args = ap.parse_args()
sys.path.insert(0, os.path.join(ROOT, "tools"))
from step_ab import load_variant
lib = load_variant(args.lib)
BF, DEV = torch.bfloat16, "cuda:0"
torch.manual_seed(0)

def tune(s):
    assert lib.qfx_gemm_tune(s.encode(), None) == 0, s

def problem(Ms, N, K1, K2, epi):
    """one grouped launch: a group per entry of Ms (own weights), shared epilogue kind"""
    gs, keep, refs = [], [], []
    for M in Ms:
        a1 = (torch.randn(M, K1, device=DEV) * 0.5).to(BF); b1 = (torch.randn(N, K1, device=DEV) * 0.05).to(BF)
        bias = (torch.randn(N, device=DEV) * 0.1).to(BF)
        out = torch.zeros(M, N, dtype=BF, device=DEV)
        g = L.GemmArgs()
        g.A1, g.B1, g.lda1, g.ldb1, g.K1 = a1.data_ptr(), b1.data_ptr(), K1, K1, K1
        g.M, g.N, g.bias, g.C, g.ldc, g.rows_per_batch, g.epi = M, N, bias.data_ptr(), out.data_ptr(), N, M, epi
        ref = a1.float() @ b1.float().t() + bias.float()
        if K2:
            a2 = (torch.randn(M, K2, device=DEV) * 0.5).to(BF); b2 = (torch.randn(N, K2, device=DEV) * 0.05).to(BF)
            g.A2, g.B2, g.lda2, g.ldb2, g.K2 = a2.data_ptr(), b2.data_ptr(), K2, K2, K2
            ref = ref.to(BF).float() + a2.float() @ b2.float().t()           # base output rounded to bf16 before the LoRA add
            keep += [a2, b2]
        extra = {}
        if epi == L.EPI_GELU:
            out2 = torch.zeros(M, N, dtype=BF, device=DEV); g.C2, g.ldc2 = out2.data_ptr(), N; extra["out2"] = out2
            refs.append((ref.to(BF).float(), torch.nn.functional.gelu(ref.to(BF).float(), approximate="tanh")))
        elif epi == L.EPI_GATE_RES:
            gate = (torch.randn(1, N, device=DEV)).to(BF); aux = torch.randn(M, N, device=DEV).to(BF)
            g.gate, g.gate_bstride, g.aux, g.ldaux = gate.data_ptr(), N, aux.data_ptr(), N
            keep += [gate, aux]
            y = ref.to(BF).float()
            refs.append((aux.float() + (gate.float() * y).to(BF).float(),))
        elif epi == L.EPI_DGELU:
            aux = torch.randn(M, N, device=DEV).to(BF); g.aux, g.ldaux = aux.data_ptr(), N; keep.append(aux)
            h = aux.float(); t = torch.tanh(0.7978845608 * (h + 0.044715 * h ** 3))
            dg = 0.5 * (1 + t) + 0.5 * h * (1 - t * t) * 0.7978845608 * (1 + 3 * 0.044715 * h * h)
            refs.append((ref.to(BF).float() * dg,))
        else:
            refs.append((ref,))
        keep += [a1, b1, bias, out]; gs.append((g, out, extra))
    arr = (L.GemmArgs * len(gs))(*[g for g, _, _ in gs])
    return arr, gs, refs, keep

st = torch.cuda.current_stream().cuda_stream
res = {}
cases = [("N3072 K12288 plain", [2048, 384], 3072, 12288, 0, L.EPI_NONE), ("N3072 K12288 gate_res", [2048, 384], 3072, 12288, 0, L.EPI_GATE_RES),
         ("N3072 K9216+192", [2048, 384], 3072, 9216, 192, L.EPI_NONE), ("N3072 K12288+64 gelu", [2048, 384], 3072, 12288, 64, L.EPI_GELU),
         ("N3072 K12288 dgelu ragged M", [2000, 300], 3072, 12288, 0, L.EPI_DGELU)]
if args.bias: tune("splitk_bias=" + args.bias)
for name, Ms, N, K1, K2, epi in cases[:args.cases]:
    arr, gs, refs, keep = problem(Ms, N, K1, K2, epi)
    outs = {}
    for mode in ("0", "1"):
        tune("splitk=" + mode)
        for g, out, ex in gs: out.zero_()
        assert lib.qfx_gemm_grouped(arr, len(gs), st) == 0
        torch.cuda.synchronize()
        outs[mode] = [out.clone() for _, out, _ in gs] + [ex["out2"].clone() for _, _, ex in gs if "out2" in ex]
    err = {}
    for mode in ("0", "1"):
        e = 0.0
        for (g, out, ex), ref, o in zip(gs, refs, outs[mode]):
            e = max(e, ((o.float() - ref[0]).abs().max() / ref[0].abs().max()).item())
        err[mode] = e
    d01 = max(((a.float() - b.float()).abs().max() / a.float().abs().max()).item() for a, b in zip(outs["0"], outs["1"]))
    t = {}
    for rnd in range(4):
        for mode in ("0", "1"):
            tune("splitk=" + mode)
            for _ in range(3): lib.qfx_gemm_grouped(arr, len(gs), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps): lib.qfx_gemm_grouped(arr, len(gs), st)
            e1.record(); torch.cuda.synchronize()
            if rnd: t.setdefault(mode, []).append(e0.elapsed_time(e1) / args.reps * 1e3)
    # run-to-run reproducibility of the split form
    tune("splitk=1")
    lib.qfx_gemm_grouped(arr, len(gs), st); torch.cuda.synchronize(); a = [out.clone() for _, out, _ in gs]
    lib.qfx_gemm_grouped(arr, len(gs), st); torch.cuda.synchronize(); rep = all(torch.equal(x, out) for x, (_, out, _) in zip(a, gs))
    res[name] = dict(us_unsplit=round(sorted(t["0"])[1], 1), us_split=round(sorted(t["1"])[1], 1), err_unsplit=err["0"], err_split=err["1"], split_vs_unsplit=d01, reproducible=rep)
    print(name, res[name], flush=True)
tune("splitk=1")
if args.out: json.dump(res, open(args.out, "w"), indent=1)
