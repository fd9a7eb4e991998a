#!/usr/bin/env python
"""In-situ A/B of kernel variants: replay the REAL launch programs of a Qwen DiT training step (forward + backward, the argument
structs the product builds) once per variant library, round-robin in one process, with a HIP-event pair around every launch.

    python tools/step_ab.py base,v1,v2 [--layers 6] [--reps 5] [--batch 1] [--res 512] [--only gemm]

A variant `name` is the shared library tools/_ab/libqfx_<name>.so (built by tools/build_variants.py from csrc/ with extra -D
flags); `base` is the product library.  Each C-ABI call of the program is redirected to the same symbol of the variant library,
so weights are cold (every block has its own), operands / epilogues / row maps are exactly the step's, and neighbours in the
stream are the real neighbours.  Output: per variant, summed time per launch class (entry point + GEMM shape/epilogue) in us per
DiT block, and the total.  Variants must be numerically interchangeable with base (same buffers are overwritten)."""
from __future__ import annotations

import argparse
import collections
import ctypes as C
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))


def load_variant(name):
    from qflux_amd import _lib as L
    if name == "base":
        return L.lib
    lib = C.CDLL(os.path.join(ROOT, "tools", "_ab", f"libqfx_{name}.so"))
    for sym, (res, args) in L.SYMBOLS.items():
        fn = getattr(lib, sym, None)      # a library built from an older revision lacks the newer entry points
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    return lib


def klass(name, args):
    from qflux_amd import _lib as L
    if name in ("qfx_gemm_bf16", "qfx_gemm_grouped"):
        a0 = args[0]
        obj = getattr(a0, "_obj", a0)
        gs = [obj] if isinstance(obj, L.GemmArgs) else [obj[i] for i in range(args[1])]
        g = gs[0]
        return f"gemm n={len(gs)} N={g.N} K={g.K1}+{g.K2} epi={g.epi} M={'+'.join(str(x.M) for x in gs[:2])}"
    if name.startswith("qfx_lora_down"):
        obj = getattr(args[0], "_obj", args[0])
        n = 1 if isinstance(obj, L.LoraDownArgs) else args[1]
        a = obj if n == 1 and isinstance(obj, L.LoraDownArgs) else obj[0]
        return f"lora_down n={n} R={a.R} K={a.K}"
    return name.replace("qfx_", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants")
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--only", default="", help="substring filter on the class names that are printed")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    names = args.variants.split(",")
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    with torch.device(dev):
        dit = QwenImageTransformer2DModel(num_layers=args.layers)
    with torch.no_grad():
        for n, p in dit.named_parameters():
            if p.ndim == 2:
                p.normal_(0.0, 0.02)
            elif "norm" in n:
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02)
    dit.add_adapter(LoraConfig(r=args.rank, lora_alpha=args.rank, init_lora_weights="gaussian"), "default",
                    generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        for n, p in dit.named_parameters():
            if "lora_B" in n:
                p.normal_(0.0, 0.01)
    os.environ["QFX_SIDE_GRADS"] = "0"      # one stream: every launch is timed on its own
    step = QwenLoraTrainStep(dit, lr=1e-4)
    B, side, T = args.batch, args.res // 16, 384
    emb = dict(image_latents=torch.randn(B, side * side, 64).half().to(dev), control_latents=torch.randn(B, side * side, 64).half().to(dev),
               prompt_embeds=(torch.randn(B, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, side, side)] * 2] * B)
    for _ in range(2):
        step.train_step(emb)
    torch.cuda.synchronize()
    plan = list(dit._plans.values())[0]
    libs = {n: load_variant(n) for n in names}
    st_obj = torch.cuda.current_stream()
    st = st_obj.cuda_stream
    # warm the chip (clocks / power state) before the first timed replay
    w = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    for _ in range(30):
        w @ w
    torch.cuda.synchronize()
    res = {n: collections.defaultdict(list) for n in names}
    totals = {n: [] for n in names}
    for rep in range(args.reps + 1):
        for vn in names:
            lib = libs[vn]
            evs = []
            dit.refresh_lora_operands()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(st_obj)
            for prog in (plan.fwd, plan.bwd):
                for ent in prog.calls:
                    fn, a = ent[0], ent[1]
                    if fn is None:
                        a()
                        continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st_obj)
                    rc = getattr(lib, fn.__name__)(*a, st)
                    e1.record(st_obj)
                    if rc != 0:
                        raise RuntimeError(f"{vn}: {fn.__name__} -> {rc}")
                    evs.append((klass(fn.__name__, a), e0, e1))
            t1.record(st_obj)
            torch.cuda.synchronize()
            step.zero_grad()
            if rep == 0:
                continue     # first round of every variant: code objects load, caches warm
            per = collections.defaultdict(float)
            for k, e0, e1 in evs:
                per[k] += e0.elapsed_time(e1) * 1e3
            for k, v in per.items():
                res[vn][k].append(v / args.layers)
            totals[vn].append(t0.elapsed_time(t1) * 1e3 / args.layers)
    keys = sorted(res[names[0]].keys(), key=lambda k: -statistics.median(res[names[0]][k]))
    out = {"layers": args.layers, "reps": args.reps, "unit": "us per DiT block (median over reps)", "variants": {}}
    print(f"{'class':64s} " + " ".join(f"{n:>10s}" for n in names))
    for k in keys:
        if args.only and args.only not in k:
            continue
        print(f"{k[:64]:64s} " + " ".join(f"{statistics.median(res[n][k]):10.1f}" for n in names))
    print(f"{'TOTAL (wall incl. event overhead)':64s} " + " ".join(f"{statistics.median(totals[n]):10.1f}" for n in names))
    print(f"{'SUM of launches':64s} " + " ".join(f"{sum(statistics.median(v) for v in res[n].values()):10.1f}" for n in names))
    for n in names:
        out["variants"][n] = {"total": statistics.median(totals[n]), "sum": sum(statistics.median(v) for v in res[n].values()),
                              "classes": {k: statistics.median(v) for k, v in res[n].items()}}
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
