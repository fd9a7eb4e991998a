#!/usr/bin/env python
"""rocprofv3 PMC passes -> profiles/rNN_pmc_hbm.json  (the file bench.py reads `roofline.traffic` from).

Collect (two SEPARATE passes; FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"; never
combine --pmc with --sys-trace / --hip-trace: gpurun refuses that):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_FETCH_SIZE -o p -- \
        python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_WRITE_SIZE -o p -- \
        python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline

Convert:

    python tools/pmc_to_json.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -o profiles/r02_pmc_hbm.json

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are KB per dispatch, derived from the L2's fabric-side
request counters (Infinity-Cache hits are counted, not excluded); on gfx950 FETCH_SIZE reports exactly half the bytes of wide
coalesced reads (16 B per lane, global_load and LDS-DMA alike) -> x2; WRITE_SIZE is uncalibrated and taken as is.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import re
import sys

KERNELS = ("gemm256_kernel", "gemm_kernel", "ln_mod_fwd_kernel", "ln_mod_bwd_kernel", "ln_down_kernel", "mod_grad_kernel", "gemm_fp8_kernel", "quant_mxfp8_kernel", "lora_down_kernel",
           "lora_grad_kernel", "lora_grad_reduce_kernel", "lora_head_reduce_kernel", "attn_fwd_kernel", "attn_fwd64_kernel", "attn_fwd64p_kernel", "attn_bwd_dq_kernel", "attn_bwd_dq64_kernel", "attn_bwd_dkv_kernel", "attn_bwd1_kernel", "attn_dq_finish_kernel", "attn_prep_kernel", "attn_bwd_kernel", "qk_norm_rope_kernel",
           "mod_gemv_kernel")


_NAME_RE = re.compile(r"(?:^|[\s:])(" + "|".join(sorted(KERNELS, key=len, reverse=True)) + r")\s*[<(]")


def short_name(full: str):
    """'void (anonymous namespace)::gemm256_kernel<0, 128>(...)' -> 'gemm256_kernel' (template arguments folded together)."""
    m = _NAME_RE.search(full)
    return m.group(1) if m else None


def read_pass(folder: str, counter: str):
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise FileNotFoundError(f"no *counter_collection.csv under {folder}")
    acc = {}
    for path in files:
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short_name(row["Kernel_Name"])
                if k not in KERNELS:
                    continue
                rec = acc.setdefault(k, [0, 0.0])
                rec[0] += 1
                rec[1] += float(row["Counter_Value"])
    return acc


def mfma_summary(folder, out, cmd):
    """--pmc MfmaUtil VALUBusy pass -> per kernel INSTANTIATION averages (derived metrics use the gfx94x formulas: ROCm 7.2 ships none
    for gfx950).  python tools/pmc_to_json.py --mfma gpurun_out/pmc_mfma -o profiles/r02_pmc_mfma.json"""
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for path in files:
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short_name(row["Kernel_Name"])
                if k is None:
                    continue
                m = re.search(re.escape(k) + r"(<[^(]*>)?", row["Kernel_Name"])
                name = m.group(0) if m else k
                rec = acc.setdefault(name, {})
                c = rec.setdefault(row["Counter_Name"], [0, 0.0])
                c[0] += 1
                c[1] += float(row["Counter_Value"])
    res = {"source": f"rocprofv3 --pmc MfmaUtil VALUBusy --kernel-trace -- {cmd} (own pass; kernels are serialised by the counter collection)",
           "generator": "tools/pmc_to_json.py --mfma", "kernels": {}}
    for name, rec in sorted(acc.items()):
        res["kernels"][name] = {"launches": max(v[0] for v in rec.values()),
                                **{cn + "_pct": round(v[1] / v[0], 2) for cn, v in rec.items()}}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res["kernels"].items() if "gemm" in k or "attn" in k}))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("fetch_dir", nargs="?")
    ap.add_argument("write_dir", nargs="?")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--mfma", default=None, help="folder of a --pmc MfmaUtil VALUBusy pass: write the MFMA-utilisation summary instead")
    ap.add_argument("--cmd", default="python bench.py --steps 2 --warmup 1 --no-cpu-baseline")
    args = ap.parse_args(argv)
    if args.mfma:
        return mfma_summary(args.mfma, args.out, args.cmd)
    fetch = read_pass(args.fetch_dir, "FETCH_SIZE")
    write = read_pass(args.write_dir, "WRITE_SIZE")
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- {args.cmd}",
           "generator": "tools/pmc_to_json.py",
           "units": "FETCH_SIZE/WRITE_SIZE are KB per dispatch; gfx950 correction: FETCH_SIZE x2 for wide coalesced reads "
                    "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted",
           "kernels": {}}
    for k, (n, kb) in fetch.items():
        wn, wkb = write.get(k, (0, 0.0))
        f_raw = kb / n
        f_corr = int(round(f_raw * 1024 * 2))
        w = int(round(wkb / wn * 1024)) if wn else 0
        out["kernels"][k] = {"launches": n, "fetch_kb_raw_per_launch": round(f_raw, 1), "fetch_bytes_corrected_per_launch": f_corr,
                             "write_bytes_per_launch": w, "traffic_bytes_per_launch": f_corr + w}
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v["traffic_bytes_per_launch"] for k, v in out["kernels"].items()}))


if __name__ == "__main__":
    sys.exit(main())
