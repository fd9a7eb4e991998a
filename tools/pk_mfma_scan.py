#!/usr/bin/env python
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for PACKED fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32)
that read a register whose last writer in program text is an MFMA -- the one pattern that was necessary for the run-to-run differences
of the fused dQ epilogue in round 4 (profiles/r05_nondeterminism.md: `v_pk_mul_f32 vdst, s[..], v[accumulator pair]`, 59 / 59 launches
differ; gone with the product kept scalar).  ADVICE r5: a compiler update, or a new epilogue that multiplies accumulators, can bring
the pattern back in any translation unit -- tests/test_isa_scan_cpu.py runs this scan over every csrc/*.hip at build flags and pins
the per-kernel counts.
    python tools/pk_mfma_scan.py file.s [...]      -> JSON {kernel: [line numbers]}
Linear scan per kernel (control flow ignored: a register's "last writer" is the last one in text order, which over-reports at joins
rather than under-reports)."""
import json, re, sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(path):
    sites, kern, writer = {}, None, {}
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        if s.endswith(":"):
            if s.startswith("_Z") or s.startswith("qfx"):
                kern, writer = s[:-1], {}
            continue
        if kern is None:
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if not ops:
            continue
        if mn.startswith("v_pk_") and mn.endswith("_f32"):
            src = set()
            for o in ops[1:]:
                src |= regs(o)
            if any(writer.get(r) == "mfma" for r in src):
                sites.setdefault(kern, []).append(ln)
        if mn.startswith(("v_mfma", "v_smfmac")):
            for r in regs(ops[0]):
                writer[r] = "mfma"
        elif mn.startswith(("v_", "ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")):
            for r in regs(ops[0]):
                writer[r] = "other"
        elif mn.startswith("ASMSTART"):
            pass
    return sites


if __name__ == "__main__":
    out = {}
    for p in sys.argv[1:]:
        out[p] = scan(p)
    print(json.dumps(out, indent=1))
