#!/usr/bin/env python
"""Vector-memory instruction order of the persistent GEMM's landing-buffer epilogues (tests/test_isa_scan_cpu.py uses seq()):
    python tools/gemm_epi_vmem_seq.py [qfx_gemm.s]
The counted waits of the landing-buffer side (qfx_gemm.hip, epilogue) are only right if the ONLY vector-memory instructions between
the first counted wait and the end of the passes are the requests (D), the stores (S) and the waits (wN) themselves -- a scratch reload
(X) or a plain load (L) there has its own compiler wait that drains the request just issued, and shifts the count."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qwen-image-finetune_amd", "csrc")


def asm_text(path=None, flags=()):
    if path:
        return open(path).read()
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *flags,
                        "-S", "--cuda-device-only", os.path.join(CSRC, "qfx_gemm.hip"), "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    return r.stdout


def seq(text, name):
    """-> (tokens of the function in program order, index of the landing-buffer side's first token)."""
    lines = text.split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith(name + ":")][0]
    end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith(".Lfunc_end")][0]
    toks = []
    for l in lines[start:end]:
        if "global_load_lds" in l:
            toks.append("D")
        elif "global_store" in l or "buffer_store" in l:
            toks.append("S")
        elif "scratch_" in l:
            toks.append("X")
        elif "global_load" in l or "buffer_load" in l:
            toks.append("L")
        elif (m := re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)):
            toks.append("w" + m.group(1))
    return toks


def landing_side(toks):
    """The landing-buffer passes: the longest run that starts with 'w1 D S' and consists of w/D/S only."""
    best = []
    for i in range(len(toks) - 2):
        if toks[i] == "w1" and toks[i + 1] == "D" and toks[i + 2] == "S":
            j = i
            while j < len(toks) and (toks[j] in ("D", "S") or toks[j].startswith("w")):
                j += 1
            if j - i > len(best):
                best = toks[i:j]
    return best


KERNELS = {
    "dgelu 256x256": "_ZN12_GLOBAL__N_114gemm256_kernelILi3ELi256ELi256ELb0ELb0EEEvNS_11GroupedArgsE",
    "dgelu 160x192": "_ZN12_GLOBAL__N_114gemm256_kernelILi3ELi160ELi192ELb0ELb0EEEvNS_11GroupedArgsE",
    "gate_res 256x256": "_ZN12_GLOBAL__N_114gemm256_kernelILi2ELi256ELi256ELb0ELb0EEEvNS_11GroupedArgsE",
    "gate_res 160x192": "_ZN12_GLOBAL__N_114gemm256_kernelILi2ELi160ELi192ELb0ELb0EEEvNS_11GroupedArgsE",
}

if __name__ == "__main__":
    text = asm_text(sys.argv[1] if len(sys.argv) > 1 else None)
    for k, name in KERNELS.items():
        t = seq(text, name)
        print(k, "\n  all:", " ".join(t), "\n  landing side:", " ".join(landing_side(t)))
