#!/usr/bin/env python
"""Build kernel-variant libraries for tools/step_ab.py:  python tools/build_variants.py name=-DFLAG1,-DFLAG2 name2=-DX ...
-> tools/_ab/libqfx_<name>.so (the four csrc files with the extra flags; objects of unaffected files are shared via a cache
keyed by (file, flags that occur in it))."""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qwen-image-finetune_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab")
SOURCES = ["qfx_gemm.hip", "qfx_gemm_fp8.hip", "qfx_skinny.hip", "qfx_elem.hip", "qfx_attn.hip", "qfx_attn64.hip", "qfx_attn_bwd1.hip", "qfx_cond.hip"]
EXTRA_FLAGS = {"qfx_attn64.hip": ["-fno-slp-vectorize"], "qfx_attn_bwd1.hip": ["-fno-slp-vectorize"]}


def main():
    os.makedirs(OUT, exist_ok=True)
    jobs = {}
    plans = []
    for spec in sys.argv[1:]:
        name, _, fl = spec.partition("=")
        flags = [f for f in fl.split(",") if f]
        objs = []
        for src in SOURCES:
            text = (open(os.path.join(CSRC, src)).read() + open(os.path.join(CSRC, "qfx_common.h")).read() +
                    open(os.path.join(CSRC, "qfx_attn_common.h")).read() + open(os.path.join(ROOT, "include", "qfx.h")).read())
            rel = [f for f in flags if re.sub(r"^-D", "", f).split("=")[0] in text]
            key = hashlib.sha1((src + "|" + " ".join(rel) + "|" + hashlib.sha1(text.encode()).hexdigest()).encode()).hexdigest()[:16]
            obj = os.path.join(OUT, f"{src[:-4]}_{key}.o")
            if not os.path.exists(obj) and obj not in jobs:
                cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                       *EXTRA_FLAGS.get(src, []), *rel, "-c", os.path.join(CSRC, src), "-o", obj]
                jobs[obj] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            objs.append(obj)
        plans.append((name, objs))
    for obj, p in jobs.items():
        out, _ = p.communicate()
        if p.returncode:
            raise SystemExit(f"compile failed for {obj}:\n{out}")
    for name, objs in plans:
        lib = os.path.join(OUT, f"libqfx_{name}.so")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise SystemExit(r.stdout)
        print("built", lib)


if __name__ == "__main__":
    main()
