#!/usr/bin/env python
"""Build libqfx from the csrc/ + include/ of a git revision:  python tools/build_rev.py <rev> <name>  ->  tools/_ab/libqfx_<name>.so
(loaded by tools/step_ab.load_variant; used for same-box A/B of a previous round's kernels against HEAD)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rev, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "tools", "_ab", "src_" + name)
os.makedirs(src, exist_ok=True)
files = subprocess.run(["git", "ls-tree", "--name-only", rev, "qwen-image-finetune_amd/csrc/"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
for f in files + ["include/qfx.h"]:
    open(os.path.join(src, os.path.basename(f)), "w").write(subprocess.run(["git", "show", f"{rev}:{f}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout)
entry = subprocess.run(["git", "show", f"{rev}:__graft_entry__.py"], cwd=ROOT, capture_output=True, text=True, check=True).stdout
noslp = "-fno-slp-vectorize" in entry
procs, objs = [], []
for f in sorted(os.listdir(src)):
    if f.endswith(".hip"):
        obj = os.path.join(src, f[:-4] + ".o")
        extra = ["-fno-slp-vectorize"] if (noslp and f == "qfx_attn64.hip") else []
        procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + src, *extra, "-c", os.path.join(src, f), "-o", obj]))
        objs.append(obj)
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(ROOT, "tools", "_ab", f"libqfx_{name}.so")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print("built", lib)
