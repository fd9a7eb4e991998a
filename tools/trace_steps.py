#!/usr/bin/env python
"""Per-step kernel-class times from a rocprofv3 kernel trace of tools/step_ablate.py: steps are delimited by adamw_kernel launches and
classified by which kernel classes occur in them (e.g. steps with / without lora_grad launches), then averaged per class.

    python tools/trace_steps.py <kernel_trace.csv> [-o out.json]
"""
import argparse, collections, csv, json, re, statistics

ap = argparse.ArgumentParser(); ap.add_argument("trace"); ap.add_argument("-o", default=None); a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.trace)):
    m = re.search(r"(\w+)(?:<[^(]*>)?\(", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2] == "adamw_kernel"]
steps = []
for lo, hi in zip(marks[:-1], marks[1:]):
    seg = rows[lo + 1:hi + 1]
    by = collections.Counter(); cnt = collections.Counter()
    for s, e, n in seg: by[n] += e - s; cnt[n] += 1
    steps.append({"wall": (rows[hi][1] - rows[lo][1]) / 1e3, "by": by, "cnt": cnt})
groups = collections.defaultdict(list)
for st in steps:
    if st["cnt"].get("gemm256_kernel", 0) < 400: continue          # not a full 60-block step
    key = "with lora_grad" if st["cnt"].get("lora_grad_kernel", 0) else "lora_grad removed"
    groups[key].append(st)
out = {}
for k, ss in groups.items():
    ent = {"steps": len(ss), "wall_ms": round(statistics.median(s["wall"] for s in ss) / 1e3, 3)}
    for kern in ("gemm256_kernel", "attn_bwd_dkv_kernel", "attn_bwd_dq64_kernel", "attn_fwd64_kernel", "ln_mod_bwd_kernel", "ln_down_kernel", "lora_grad_kernel"):
        vals = [s["by"].get(kern, 0) / 1e3 / max(1, s["cnt"].get(kern, 0)) for s in ss]
        ent[kern + "_avg_us"] = round(statistics.median(vals), 2)
        ent[kern + "_launches"] = ss[0]["cnt"].get(kern, 0)
    out[k] = ent
print(json.dumps(out, indent=1))
if a.o: json.dump(out, open(a.o, "w"), indent=1)
