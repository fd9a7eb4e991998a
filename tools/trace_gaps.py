#!/usr/bin/env python
"""Idle time inside one step from a rocprofv3 kernel trace: union of kernel intervals over all queues between two consecutive
adamw_kernel launches, the gaps between them by the kernel that follows each gap, and busy time by kernel class.

    python tools/trace_gaps.py gpurun_out/r05prof/stats/p_kernel_trace.csv [-o profiles/r05_step_gaps.json]
"""
import argparse, collections, csv, json, re

ap = argparse.ArgumentParser(); ap.add_argument("trace"); ap.add_argument("-o", default=None); a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.trace)):
    m = re.search(r"(\w+)(?:<[^(]*>)?\(", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40], int(r["Queue_Id"])))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2] == "adamw_kernel"]
# the last two optimizer launches bracket one full step
lo, hi = marks[-2], marks[-1]
step = rows[lo + 1:hi + 1]
t0, t1 = rows[lo][1], rows[hi][1]
busy = 0; cur_s, cur_e = None, None
gaps = collections.Counter(); gapn = collections.Counter()
prev_end = t0
for s, e, n, q in step:
    if s > prev_end:
        gaps[n] += s - prev_end; gapn[n] += 1
    prev_end = max(prev_end, e)
by = collections.Counter(); cnt = collections.Counter()
for s, e, n, q in step: by[n] += e - s; cnt[n] += 1
wall = t1 - t0
idle = sum(gaps.values())
out = {"wall_us": wall / 1e3, "idle_us": idle / 1e3, "idle_frac": idle / wall, "kernels": len(step), "queues": sorted({q for *_, q in step}),
       "gap_before_us": {k: [round(v / 1e3, 1), gapn[k], round(v / 1e3 / gapn[k], 2)] for k, v in gaps.most_common(14)},
       "kernel_time_us": {k: [round(v / 1e3, 1), cnt[k]] for k, v in by.most_common(20)}}
print(json.dumps(out, indent=1))
if a.o: json.dump(out, open(a.o, "w"), indent=1)
