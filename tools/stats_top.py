#!/usr/bin/env python
"""Top rows of a rocprofv3 kernel_stats.csv as ms per step:  python tools/stats_top.py <csv> <steps incl. warm-up> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = 0.0
for r in rows:
    tot += int(r["TotalDurationNs"]) / 1e6 / steps
for r in rows[:n]:
    print(f"{r['Name'][:86]:86s} {int(r['Calls']) / steps:7.1f}/step {int(r['TotalDurationNs']) / 1e6 / steps:8.2f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
print(f"sum of all kernels {tot:.1f} ms/step")
