#!/usr/bin/env python
"""Audit of a hand-allocated-accumulator kernel (cdna_hip_programming.md 5.7 item 4):  python tools/agpr_audit.py file.s [kernel substring]
For every kernel whose name contains the substring: vgpr / agpr counts, scratch, spills, and every compiler-emitted v_accvgpr_* or
a[...] reference OUTSIDE ;;#ASMSTART / ;;#ASMEND; plus the instruction mix of the hottest loop (MFMA vs other issues)."""
import re, sys
path = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "fwd64"
lines = open(path).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\S*):", lines[i])
    if m and sub in m.group(1):
        name = m.group(1); inasm = False; bad = []; n = 0; mf = 0; other = 0; loops = {}
        j = i + 1
        while j < len(lines) and "s_endpgm" not in lines[j]:
            l = lines[j]
            if ";;#ASMSTART" in l: inasm = True
            elif ";;#ASMEND" in l: inasm = False
            else:
                t = l.split(";")[0].strip()
                if t and not t.endswith(":") and not t.startswith("."):
                    n += 1
                    if not inasm and (re.search(r"\ba\[?\d", t) or "accvgpr" in t): bad.append((j + 1, t))
            j += 1
        desc = {}
        for k in range(j, min(j + 400, len(lines))):
            mm = re.match(r"\s*\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size|group_segment_fixed_size|next_free_sgpr)\s+(\S+)", lines[k])
            if mm: desc[mm.group(1)] = mm.group(2)
            mm = re.match(r";\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|TotalNumVgprs|vgpr_spill_count)\S*:?\s*(\S+)", lines[k].replace(".", ""))
            if mm: desc[mm.group(1)] = mm.group(2)
        print(name[:80], "instructions", n, desc)
        print("  compiler accesses to the accumulator half outside asm:", len(bad))
        for b in bad[:10]: print("   ", b)
        i = j
    i += 1
