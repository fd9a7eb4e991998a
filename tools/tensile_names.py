#!/usr/bin/env python
"""Round 6 (VERDICT r5 #2b): which Tensile kernel does the vendor library pick per shape?  Target of `rocprofv3 --kernel-trace`:
one torch.matmul (bf16, C = A B^T) per shape, 3 launches each, separated by a marker fill of a distinctive size; `--parse <csv>`
turns the kernel trace into {shape: kernel name, grid, workgroup, LDS, VGPR/AGPR, avg us}."""
import csv, json, os, sys
SHAPES = [(2048, 3072, 3072), (2432, 3072, 3072), (2432, 9216, 3072), (2432, 12288, 3072), (2432, 3072, 12288), (2432, 3072, 9216),
          (4864, 3072, 3072), (4864, 12288, 3072), (8192, 8192, 8192)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out, cur = {}, []
    it = iter(SHAPES)
    for r in rows:
        nm = r["Kernel_Name"]
        if "Cijk" in nm or "gemm" in nm.lower():
            cur.append(r)
        elif cur:      # a non-GEMM kernel (the marker fill) closes a group
            if len(cur) < 5:      # the warm-up launch
                cur = []
                continue
            shp = next(it, None)
            if shp is None:
                break
            last = cur[-1]
            us = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in cur[-3:]]
            out["%dx%dx%d" % shp] = {"kernel": last["Kernel_Name"], "grid": last.get("Grid_Size_X", last.get("Grid_Size")), "workgroup": last.get("Workgroup_Size_X", last.get("Workgroup_Size")),
                                     "lds": last.get("LDS_Block_Size"), "vgpr": last.get("VGPR_Count"), "agpr": last.get("Accum_VGPR_Count"),
                                     "sgpr": last.get("SGPR_Count"), "us": [round(u, 1) for u in us], "launches_in_group": len(cur),
                                     "kernels_in_group": sorted({x["Kernel_Name"][:120] for x in cur})}
            cur = []
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_tensile_kernels.json"), "w"), indent=1)
    for k, v in out.items():
        print(k, v["us"], v["grid"], v["workgroup"], v["lds"], v["vgpr"], v["agpr"], v["kernel"][:200])
    sys.exit(0)
import torch
DEV, BF = "cuda:0", torch.bfloat16
marker = torch.empty(12345, device=DEV)
warm = torch.randn(256, 256, device=DEV).to(BF)
torch.matmul(warm, warm.t()); marker.fill_(1.0); torch.cuda.synchronize()
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=DEV).to(BF)
    b = torch.randn(N, K, device=DEV).to(BF)
    o = torch.empty(M, N, dtype=BF, device=DEV)
    torch.cuda.synchronize()
    marker.fill_(0.0)      # opens the group (closes the previous one / the warm-up)
    for _ in range(5):
        torch.matmul(a, b.t(), out=o)
    torch.cuda.synchronize()
marker.fill_(2.0)
torch.cuda.synchronize()
