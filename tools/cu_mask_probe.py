#!/usr/bin/env python
"""Which CUs a CU-masked stream (qfx_stream_create_cu_masked) really gets: blocks report HW_ID / XCC_ID."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops
from qflux_amd._lib import lib, check
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)


def probe(stream, n):
    out = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    check(lib.qfx_debug_where(out.data_ptr(), n, stream.cuda_stream), "where")
    stream.synchronize()
    v = out.cpu().view(n, 2)
    cus = set()
    for hw, xcc in v.tolist():
        hw &= 0xFFFFFFFF
        cus.add((xcc & 0xF, (hw >> 13) & 0x7, (hw >> 12) & 1, (hw >> 8) & 0xF))   # (xcc, se, sh, cu)
    return sorted(cus)


res = {}
for n_cus in (16, 32):
    st = ops.side_stream(dev, n_cus)
    c = probe(st, 256)
    per_xcc = {}
    for x in c:
        per_xcc[x[0]] = per_xcc.get(x[0], 0) + 1
    res[f"mask{n_cus}"] = {"distinct_cus": len(c), "per_xcc": per_xcc, "external": isinstance(st, torch.cuda.ExternalStream)}
c = probe(torch.cuda.current_stream(), 2048)
res["unmasked"] = {"distinct_cus": len(c)}
print(json.dumps(res))
