#!/usr/bin/env python
"""Round 6: is the one-pass attention backward's fence-free CU-to-CU hand-off (sc1 stores / sc1 LDS-DMA loads + a counter behind
s_waitcnt vmcnt(0)) sound UNDER LOAD?  The same family of hand-off failed in lora_grad under the 60-block step and passed every small
test (profiles/r06_grad_handoff.json).  Here: many launches of qfx_attn_bwd_fused, compared bit for bit with the first one, while a
second stream keeps the chip busy with large GEMMs and copies (uneven load, other L2 traffic); shapes whose heads stay inside one XCD
(S = 2432, 24 heads) and shapes whose heads straddle XCDs (S = 8576; 2432 with 25 heads).
    python tools/attn_onepass_stress.py [--iters 200]"""
import argparse, ctypes as C, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from qflux_amd import ops, _lib as L
import test_attention_onepass_gpu as T

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=200); ap.add_argument("--S", default="2432:24:1,2432:25:1,8576:24:1,1216:7:3")
args = ap.parse_args()
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
ga, gb = torch.randn(4096, 8192, device=dev).bfloat16(), torch.randn(8192, 8192, device=dev).bfloat16()
big, big2 = torch.empty(256 << 20, dtype=torch.uint8, device=dev), torch.empty(256 << 20, dtype=torch.uint8, device=dev)
out = {}
for spec in args.S.split(","):
    S, H, Bn = (int(x) for x in spec.split(":"))
    a, t, keep = T._setup(S, H, Bn, 0, 16, fused=True, seed=11)
    ws = ops.attn_bwd_fused_workspace(a)
    st = ops.stream_ptr()
    ref = None
    bad = 0
    for it in range(args.iters):
        with torch.cuda.stream(side):      # uneven company: a GEMM burst every iteration, a 256 MB copy every third
            for _ in range(1 + it % 3):
                torch.matmul(ga, gb)
            if it % 3 == 0:
                big2.copy_(big)
        t["dqkv"].zero_(); t["dsum"].zero_()
        for p_ in t["parts"]:
            p_.zero_()
        assert L.lib.qfx_attn_bwd_fused(C.byref(a), st) == 0
        torch.cuda.current_stream().synchronize()
        cur = [t["dqkv"].clone(), t["dsum"].clone()] + [p_.clone() for p_ in t["parts"]]
        if ref is None:
            ref = cur
        elif not all(torch.equal(x, y) for x, y in zip(cur, ref)):
            bad += 1
    torch.cuda.synchronize()
    turn_zero = bool((ws[1] == 0).all())
    out[spec] = {"iters": args.iters, "launches_that_differ_from_the_first": bad, "turn_counters_back_at_zero": turn_zero}
    print(spec, out[spec], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_attn_onepass_stress.json"), "w"), indent=1)
