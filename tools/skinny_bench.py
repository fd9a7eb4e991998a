#!/usr/bin/env python
"""Where do the rank-r / LayerNorm launches of the step spend their time?  Stand-alone timings at the headline shapes (M = 2048
image + 384 text rows, D = 3072, r = 16): fused LN+down (as the step launches it: two problems in one grid), the same launch with
the adapters switched off (W_hi = NULL: LayerNorm part only), the two separate kernels, the single-adapter down projection.
Inputs rotate through a ring of buffers larger than the Infinity Cache so that the activations come from HBM as in the step."""
import ctypes as C, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L, ops
DEV, BF = "cuda:0", torch.bfloat16
D, R = 3072, 48
ROWS = (2048, 384)
NRING = 24          # 24 x (15 MB in + 15 MB out) > 256 MB

def split(a):
    hi = a.to(BF); lo = (a - hi.float()).to(BF); return hi, lo

def bench(fn, iters=NRING * 2):
    for i in range(4): fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

xs = [[torch.randn(r, D, device=DEV).to(BF) for r in ROWS] for _ in range(NRING)]
ys = [[torch.empty(r, D, dtype=BF, device=DEV) for r in ROWS] for _ in range(NRING)]
mod = (torch.randn(1, 2 * D, device=DEV) * 0.3).to(BF)
A = torch.randn(R, D, device=DEV) * 0.05
a_hi, a_lo = split(A)
a16_hi, a16_lo = a_hi[:16].contiguous(), a_lo[:16].contiguous()
exts = [torch.zeros(r, 192, dtype=BF, device=DEV) for r in ROWS]
uts = [(torch.zeros(R, (r + 127) // 128 * 128, dtype=BF, device=DEV), torch.zeros(R, (r + 127) // 128 * 128, dtype=BF, device=DEV)) for r in ROWS]

def ln_down(i, adapted=True):
    arr = (L.LnDownArgs * 2)()
    for j, r in enumerate(ROWS):
        a = arr[j]
        a.ln.x, a.ln.shift, a.ln.scale, a.ln.mod_bstride, a.ln.y = xs[i % NRING][j].data_ptr(), mod[:, :D].data_ptr(), mod[:, D:].data_ptr(), 2 * D, ys[i % NRING][j].data_ptr()
        a.ln.rows, a.ln.D, a.ln.rows_per_batch, a.ln.eps = r, D, r, 1e-6
        if adapted:
            a.W_hi, a.W_lo, a.ldw, a.R = a_hi.data_ptr(), a_lo.data_ptr(), D, R
            a.ext, a.ld_ext = exts[j].data_ptr(), exts[j].stride(0)
            a.Ut_hi, a.Ut_lo, a.ld_ut = uts[j][0].data_ptr(), uts[j][1].data_ptr(), uts[j][0].stride(0)
            a.group_R, a.group_stride = 16, 64
    L.check(L.lib.qfx_ln_down_fwd(arr, 2, ops.stream_ptr()), "ln_down")

def ln_only(i):
    for j, r in enumerate(ROWS):
        L.check(L.lib.qfx_ln_modulate_fwd(xs[i % NRING][j].data_ptr(), mod[:, :D].data_ptr(), mod[:, D:].data_ptr(), 2 * D, ys[i % NRING][j].data_ptr(), r, D, r, 1e-6, ops.stream_ptr()), "ln")

def down48(i):
    for j, r in enumerate(ROWS):
        ops.lora_down(xs[i % NRING][j], a_hi, a_lo, ext=exts[j], Ut=uts[j], group_R=16, group_stride=64)

def down16(i):
    ops.lora_down(xs[i % NRING][0], a16_hi, a16_lo, ext=exts[0], Ut=(uts[0][0][:16], uts[0][1][:16]), group_R=16, group_stride=0)

res = {"ln_down_fused_2problems": bench(ln_down), "ln_down_adapters_off": bench(lambda i: ln_down(i, False)),
       "ln_modulate_fwd_x2_launches": bench(ln_only), "lora_down_R48_x2_launches": bench(down48), "lora_down_R16_img_only": bench(down16)}
for k, v in res.items(): print(f"{k:34s} {v:7.1f} us")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "skinny_bench.json"), "w"), indent=1)
