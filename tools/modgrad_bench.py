"""mod_grad stand-alone timing (variant libraries via QFX_LIB_PATH)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qwen-image-finetune_amd"))
from qflux_amd import _lib as L, ops
DEV, BF = "cuda:0", torch.bfloat16
D = 3072
for rows in (2048, 384):
    N = 12
    xs = [torch.randn(rows, D, device=DEV).to(BF) for _ in range(4 * N)]
    outs = torch.zeros(3, 1, D, device=DEV)
    for gate in (False, True):
        def run(i):
            a = L.ModGradArgs()
            a.dy, a.ld_dy, a.x, a.ld_x = xs[4 * (i % N)].data_ptr(), D, xs[4 * (i % N) + 1].data_ptr(), D
            if gate:
                a.dxo, a.ld_dxo, a.y, a.ld_y = xs[4 * (i % N) + 2].data_ptr(), D, xs[4 * (i % N) + 3].data_ptr(), D
                a.dgate = outs[2].data_ptr()
            a.dshift, a.dscale, a.out_bstride = outs[0].data_ptr(), outs[1].data_ptr(), D
            a.rows, a.D, a.rows_per_batch, a.eps = rows, D, rows, 1e-6
            L.check(L.lib.qfx_mod_grad(C.byref(a), ops.stream_ptr()), "mg")
        for i in range(3): run(i)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(2 * N): run(i)
        e.record(); torch.cuda.synchronize()
        print(f"{os.environ.get('QFX_LIB_PATH', 'base'):40s} rows {rows} gate {gate}: {s.elapsed_time(e) / (2 * N) * 1e3:.1f} us", flush=True)
