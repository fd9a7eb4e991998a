"""GPU-side cost of a dependent launch in one stream: N tiny kernels back to back (events around the whole train)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import ops
x = torch.randn(1024, device="cuda")
out = torch.zeros(1, device="cuda")
big = torch.randn(2432, 3072, device="cuda").to(torch.bfloat16)
def run(fn, n=2000):
    for _ in range(50): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("tiny sumsq kernel: %.2f us per launch" % run(lambda: ops.sumsq(x, out)))
y = torch.empty_like(big)
print("torch copy 15MB: %.2f us per launch" % run(lambda: y.copy_(big), 500))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): ops.sumsq(x, out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(200): ops.sumsq(x, out)
torch.cuda.synchronize()
print("tiny kernel inside a 200-node hipGraph: %.2f us per node" % (run(lambda: g.replay(), 20) / 200))
# GPU-side gap with the host far ahead: block the GPU with ~15 ms of GEMMs, enqueue 1000 tiny kernels behind them
w = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
def gpu_side(fn, n=1000):
    torch.cuda.synchronize()
    for _ in range(14): w @ w
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("tiny kernel, host ahead (GPU-side dependent-launch cost): %.2f us" % gpu_side(lambda: ops.sumsq(x, out)))
xs = [torch.randn(1024, device="cuda") for _ in range(2)]
from qflux_amd import _lib as L
print("ln_modulate_fwd-sized elementwise (15 MB in, 15 MB out), host ahead: %.2f us" % gpu_side(lambda: y.copy_(big), 600))
