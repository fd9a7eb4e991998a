"""Race probe: the LoRA gradients of ONE forward_backward on identical models / inputs must agree up to fp32-atomic order."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/qwen-image-finetune_amd", ROOT + "/tests/golden", ROOT + "/tests"):
    sys.path.insert(0, p)
from common import TINY
from parity_util import build_pair, tiny_embeddings
from qflux_amd.trainer import QwenLoraTrainStep
DEV = "cuda:0"
tg = sys.argv[1] if len(sys.argv) > 1 else "all-linear"
if tg != "all-linear":
    tg = tuple(tg.split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
_, a = build_pair(dict(TINY), device=DEV, targets=tg)
sa = QwenLoraTrainStep(a, lr=1e-2)
e, n, u = tiny_embeddings(seed=11)
ref = None
bad = {}
for rep in range(reps):
    sa.zero_grad()
    sa.forward_backward(e, noise=n, u=u)
    torch.cuda.synchronize()
    g = a.lora_store.gflat.clone()
    if ref is None:
        ref = g
        continue
    d = (g - ref).abs()
    for name, p_, off, k in a.lora_store.entries:
        m = d[off:off + k].max().item() / (ref[off:off + k].abs().max().item() + 1e-30)
        if m > 1e-4:
            bad.setdefault(name, []).append(round(m, 4))
print("targets", tg, "reps", reps, "differing tensors:", bad if bad else "none")
