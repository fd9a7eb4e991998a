"""Determinism probe: two identical models trained eagerly on the same batches must stay equal up to fp32-atomic order
(1e-6); a larger difference is a race.  python tools/dbg_graph.py <targets> <reps>"""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bad = 0
for rep in range(reps):
    _, a = build_pair(dict(TINY), device=DEV, targets=tg)
    _, b = build_pair(dict(TINY), device=DEV, targets=tg)
    sa, sb = QwenLoraTrainStep(a, lr=1e-2), QwenLoraTrainStep(b, lr=1e-2)
    worst = 0.0
    for s in (11, 12, 13):
        e, n, u = tiny_embeddings(seed=s)
        sa.train_step(e, noise=n, u=u); sb.train_step(e, noise=n, u=u)
        worst = max(worst, (a.lora_store.pflat - b.lora_store.pflat).abs().max().item())
    if worst > 1e-5:
        d = (a.lora_store.pflat - b.lora_store.pflat).abs()
        i = int(d.argmax())
        name = [n_ for n_, p_, off, k in a.lora_store.entries if off <= i < off + k]
        print("rep", rep, "DIVERGED", worst, "at", name)
        bad += 1
print("targets", tg, "side", os.environ.get("QFX_SIDE_GRADS", "1"), "diverged", bad, "of", reps)
