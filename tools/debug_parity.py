import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_util import *
from common import TINY
from oracle import qwen_dit as O
from qflux_amd.trainer import QwenLoraTrainStep
cfg = dict(TINY)
oracle, hip = build_pair(cfg, r=4, device="cuda:0")
emb, noise, u = tiny_embeddings()
loss_o, pred_o = O.qwen_compute_loss(oracle, emb, noise, u, BF, return_pred=True)
loss_o.backward()
step = QwenLoraTrainStep(hip)
loss_h = step.forward_backward(emb, noise=noise, u=u)
torch.cuda.synchronize()
og = {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}
for n, p in hip.named_parameters():
    if "lora" in n:
        g, r = p.grad.float().cpu(), og[n]
        cos = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
        print(f"{n:70s} rel={relmax(g, r):.3e} cos={cos:.4f} |ref|max={r.abs().max():.3e} |hip|max={g.abs().max():.3e} ratio={(g.norm()/r.norm()).item():.3f}")
