#!/usr/bin/env python
"""Does replaying the training step from ONE hipGraph beat the Python replay of its ~1.45k launches?

    python tools/graph_probe.py [--layers 60] [--steps 20] [--batch 1]

Captures refresh_lora_operands + forward program + loss + backward program (side-stream gradient launches included: the
fork / join events are captured with them) + clip + AdamW into a torch.cuda.CUDAGraph and times eager vs graph replay
back-to-back, alternating, on the same plan and buffers."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    with torch.device(dev):
        dit = QwenImageTransformer2DModel(num_layers=args.layers)
    with torch.no_grad():
        for n, p in dit.named_parameters():
            if "norm" in n and p.ndim == 1:
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02)
    dit.add_adapter(LoraConfig(r=16, lora_alpha=16, init_lora_weights="gaussian"), "default", generator=torch.Generator().manual_seed(1))
    step = QwenLoraTrainStep(dit, lr=1e-4)
    B, side, T = args.batch, 32, 384
    emb = dict(image_latents=torch.randn(B, side * side, 64).half().to(dev), control_latents=torch.randn(B, side * side, 64).half().to(dev),
               prompt_embeds=(torch.randn(B, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, side, side)] * 2] * B)

    def eager():
        return step.train_step(emb)

    for _ in range(3):
        eager()
    torch.cuda.synchronize()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    gstep = step.capture_graph(emb)
    res = {"eager_ms": [], "graph_ms": []}
    for _ in range(3):
        res["eager_ms"].append(timed(eager, args.steps))
        res["graph_ms"].append(timed(lambda: gstep(emb), args.steps))
    res["layers"], res["steps"], res["batch"] = args.layers, args.steps, B
    print(json.dumps(res))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
