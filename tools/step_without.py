#!/usr/bin/env python
"""Deletion ablation of the headline step: the whole training step timed with ALL launches of one entry point removed from the plan's
forward / backward programs (results are then wrong, timing is what is measured) -- the marginal cost of each kernel class in the real
step, launch boundaries and side-stream overlap included.  Classes interleaved round-robin on one box.
    python tools/step_without.py [--steps 10] [--rounds 3]      -> gpurun_out/step_without.json"""
from __future__ import annotations
import argparse, collections, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd")); sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=60); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--rounds", type=int, default=3)
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
            if p.ndim == 2: p.normal_(0.0, 0.02)
            elif "norm" in n: p.fill_(1.0)
            else: p.zero_()
    dit.add_adapter(LoraConfig(r=16, lora_alpha=16, init_lora_weights="gaussian"), "default", generator=torch.Generator().manual_seed(1234))
    step = QwenLoraTrainStep(dit, lr=0.0, weight_decay=0.0, max_grad_norm=1.0)
    B, side, T = 1, 32, 384
    S_t = side * side
    emb = dict(image_latents=torch.randn(B, S_t, 64).half().to(dev), control_latents=torch.randn(B, S_t, 64).half().to(dev),
               prompt_embeds=(torch.randn(B, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, side, side), (1, side, side)]] * B)
    for _ in range(4):
        step.train_step(emb)
    torch.cuda.synchronize()
    plan = list(dit._plans.values())[0]
    full = {"fwd": list(plan.fwd.calls), "bwd": list(plan.bwd.calls)}
    count = collections.Counter()
    for k in full:
        for ent in full[k]:
            if ent[0] is not None:
                count[ent[0].__name__ + ("@side" if len(ent) > 2 else "")] += 1
    classes = ["none"] + sorted(count)

    def install(cls):
        for k, prog in (("fwd", plan.fwd), ("bwd", plan.bwd)):
            prog.calls = [e for e in full[k] if e[0] is None or (e[0].__name__ + ("@side" if len(e) > 2 else "")) != cls]

    res = {c: [] for c in classes}
    for r in range(args.rounds):
        for c in classes:
            install(c)
            step.train_step(emb); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step.train_step(emb)
            torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / args.steps * 1e3)
    install("none")
    med = {c: sorted(v)[len(v) // 2] for c, v in res.items()}
    out = dict(config=f"Qwen {args.layers} blocks B=1 512^2 r=16", step_ms=med["none"],
               marginal_ms={c: dict(launches=count[c], ms=round(med["none"] - med[c], 3), us_per_launch=round((med["none"] - med[c]) / count[c] * 1e3, 2))
                            for c in classes if c != "none"})
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "step_without.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
