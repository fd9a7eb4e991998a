#!/usr/bin/env python
"""Where the side-stream gradient launches land: per DiT block, duration of the lora_grad batch on the side stream (while the
main stream runs the next block's first GEMMs) and how long the main stream sits at the join."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 60
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=layers)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
S_t, T = 1024, 384
emb = dict(image_latents=torch.randn(1, S_t, 64).half().to(dev), control_latents=torch.randn(1, S_t, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
noise = torch.randn(1, S_t, 64); u = torch.tensor([0.4])
step = QwenLoraTrainStep(dit)
for _ in range(4):
    step.train_step(emb, noise=noise, u=u)
plan = list(dit._plans.values())[0]
prog = plan.bwd
main = torch.cuda.current_stream()
E = lambda: torch.cuda.Event(enable_timing=True)
side_ev, join_ev = [], []
t0 = E(); t0.record(main)
calls = prog.calls
i = 0
while i < len(calls):
    ent = calls[i]
    fn, args = ent[0], ent[1]
    if fn is None:
        src = getattr(args, "__code__", None)
        is_join = src is not None and "wait_event" in src.co_names and src.co_name == "<lambda>"
        if is_join:
            a, b = E(), E(); a.record(main); args(); b.record(main); join_ev.append((a, b))
        else:
            args()
    elif len(ent) > 2:
        a, b = E(), E(); a.record(prog.side); rc = fn(*args, prog.side.cuda_stream); b.record(prog.side); side_ev.append((a, b)); assert rc == 0
    else:
        rc = fn(*args, main.cuda_stream); assert rc == 0
    i += 1
t1 = E(); t1.record(main)
torch.cuda.synchronize()
sd = [a.elapsed_time(b) * 1e3 for a, b in side_ev]
st = [t0.elapsed_time(a) * 1e3 for a, _ in side_ev]
jn = [a.elapsed_time(b) * 1e3 for a, b in join_ev]
out = {"bwd_ms": t0.elapsed_time(t1), "side_launches": len(sd), "side_us_avg": sum(sd) / len(sd), "side_us_max": max(sd),
       "join_wait_us_avg": sum(jn) / len(jn), "join_wait_us_max": max(jn), "join_wait_total_ms": sum(jn) / 1e3,
       "side_us_first8": [round(x, 1) for x in sd[:8]], "join_us_first8": [round(x, 1) for x in jn[:8]]}
print(json.dumps(out))
