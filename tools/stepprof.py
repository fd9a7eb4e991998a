#!/usr/bin/env python
"""In-process per-call profile of one training step: replays the forward/backward launch programs with a HIP event
pair around every C-ABI call and aggregates by entry point (and GEMM shape).  Writes gpurun_out/stepprof.json."""
import collections, ctypes as C, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 60
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=layers)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else p.fill_(1.0 if "norm" in n else 0.0)
dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
step = QwenLoraTrainStep(dit)
side = res // 16; S_t = side * side; T = 384
emb = dict(image_latents=torch.randn(1, S_t, 64).half().to(dev), control_latents=torch.randn(1, S_t, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, T, 3584) * 4).half().to(dev), prompt_embeds_mask=None,
           img_shapes=[[(1, side, side), (1, side, side)]])
for _ in range(2):
    step.train_step(emb)
torch.cuda.synchronize()
plan = list(dit._plans.values())[0]
agg = collections.defaultdict(lambda: [0, 0.0])

def label(fn, args):
    name = fn.__name__
    if name in ("qfx_gemm_bf16", "qfx_gemm_grouped"):
        a = args[0]
        try:
            g = a._obj
        except AttributeError:
            g = a
        gs = [g] if isinstance(g, L.GemmArgs) else list(g)
        return name + ":" + "+".join(f"{x.M}x{x.N}x{x.K1 + x.K2}" for x in gs[:2]) + (f"(x{len(gs)})" if len(gs) > 2 else "") + f":epi{gs[0].epi}"
    return name

for phase, prog in (("fwd", plan.fwd), ("bwd", plan.bwd)):
    st_obj = torch.cuda.current_stream(); st = st_obj.cuda_stream
    evs = []
    for ent in prog.calls:
        fn, args = ent[0], ent[1]      # (side-stream entries carry a third field; the profiled replay keeps one stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st_obj)
        if fn is None:
            args(); lab = "py:" + getattr(args, "__name__", "callable")
        else:
            rc = fn(*args, st); assert rc == 0; lab = label(fn, args)
        e1.record(st_obj)
        evs.append((phase + ":" + lab, e0, e1))
    torch.cuda.synchronize()
    for lab, e0, e1 in evs:
        agg[lab][0] += 1; agg[lab][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
out = {"total_ms": tot, "calls": {k: {"n": v[0], "ms": round(v[1], 3), "avg_us": round(v[1] / v[0] * 1e3, 1)} for k, v in rows}}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stepprof.json"), "w"), indent=1)
print("total ms (sum of events):", round(tot, 2))
for k, v in rows[:45]:
    print(f"{k:75s} n={v[0]:5d} {v[1]:8.2f} ms {100 * v[1] / tot:5.1f}%  avg {v[1] / v[0] * 1e3:8.1f} us")
