#!/usr/bin/env python
"""Host-side cost of one training step: wall time of train_step() WITHOUT synchronising (the Python replay of ~1.46 k ctypes calls
runs ahead of the GPU) against the GPU time of the step; and the per-step time when the caller reads the loss every step (the
reference's loop logs it: the enqueue time is then exposed unless it is shorter than the GPU time of the previous step's tail)."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd")); sys.path.insert(0, ROOT)
from qflux_amd.models import QwenImageTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
dev = torch.device("cuda", 0); torch.manual_seed(0)
with torch.device(dev):
    dit = QwenImageTransformer2DModel(num_layers=60)
with torch.no_grad():
    for n, p in dit.named_parameters():
        p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.zero_())
dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
step = QwenLoraTrainStep(dit, lr=1e-4)
emb = dict(image_latents=torch.randn(1, 1024, 64).half().to(dev), control_latents=torch.randn(1, 1024, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, 384, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
for _ in range(5): step.train_step(emb)
torch.cuda.synchronize()
res = {}
# (a) enqueue time: GPU idle at start, no sync inside
t = []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step.train_step(emb); t.append((time.perf_counter() - t0) * 1e3)
res["enqueue_ms_per_step"] = round(sorted(t)[len(t) // 2], 2)
# (b) back to back, no sync
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step.train_step(emb)
torch.cuda.synchronize(); res["ms_per_step_no_sync"] = round((time.perf_counter() - t0) / 20 * 1e3, 2)
# (c) loss read every step
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step.train_step(emb).item()
res["ms_per_step_loss_item_each_step"] = round((time.perf_counter() - t0) / 20 * 1e3, 2)
# (d) captured graph, loss read every step
try:
    g = step.capture_graph(emb)
    for _ in range(3): g(emb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g(emb).item()
    res["ms_per_step_graph_loss_item_each_step"] = round((time.perf_counter() - t0) / 20 * 1e3, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g(emb)
    torch.cuda.synchronize(); res["ms_per_step_graph_no_sync"] = round((time.perf_counter() - t0) / 20 * 1e3, 2)
except Exception as e:  # noqa: BLE001
    res["graph_error"] = repr(e)[:200]
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "enqueue_time.json"), "w"), indent=1)
