import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
import test_fullsize_cfgs_gpu as T
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep
hip = T._qwen_full(2, None)
emb, noise, u = T._emb_1024()
hip.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
with torch.no_grad():
    for n, p in hip.named_parameters():
        if "lora_B" in n:
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(9)).to(p.device) * 1e-2)
step = QwenLoraTrainStep(hip)
st = hip.lora_store
gs = []
for sc in (1.0, 1.0, 0.5, 0.5):
    l = step.forward_backward(emb, noise=noise, u=u, grad_scale=sc).item()
    gs.append((sc, l, st.gflat.clone())); step.zero_grad()
g1 = gs[0][2]
print("fused", os.environ.get("QFX_FUSE_HEAD_LORA", "1"))
print("repeat 1.0 equal:", torch.equal(gs[0][2], gs[1][2]), " repeat 0.5 equal:", torch.equal(gs[2][2], gs[3][2]))
d = (gs[2][2] * 2 - g1).abs()
print("lin", (d.max() / g1.abs().max()).item())
# which parameters deviate
off = 0
names = []
for n, p in hip.named_parameters():
    if "lora" in n:
        o = st.offset_of(p); k = p.numel()
        e = d[o:o + k].max().item() / (g1[o:o + k].abs().max().item() + 1e-30)
        if e > 1e-6: names.append((n, e))
print(len(names), names[:12])
