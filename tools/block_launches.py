#!/usr/bin/env python
"""Launch list of ONE middle block (forward and backward) of a plan: python tools/block_launches.py [qwen|flux] [targets]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from qflux_amd.modules import LoraConfig
which = sys.argv[1] if len(sys.argv) > 1 else "qwen"
targets = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda", 0)
if targets == "regex":
    import importlib.util
    spec = importlib.util.spec_from_file_location("tfg", os.path.join(ROOT, "tests", "test_flux_gpu.py")); m_ = importlib.util.module_from_spec(spec); spec.loader.exec_module(m_)
    targets = m_._REFERENCE_REGEX
kw = {} if not targets else dict(target_modules=targets)
if which == "qwen":
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.trainer import QwenLoraTrainStep
    with torch.device(dev):
        dit = QwenImageTransformer2DModel(num_layers=3)
    with torch.no_grad():
        for n, p in dit.named_parameters(): p.normal_(0.0, 0.02)
    dit.add_adapter(LoraConfig(r=16, lora_alpha=16, **kw), "default")
    st = QwenLoraTrainStep(dit)
    emb = dict(image_latents=torch.randn(1, 1024, 64).half().to(dev), control_latents=torch.randn(1, 1024, 64).half().to(dev),
               prompt_embeds=torch.randn(1, 384, 3584).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
    st.train_step(emb)
    marks = ("transformer_blocks.1.", "transformer_blocks.0.")
else:
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.trainer import FluxKontextTrainStep
    from qflux_amd.trainer.flux_step import prepare_latent_image_ids
    with torch.device(dev):
        dit = FluxTransformer2DModel(num_layers=1, num_single_layers=3, guidance_embeds=True)
    with torch.no_grad():
        for n, p in dit.named_parameters(): p.normal_(0.0, 0.02)
    dit.add_adapter(LoraConfig(r=16, lora_alpha=16, **kw), "default")
    st = FluxKontextTrainStep(dit)
    ctl = prepare_latent_image_ids(32, 32); ctl[:, 0] = 1
    emb = dict(image_latents=torch.randn(1, 1024, 64).half().to(dev), control_latents=torch.randn(1, 1024, 64).half().to(dev),
               prompt_embeds=torch.randn(1, 512, 4096).half().to(dev), pooled_prompt_embeds=torch.randn(1, 768).half().to(dev),
               text_ids=torch.zeros(512, 3), control_ids=ctl, latent_hw=(32, 32))
    st.train_step(emb)
    marks = ("single_transformer_blocks.1.", "single_transformer_blocks.0.")
plan = list(dit._plans.values())[0]
idx = {pre: i for i, pre in plan.bwd.marks}
lo, hi = idx.get(marks[0] if marks[0] in idx else None, 0), idx.get(marks[1], len(plan.bwd.calls))
mk = sorted(plan.bwd.marks)
print("backward marks:", [(i, p) for i, p in mk])
a = [i for i, p in mk if p == marks[0]][0]; prev = max([i for i, p in mk if i < a] + [0])
names = []
for ent in plan.bwd.calls[prev:a]:
    if ent[0] is None: names.append("py"); continue
    n = ent[0].__name__ + (f"[{ent[1][1]}]" if "batch" in ent[0].__name__ or "grouped" in ent[0].__name__ else "") + ("@side" if len(ent) > 2 else "")
    names.append(n)
print(f"backward segment up to the middle block's mark: {len([n for n in names if n != 'py'])} launches\n  " + "\n  ".join(names))
fm = [i for i, e in enumerate(plan.fwd.calls) if e[0] is not None and e[0].__name__.startswith("qfx_attn_fwd")]
seg = plan.fwd.calls[fm[-2] + 1:fm[-1] + 1] if len(fm) >= 2 else plan.fwd.calls
fn = [("py" if e[0] is None else e[0].__name__ + (f"[{e[1][1]}]" if "batch" in e[0].__name__ or "grouped" in e[0].__name__ else "")) for e in seg]
print(f"forward, attention to attention (one block): {len([n for n in fn if n != 'py'])} launches\n  " + "\n  ".join(fn))
