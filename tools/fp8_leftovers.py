#!/usr/bin/env python
"""Which GEMM launches of a step still contract bf16 operands with the MX-FP8 trunk on ("mxfp8-fb")?  Lists them by program with
shape / segment info (2 blocks; Qwen and FLUX-Kontext shaped)."""
import collections, ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
from qflux_amd import _lib as L
from qflux_amd.models import QwenImageTransformer2DModel, FluxTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import QwenLoraTrainStep, FluxKontextTrainStep
from qflux_amd.trainer.flux_step import prepare_latent_image_ids
dev = torch.device("cuda", 0); torch.manual_seed(0)


def init(dit):
    with torch.no_grad():
        for n, p in dit.named_parameters():
            p.normal_(0.0, 0.02) if p.ndim == 2 else (p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02))
    dit.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
    dit.quantize_trunk("mxfp8-fb")


def report(tag, plan):
    for pname in ("fwd", "bwd"):
        prog = getattr(plan, pname)
        cnt = collections.Counter()
        for ent in prog.calls:
            fn, args = ent[0], ent[1]
            if fn is None:
                continue
            n = fn.__name__
            if n == "qfx_gemm_bf16":
                g = C.cast(args[0], C.POINTER(L.GemmArgs)).contents if not isinstance(args[0], L.GemmArgs) else args[0]
                cnt[f"bf16 M={g.M} N={g.N} K1={g.K1} K2={g.K2} plain2={g.seg2_plain} epi={g.epi}"] += 1
            elif n == "qfx_gemm_grouped":
                arr = args[0]
                for i in range(args[1]):
                    g = arr[i]
                    cnt[f"bf16(grouped) M={g.M} N={g.N} K1={g.K1} K2={g.K2} plain2={g.seg2_plain} epi={g.epi}"] += 1
            elif n.startswith("qfx_gemm_mxfp8"):
                cnt["MX-FP8 " + n] += (args[1] if n.endswith("grouped") else 1)
            elif n == "qfx_quant_mxfp8":
                cnt["stand-alone quantisation pass"] += 1
        print(f"== {tag} {pname}")
        for k, v in sorted(cnt.items()):
            print(f"   {v:3d} x {k}")


with torch.device(dev):
    q = QwenImageTransformer2DModel(num_layers=2)
init(q)
st = QwenLoraTrainStep(q, lr=1e-4)
emb = dict(image_latents=torch.randn(1, 1024, 64).half().to(dev), control_latents=torch.randn(1, 1024, 64).half().to(dev),
           prompt_embeds=(torch.randn(1, 384, 3584) * 4).half().to(dev), prompt_embeds_mask=None, img_shapes=[[(1, 32, 32), (1, 32, 32)]])
st.train_step(emb)
report("qwen 2 blocks", list(q._plans.values())[0])
del st, q
with torch.device(dev):
    f = FluxTransformer2DModel(num_layers=1, num_single_layers=1, guidance_embeds=True)
init(f)
st = FluxKontextTrainStep(f)
side = 32; S_t = side * side; T = 512
ctl = prepare_latent_image_ids(side, side); ctl[:, 0] = 1
emb = dict(image_latents=torch.randn(1, S_t, 64).half().to(dev), control_latents=torch.randn(1, S_t, 64).half().to(dev),
           prompt_embeds=torch.randn(1, T, 4096).half().to(dev), pooled_prompt_embeds=torch.randn(1, 768).half().to(dev),
           text_ids=torch.zeros(T, 3), control_ids=ctl, latent_hw=(side, side))
st.train_step(emb)
report("flux 1+1 blocks", list(f._plans.values())[0])
