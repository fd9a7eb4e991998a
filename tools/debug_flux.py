import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_util import *
from common import FLUX_TINY, fill_weights
from oracle import flux_dit as FO, qwen_dit as O
from qflux_amd.models import FluxTransformer2DModel
from qflux_amd.modules import LoraConfig
from qflux_amd.trainer import FluxKontextTrainStep
cfg = dict(FLUX_TINY, guidance_embeds=True, joint_attention_dim=64)
B, (h, w), T, r = 2, (4, 6), 7, 4
oracle = FO.OracleFluxDiT(**cfg)
O.add_lora(oracle, r=r, lora_alpha=2 * r, adapter_name="lora_edit")
fill_weights(oracle, seed=5)
for n, p in oracle.named_parameters():
    if "lora" not in n: p.data = p.data.to(BF)
with torch.device("cuda:0"):
    hip = FluxTransformer2DModel(**cfg)
hip.add_adapter(LoraConfig(r=r, lora_alpha=2 * r), "lora_edit")
hip.load_state_dict(oracle.state_dict(), strict=True)
g = torch.Generator().manual_seed(31); S_t = h * w
ctl_ids = FO.prepare_latent_image_ids(h, w); ctl_ids[:, 0] = 1
emb = dict(image_latents=torch.randn(B, S_t, 64, generator=g).half(), control_latents=torch.randn(B, S_t, 64, generator=g).half(),
           control_ids=ctl_ids, text_ids=torch.zeros(T, 3), latent_hw=(h, w),
           pooled_prompt_embeds=torch.randn(B, cfg["pooled_projection_dim"], generator=g).half(),
           prompt_embeds=torch.randn(B, T, cfg["joint_attention_dim"], generator=g).half())
noise = torch.randn(B, S_t, 64, generator=g).to(BF); t = torch.tensor([0.7109, 0.1611]).to(BF)
cap = {}
def hook(name):
    def f(m, i, o): cap[name] = o
    return f
oracle.time_text_embed.register_forward_hook(hook("temb"))
oracle.x_embedder.register_forward_hook(hook("x0")); oracle.context_embedder.register_forward_hook(hook("c0"))
for i, b in enumerate(oracle.transformer_blocks): b.register_forward_hook(hook(f"d{i}"))
for i, b in enumerate(oracle.single_transformer_blocks): b.register_forward_hook(hook(f"s{i}"))
oracle.norm_out.register_forward_hook(hook("xn"))
emb_o = dict(emb, control_latents=emb["control_latents"].to(BF))
loss_o, pred_o = FO.flux_compute_loss(oracle, emb_o, noise, t, BF, return_pred=True)
step = FluxKontextTrainStep(hip)
loss_h = step.forward_backward(emb, noise=noise, t=t)
torch.cuda.synchronize()
plan = list(hip._plans.values())[0]; A = plan.A
S = T + 2 * S_t; D = 128
def cmp(name, got, ref): print(f"{name:12s} rel={relmax(got, ref):.3e}")
cmp("temb", A["temb"][0], cap["temb"])
cmp("x0", A["X"]["img"][0].view(B, -1, D), cap["x0"]); cmp("c0", A["X"]["txt"][0].view(B, T, D), cap["c0"])
cmp("d0.img", A["X"]["img"][1].view(B, -1, D), cap["d0"][1]); cmp("d0.txt", A["X"]["txt"][1].view(B, T, D), cap["d0"][0])
J0 = A["J"][0].view(B, S, D)
cmp("d1.img", J0[:, T:], cap["d1"][1]); cmp("d1.txt", J0[:, :T], cap["d1"][0])
for i in range(2):
    Ji = A["J"][i + 1].view(B, S, D)
    cmp(f"s{i}.img", Ji[:, T:], cap[f"s{i}"][1]); cmp(f"s{i}.txt", Ji[:, :T], cap[f"s{i}"][0])
cmp("xn", A["xn_out"].view(B, -1, D), cap["xn"])
cmp("pred", A["out"].view(B, -1, 64)[:, :S_t], pred_o)
print("loss", loss_o.item(), loss_h.item())
