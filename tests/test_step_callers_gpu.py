"""GPU: the HIP training-step callers against vectors produced by EXECUTING the bodies of the reference's own step callers
(tests/golden/make_golden.py --step; QwenImageEditTrainer._compute_loss qwen_image_edit_trainer.py:777-861,
FluxKontextLoraTrainer._compute_loss_shared_mode / _compute_loss_multi_resolution_mode flux_kontext_trainer.py:494-796):
sigma lookup, x_t, concat order, slice, target sign, criterion -- bf16 HIP step vs the reference's fp32 run, zero-initialised
adapters (lora_B = 0, peft's init), so the adapted model IS the reference's frozen model."""
import os
import sys

import pytest
import torch
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "qwen-image-finetune_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_qwen_step_vs_executed_reference_compute_loss(golden_dir):
    from common import TINY, fill_weights
    from oracle import qwen_dit as O
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    t = load_file(os.path.join(golden_dir, "ref_step_callers.safetensors"))
    oracle = O.OracleQwenDiT(**TINY)
    fill_weights(oracle, seed=1)
    with torch.device(DEV):
        hip = QwenImageTransformer2DModel(**TINY)
    hip.load_state_dict({k: v.to(BF) for k, v in oracle.state_dict().items()}, strict=True)
    hip.add_adapter(LoraConfig(r=4, lora_alpha=8), "lora_edit")
    B = t["qwen.image_latents"].shape[0]
    emb = dict(image_latents=t["qwen.image_latents"], control_latents=t["qwen.control_latents"], prompt_embeds=t["qwen.prompt_embeds"],
               prompt_embeds_mask=torch.ones(B, t["qwen.prompt_embeds"].shape[1], dtype=torch.int64), img_shapes=[[(1, 4, 6), (1, 4, 6)]] * B)
    step = QwenLoraTrainStep(hip)
    # what the step hands to the DiT (sigma lookup, x_t = (1-s) x0 + s noise, concat order) vs what the reference's body handed over
    packed, target, pe, t_in, S_t = step._prepare(emb, noise=t["qwen.noise"], u=t["qwen.u"])
    assert _rel(packed, t["qwen.dit_hidden_states"]) < 1e-2 and torch.allclose(t_in.cpu().float(), t["qwen.dit_timestep"], atol=1e-6)
    assert _rel(target, t["qwen.noise"] - t["qwen.image_latents"].float()) < 1e-2
    for fused in (True, False):
        if fused:
            loss = step.forward_backward(emb, noise=t["qwen.noise"], u=t["qwen.u"])
            plan = list(hip._plans.values())[0]
            e = _rel(plan.A["out"].view(B, -1, 64)[:, :S_t], t["qwen.pred"][:, :S_t])
            assert e < 2e-2, e
        else:
            loss = step.compute_loss(emb, noise=t["qwen.noise"], u=t["qwen.u"])
        rel = abs(loss.item() - t["qwen.loss"].item()) / t["qwen.loss"].item()
        print(f"qwen step (fused={fused}) vs executed reference _compute_loss: loss {loss.item():.5f} / {t['qwen.loss'].item():.5f} rel {rel:.2e}")
        assert rel < 1e-2


def test_flux_steps_vs_executed_reference_compute_loss(golden_dir):
    from common import FLUX_TINY, fill_weights
    from oracle import flux_dit as FO
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import FluxKontextTrainStep
    t = load_file(os.path.join(golden_dir, "ref_step_callers.safetensors"))
    cfg = dict(FLUX_TINY, guidance_embeds=True)
    oracle = FO.OracleFluxDiT(**cfg)
    fill_weights(oracle, seed=3)
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**cfg)
    hip.load_state_dict({k: v.to(BF) for k, v in oracle.state_dict().items()}, strict=True)
    hip.add_adapter(LoraConfig(r=4, lora_alpha=8), "lora_edit")
    step = FluxKontextTrainStep(hip)
    emb = {k[5:]: v for k, v in t.items() if k.startswith("flux.") and not k.startswith("flux.dit_")}
    S_t = emb["image_latents"].shape[1]
    loss = step.forward_backward(dict(emb, latent_hw=(4, 6)), noise=emb["noise"], t=emb["timestep"])
    plan = list(hip._plans.values())[0]
    e = _rel(plan.A["out"].view(2, -1, 64)[:, :S_t], t["flux.pred"][:, :S_t])
    rel = abs(loss.item() - t["flux.loss"].item()) / t["flux.loss"].item()
    print(f"flux shared step vs executed reference: loss rel {rel:.2e} pred rel {e:.4f}")
    assert rel < 1e-2 and e < 3e-2      # bf16 HIP (fp16-rounded cache inputs) vs the reference's FP32 run of 4 tiny blocks; vs the bf16 oracle the bar is 2e-2
    # multi-resolution caller: ragged batch with a non-square sample (5x3 target, 3x5 control)
    px = t["mr.px_shapes"].tolist()
    lat = [[(h // 16, w // 16) for _, h, w in sh] for sh in px]
    samples = []
    for i in range(2):
        n_t = lat[i][0][0] * lat[i][0][1]
        n_c = sum(a * b for a, b in lat[i][1:])
        samples.append(dict(image_latents=t["mr.image_latents"][i, :n_t], control_latents=t["mr.control_latents"][i, :n_c], hw=lat[i][0],
                            control_hw=lat[i][1:], noise=t[f"mr.noise{i}"], t=t["mr.timestep"][i]))
    T = t["mr.prompt_embeds"].shape[1]
    txt = dict(text_ids=torch.zeros(T, 3), pooled_prompt_embeds=t["mr.pooled_prompt_embeds"], prompt_embeds=t["mr.prompt_embeds"])
    loss = step.forward_backward_multires(samples, txt)
    plan = [p for k, p in hip._plans.items() if "multires" in k][0]
    out = plan.A["out"].view(2, -1, 64)
    n = t["mr.pred"].shape[1]
    e = _rel(out[:, :n], t["mr.pred"])
    rel = abs(loss.item() - t["mr.loss"].item()) / t["mr.loss"].item()
    print(f"flux multi-resolution step vs executed reference: loss rel {rel:.2e} pred rel {e:.4f}")
    assert rel < 1e-2 and e < 3e-2      # bf16 HIP (fp16-rounded cache inputs) vs the reference's FP32 run of 4 tiny blocks; vs the bf16 oracle the bar is 2e-2
    assert out[1, 30:].abs().max().item() == 0.0        # rows past the small sample's 15 + 15 tokens: exactly zero
