"""CPU: the oracle reproduces the vectors captured from the reference itself (tests/golden/make_golden.py)."""
import ast
import os

import torch
from safetensors import safe_open
from safetensors.torch import load_file

from common import TINY, fill_weights, weight_checksum
from oracle import qwen_dit as O


def _meta(path):
    with safe_open(path, "pt") as f:
        return f.metadata()


def test_forward_matches_reference_vectors(golden_dir):
    p = os.path.join(golden_dir, "qwen_tiny_fwd.safetensors")
    t, meta = load_file(p), _meta(p)
    model = O.OracleQwenDiT(**TINY)
    fill_weights(model, seed=1)
    assert torch.equal(weight_checksum(model), t["w.checksum"])
    shapes = ast.literal_eval(meta["img_shapes"])
    B = t["in.hidden_states"].shape[0]
    out = model(hidden_states=t["in.hidden_states"], encoder_hidden_states=t["in.encoder_hidden_states"],
                encoder_hidden_states_mask=t["in.mask"], timestep=t["in.timestep"],
                img_shapes=[shapes] * B, txt_seq_lens=[int(meta["txt_len"])] * B)[0]
    assert (out - t["out.sample"]).abs().max() < 1e-5


def test_grads_match_reference_vectors(golden_dir):
    t = load_file(os.path.join(golden_dir, "qwen_tiny_fwd.safetensors"))
    g = load_file(os.path.join(golden_dir, "qwen_tiny_grad.safetensors"))
    meta = _meta(os.path.join(golden_dir, "qwen_tiny_fwd.safetensors"))
    model = O.OracleQwenDiT(**TINY)
    fill_weights(model, seed=1)
    shapes = ast.literal_eval(meta["img_shapes"])
    x = t["in.hidden_states"].clone().requires_grad_(True)
    out = model(hidden_states=x, encoder_hidden_states=t["in.encoder_hidden_states"],
                encoder_hidden_states_mask=t["in.mask"], timestep=t["in.timestep"],
                img_shapes=[shapes] * 2, txt_seq_lens=[int(meta["txt_len"])] * 2)[0]
    loss = ((out - t["in.target"]) ** 2).mean()
    gx, gw = torch.autograd.grad(loss, [x, model.transformer_blocks[0].attn.to_q.weight])
    assert abs(loss.item() - g["loss"].item()) < 1e-6
    assert (gx - g["grad.hidden_states"]).abs().max() < 1e-6
    assert (gw - g["grad.blocks0_to_q_weight"]).abs().max() < 1e-6


def test_rope_tables_match_reference(golden_dir):
    p = os.path.join(golden_dir, "qwen_rope.safetensors")
    t, meta = load_file(p), _meta(p)
    for name in ("a", "b", "c"):
        shapes, tl = ast.literal_eval(meta[name])
        v, x = O.qwen_rope_tables(shapes[0], tl[0], (16, 56, 56))
        assert torch.allclose(torch.view_as_real(v), t[f"{name}.vid"], atol=1e-6)
        assert torch.allclose(torch.view_as_real(x), t[f"{name}.txt"], atol=1e-6)
    # known answer from SURVEY 8(a) a4: rotate [0..7] by pi/2 -> [-1,0,-3,2,-5,4,-7,6]
    xk = torch.arange(8.0).view(1, 1, 1, 8)
    fk = torch.polar(torch.ones(1, 4), torch.full((1, 4), torch.pi / 2))
    out = O.apply_rope_complex(xk, fk)
    assert torch.allclose(out, t["kat.out"], atol=1e-6)
    assert torch.allclose(out.flatten(), torch.tensor([-1., 0, -3, 2, -5, 4, -7, 6]), atol=1e-5)


def test_lora_step_vectors(golden_dir):
    p = os.path.join(golden_dir, "qwen_tiny_lora_step.safetensors")
    t, meta = load_file(p), _meta(p)
    model = O.OracleQwenDiT(**TINY)
    names = O.add_lora(model, r=int(meta["r"]), lora_alpha=float(meta["lora_alpha"]), adapter_name=meta["adapter"])
    assert names == ast.literal_eval(meta["targets"])
    fill_weights(model, seed=2)
    assert torch.equal(weight_checksum(model), t["w.checksum"])
    emb = dict(image_latents=t["in.image_latents"], control_latents=t["in.control_latents"],
               prompt_embeds=t["in.prompt_embeds"], prompt_embeds_mask=torch.ones(2, 5, dtype=torch.int64),
               img_shapes=[[(1, 4, 6), (1, 4, 6)]] * 2)
    loss, pred = O.qwen_compute_loss(model, emb, t["in.noise"], t["in.u"], torch.float32, return_pred=True)
    loss.backward()
    assert abs(loss.item() - t["out.loss"].item()) < 1e-6
    assert (pred - t["out.pred"]).abs().max() < 1e-5
    n = 0
    for pn, prm in model.named_parameters():
        if "lora" in pn:
            assert prm.requires_grad and (prm.grad - t["g." + pn]).abs().max() < 1e-6, pn
            n += 1
        else:
            assert not prm.requires_grad
    assert n == 2 * 4 * TINY["num_layers"]


def test_flux_forward_and_grads_match_reference_vectors(golden_dir):
    """FLUX oracle vs vectors captured from the reference's own transformer_flux.py (guidance_embeds=True)."""
    from common import FLUX_TINY
    from oracle import flux_dit as FO
    t = load_file(os.path.join(golden_dir, "flux_tiny_fwd.safetensors"))
    cfg = dict(FLUX_TINY, guidance_embeds=True)
    m = FO.OracleFluxDiT(**cfg)
    fill_weights(m, seed=3)
    assert torch.equal(weight_checksum(m), t["w.checksum"])
    x = t["in.hidden_states"].clone().requires_grad_(True)
    out = m(hidden_states=x, encoder_hidden_states=t["in.encoder_hidden_states"], pooled_projections=t["in.pooled"],
            timestep=t["in.timestep"], img_ids=t["in.img_ids"], txt_ids=t["in.txt_ids"], guidance=torch.ones(2))[0]
    assert (out - t["out.sample"]).abs().max() < 1e-5
    loss = ((out - t["in.target"]) ** 2).mean()
    gx, gw = torch.autograd.grad(loss, [x, m.single_transformer_blocks[0].attn.to_q.weight])
    assert (gx - t["grad.hidden_states"]).abs().max() < 1e-6
    assert (gw - t["grad.single0_to_q_weight"]).abs().max() < 1e-6


def test_flux_rope_identity_on_text_and_image_index():
    from oracle import flux_dit as FO
    lat = FO.prepare_latent_image_ids(3, 4)
    ctl = lat.clone(); ctl[:, 0] = 1
    ids = torch.cat([torch.zeros(5, 3), lat, ctl])
    cos, sin = FO.flux_rope_tables(ids, (16, 56, 56))
    assert cos.shape == (5 + 24, 128)
    assert torch.all(cos[:5] == 1) and torch.all(sin[:5] == 0)          # text ids are zero: identity rotation
    assert torch.equal(cos[5:17, 16:], cos[17:, 16:])                    # control image differs only on axis 0
    assert not torch.equal(cos[5:17, :16], cos[17:, :16])
    assert torch.equal(cos[:, 0::2], cos[:, 1::2])                       # repeat_interleave_real


def test_flux_multires_matches_reference_custom_vectors(golden_dir):
    """Ragged right-padded batch through the oracle vs vectors from the reference's transformer_flux_custom.py:
    per-sample RoPE, additive key mask, padded rows zeroed after every block (flux_kontext_trainer.py:579-760 caller)."""
    from common import FLUX_TINY
    from oracle import flux_dit as FO
    t = load_file(os.path.join(golden_dir, "flux_tiny_multires.safetensors"))
    m = FO.OracleFluxDiT(**dict(FLUX_TINY, guidance_embeds=True))
    fill_weights(m, seed=3)
    assert torch.equal(weight_checksum(m), t["w.checksum"])
    x = t["in.hidden_states"].clone().requires_grad_(True)
    full = t["in.attention_mask"].bool()
    out = m(hidden_states=x, encoder_hidden_states=t["in.encoder_hidden_states"], pooled_projections=t["in.pooled"],
            timestep=t["in.timestep"], img_ids=t["in.img_ids"], txt_ids=t["in.txt_ids"], guidance=torch.ones(2), attention_mask=full)[0]
    assert (out - t["out.sample"]).abs().max() < 1e-5
    T = t["in.txt_ids"].shape[0]
    pad = ~full[:, T:]
    assert pad.any() and out[pad].abs().max() == 0
    (gx,) = torch.autograd.grad(((out - t["in.target"]) ** 2).mean(), [x])
    assert (gx - t["grad.hidden_states"]).abs().max() < 1e-6
    assert gx[pad].abs().max() == 0


def test_attention_mask_mse_loss_semantics():
    """AttentionMaskMseLoss(reduction='mean') (losses/attention_mask_loss.py:146-226): masked per-token channel mean,
    summed and divided by the number of valid tokens."""
    from oracle import flux_dit as FO
    g = torch.Generator().manual_seed(3)
    p, q = torch.randn(2, 5, 8, generator=g), torch.randn(2, 5, 8, generator=g)
    mk = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], dtype=torch.bool)
    want = sum(((p[b, :n] - q[b, :n]) ** 2).mean(dim=1).sum() for b, n in ((0, 3), (1, 5))) / 8
    got = FO.attention_mask_mse_loss(p, q, mk)
    assert abs(got.item() - want.item()) < 1e-6


def test_criteria_match_reference_loss_classes(golden_dir):
    """Vectors produced by the reference's own MseLoss / MaskEditLoss / AttentionMaskMseLoss / map_mask_to_latent."""
    from oracle import flux_dit as FO
    from oracle import qwen_dit as O
    t = load_file(os.path.join(golden_dir, "losses.safetensors"))
    pr, tg, em, am = t["pred"], t["target"], t["edit_mask"], t["attention_mask"].bool()
    assert abs(O.mse_loss(pr, tg).item() - t["mse"].item()) < 1e-6
    assert abs(O.mask_edit_loss(pr, tg, em, 2.0, 1.0).item() - t["mask_edit_2_1"].item()) < 1e-6
    assert abs(O.mask_edit_loss(pr, tg, None, 2.0, 1.0).item() - t["mask_edit_none"].item()) < 1e-6
    assert abs(FO.attention_mask_mse_loss(pr, tg, am).item() - t["attn_mask_mse"].item()) < 1e-6
    assert torch.equal(O.map_mask_to_latent(t["pixel_mask"]), t["latent_mask"])


def test_qwen_multires_matches_reference_custom_vectors(golden_dir):
    """Ragged right-padded batch (per-sample shape lists AND ragged text lengths) through the oracle vs vectors from the
    reference's transformer_qwen_custom.py: per-sample RoPE placement, additive key mask, padded rows zeroed."""
    from oracle import qwen_dit as O
    t = load_file(os.path.join(golden_dir, "qwen_tiny_multires.safetensors"))
    m = O.OracleQwenDiT(**TINY)
    fill_weights(m, seed=1)
    assert torch.equal(weight_checksum(m), t["w.checksum"])
    shapes = [[tuple(int(v) for v in s) for s in sh] for sh in t["in.shapes"].tolist()]
    lens = [int(v) for v in t["in.txt_lens"]]
    full = t["in.attention_mask"].bool()
    x = t["in.hidden_states"].clone().requires_grad_(True)
    out = m(hidden_states=x, encoder_hidden_states=t["in.encoder_hidden_states"], timestep=t["in.timestep"], img_shapes=shapes,
            txt_seq_lens=lens, attention_mask=full)[0]
    assert (out - t["out.sample"]).abs().max() < 1e-5
    T = t["in.encoder_hidden_states"].shape[1]
    pad = ~full[:, T:]
    assert pad.any() and out[pad].abs().max() == 0
    (gx,) = torch.autograd.grad(((out - t["in.target"]) ** 2).mean(), [x])
    assert (gx - t["grad.hidden_states"]).abs().max() < 1e-6


# ---------------------------------------------------------------------------------------------------------------------------
# Step callers: vectors produced by EXECUTING the bodies of the reference's own
# QwenImageEditTrainer._compute_loss / FluxKontextLoraTrainer._compute_loss_shared_mode / _compute_loss_multi_resolution_mode
# (qwen_image_edit_trainer.py:777-861, flux_kontext_trainer.py:494-796) on a stub trainer -- make_golden.py --step.
def _step_vectors(golden_dir):
    return load_file(os.path.join(golden_dir, "ref_step_callers.safetensors"))


def test_qwen_step_caller_matches_the_executed_reference_body(golden_dir):
    t = _step_vectors(golden_dir)
    model = O.OracleQwenDiT(**TINY)
    fill_weights(model, seed=1)
    assert torch.equal(weight_checksum(model), t["qwen.w_checksum"])
    B = t["qwen.image_latents"].shape[0]
    emb = dict(image_latents=t["qwen.image_latents"], control_latents=t["qwen.control_latents"], prompt_embeds=t["qwen.prompt_embeds"],
               prompt_embeds_mask=torch.ones(B, t["qwen.prompt_embeds"].shape[1], dtype=torch.int64), img_shapes=[[(1, 4, 6), (1, 4, 6)]] * B)
    loss, pred = O.qwen_compute_loss(model, emb, t["qwen.noise"], t["qwen.u"], torch.float32, return_pred=True)
    assert abs(loss.item() - t["qwen.loss"].item()) < 1e-6
    assert (pred - t["qwen.pred"][:, : pred.shape[1]]).abs().max() < 1e-5
    # what reached the DiT: sigma lookup (u -> index -> timestep -> sigma), x_t, concat order, timestep / 1000
    ts, sig = O.flowmatch_sigmas()
    idx = (t["qwen.u"] * 1000).long()
    assert torch.equal(t["qwen.dit_timestep"], ts[idx] / 1000)
    s = sig[idx].view(B, 1, 1)
    x_t = (1.0 - s) * t["qwen.image_latents"].float() + s * t["qwen.noise"]
    assert torch.equal(t["qwen.dit_hidden_states"], torch.cat([x_t, t["qwen.control_latents"].float()], dim=1))


def test_flux_step_callers_match_the_executed_reference_bodies(golden_dir):
    from common import FLUX_TINY
    from oracle import flux_dit as FO
    t = _step_vectors(golden_dir)
    cfg = dict(FLUX_TINY, guidance_embeds=True)
    fo = FO.OracleFluxDiT(**cfg)
    fill_weights(fo, seed=3)
    assert torch.equal(weight_checksum(fo), t["flux.w_checksum"]) and torch.equal(weight_checksum(fo), t["mr.w_checksum"])
    emb = {k[5:]: v for k, v in t.items() if k.startswith("flux.") and not k.startswith("flux.dit_")}
    S_t = emb["image_latents"].shape[1]
    loss, pred = FO.flux_compute_loss(fo, dict(emb, latent_hw=(4, 6)), emb["noise"], emb["timestep"], torch.float32, return_pred=True)
    assert abs(loss.item() - t["flux.loss"].item()) < 1e-6 and (pred - t["flux.pred"][:, :S_t]).abs().max() < 1e-5
    assert torch.equal(t["flux.dit_guidance"], torch.ones(2))
    # multi-resolution caller: ragged batch with a NON-SQUARE sample (5x3 target + 3x5 control tokens)
    px = t["mr.px_shapes"].tolist()
    lat = [[(h // 16, w // 16) for _, h, w in sh] for sh in px]
    samples = []
    for i in range(2):
        n_t = lat[i][0][0] * lat[i][0][1]
        n_c = sum(a * b for a, b in lat[i][1:])
        samples.append(dict(image_latents=t["mr.image_latents"][i, :n_t], control_latents=t["mr.control_latents"][i, :n_c], hw=lat[i][0],
                            control_hw=lat[i][1:], noise=t[f"mr.noise{i}"], t=t["mr.timestep"][i]))
    T = t["mr.prompt_embeds"].shape[1]
    loss, pred = FO.flux_compute_loss_multires(fo, samples, dict(text_ids=torch.zeros(T, 3), pooled_prompt_embeds=t["mr.pooled_prompt_embeds"],
                                                                 prompt_embeds=t["mr.prompt_embeds"]), torch.float32, return_pred=True)
    assert abs(loss.item() - t["mr.loss"].item()) < 1e-6 and (pred - t["mr.pred"]).abs().max() < 1e-5
    assert not t["mr.dit_attention_mask"][1, T + 30:].any() and t["mr.dit_attention_mask"][1, : T + 30].all()   # 15 + 15 valid image rows
