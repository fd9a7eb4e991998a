"""Pure-PyTorch CPU restatement of the FLUX(-Kontext) DiT LoRA training step.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Follows src/qflux/models/transformer_flux.py (paths relative to /root/reference); sub-module /
parameter names equal the diffusers FluxTransformer2DModel state-dict keys.
Third-party primitives (AdaLayerNormZero/-Single, CombinedTimestep*Embeddings, apply_rotary_emb,
get_1d_rotary_pos_embed) are restated from diffusers (parity unpinned for that half).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .qwen_dit import OracleFeedForward, _AdaLNContinuous, _TimestepEmbedder, mse_loss, timestep_sinusoid  # noqa: F401


def flux_rope_tables(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0):
    """FluxPosEmbed.forward (transformer_flux.py:533-554) + diffusers get_1d_rotary_pos_embed(use_real=True,
    repeat_interleave_real=True, freqs_dtype=float64): cos/sin [S, sum(axes)] fp32, each frequency repeated twice."""
    pos = ids.float()
    cos_out, sin_out = [], []
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64, device=pos.device)[: d // 2] / d))
        ang = torch.outer(pos[:, i].to(torch.float64), freqs)
        cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def apply_rotary_emb_real(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """diffusers apply_rotary_emb(x, (cos, sin), sequence_dim=1), use_real_unbind_dim=-1: x [B,S,H,D].
    cos/sin [S,D] (shared) or [B,S,D] (per-sample RoPE of transformer_flux_custom.py:194-212)."""
    if cos.ndim == 3:
        cos, sin = cos[:, :, None, :], sin[:, :, None, :]
    else:
        cos, sin = cos[None, :, None, :], sin[None, :, None, :]
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


class _FluxAttention(nn.Module):
    """FluxAttention holder (transformer_flux.py:304-363): torch.nn.RMSNorm q/k norms."""

    def __init__(self, dim, heads, dim_head, eps=1e-6, added=True, pre_only=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.pre_only = pre_only
        self.norm_q = nn.RMSNorm(dim_head, eps=eps)
        self.norm_k = nn.RMSNorm(dim_head, eps=eps)
        self.to_q = nn.Linear(dim, inner, bias=True)
        self.to_k = nn.Linear(dim, inner, bias=True)
        self.to_v = nn.Linear(dim, inner, bias=True)
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(inner, dim, bias=True), nn.Dropout(0.0)])
        self.added = added
        if added:
            self.norm_added_q = nn.RMSNorm(dim_head, eps=eps)
            self.norm_added_k = nn.RMSNorm(dim_head, eps=eps)
            self.add_q_proj = nn.Linear(dim, inner, bias=True)
            self.add_k_proj = nn.Linear(dim, inner, bias=True)
            self.add_v_proj = nn.Linear(dim, inner, bias=True)
            self.to_add_out = nn.Linear(inner, dim, bias=True)


def flux_attention(attn: _FluxAttention, x, ctx, rope, key_mask=None):
    """FluxAttnProcessor.__call__ (transformer_flux.py:102-166): RoPE is applied AFTER the [text|image] concat."""
    H = attn.heads
    q, k, v = (f(x).unflatten(-1, (H, -1)) for f in (attn.to_q, attn.to_k, attn.to_v))
    q, k = attn.norm_q(q), attn.norm_k(k)
    if ctx is not None:
        cq, ck, cv = (f(ctx).unflatten(-1, (H, -1)) for f in (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj))
        cq, ck = attn.norm_added_q(cq), attn.norm_added_k(ck)
        q, k, v = torch.cat([cq, q], 1), torch.cat([ck, k], 1), torch.cat([cv, v], 1)
    if rope is not None:
        q, k = apply_rotary_emb_real(q, *rope), apply_rotary_emb_real(k, *rope)
    am = key_mask[:, None, None, :].to(q.dtype) if key_mask is not None else None
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=am)
    o = o.transpose(1, 2).flatten(2, 3).to(q.dtype)
    if ctx is not None:
        T = ctx.shape[1]
        c_o, x_o = o[:, :T], o[:, T:]
        return attn.to_out[0](x_o), attn.to_add_out(c_o)
    return o


class _AdaLNZero(nn.Module):
    """diffusers AdaLayerNormZero(dim): emb = linear(silu(temb)) -> 6 chunks."""

    def __init__(self, dim, n=6):
        super().__init__()
        self.linear = nn.Linear(dim, n * dim, bias=True)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.n = n

    def forward(self, x, emb):
        emb = self.linear(F.silu(emb))
        ch = emb.chunk(self.n, dim=1)
        shift, scale, gate = ch[0], ch[1], ch[2]
        x = self.norm(x) * (1 + scale[:, None]) + shift[:, None]
        return (x, gate) + tuple(ch[3:])


class OracleFluxBlock(nn.Module):
    """FluxTransformerBlock (transformer_flux.py:439-523)."""

    def __init__(self, dim, heads, dim_head):
        super().__init__()
        self.norm1 = _AdaLNZero(dim)
        self.norm1_context = _AdaLNZero(dim)
        self.attn = _FluxAttention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = OracleFeedForward(dim)
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = OracleFeedForward(dim)

    def forward(self, x, ctx, temb, rope, key_mask=None):
        nx, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(x, temb)
        nc, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(ctx, temb)
        a, ca = flux_attention(self.attn, nx, nc, rope, key_mask)
        x = x + gate_msa.unsqueeze(1) * a
        nx = self.norm2(x) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        x = x + gate_mlp.unsqueeze(1) * self.ff(nx)
        ctx = ctx + c_gate_msa.unsqueeze(1) * ca
        nc = self.norm2_context(ctx) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        ctx = ctx + c_gate_mlp.unsqueeze(1) * self.ff_context(nc)
        if ctx.dtype == torch.float16:
            ctx = ctx.clip(-65504, 65504)
        return ctx, x


class OracleFluxSingleBlock(nn.Module):
    """FluxSingleTransformerBlock (transformer_flux.py:385-436)."""

    def __init__(self, dim, heads, dim_head, mlp_ratio=4.0):
        super().__init__()
        self.norm = _AdaLNZero(dim, n=3)
        self.proj_mlp = nn.Linear(dim, int(dim * mlp_ratio))
        self.proj_out = nn.Linear(dim + int(dim * mlp_ratio), dim)
        self.attn = _FluxAttention(dim, heads, dim_head, added=False, pre_only=True)

    def forward(self, x, ctx, temb, rope, key_mask=None):
        T = ctx.shape[1]
        h = torch.cat([ctx, x], dim=1)
        res = h
        nh, gate = self.norm(h, temb)
        mlp = F.gelu(self.proj_mlp(nh), approximate="tanh")
        a = flux_attention(self.attn, nh, None, rope, key_mask)
        h = torch.cat([a, mlp], dim=2)
        h = res + gate.unsqueeze(1) * self.proj_out(h)
        if h.dtype == torch.float16:
            h = h.clip(-65504, 65504)
        return h[:, :T], h[:, T:]


class _TextProj(nn.Module):
    """diffusers PixArtAlphaTextProjection(in, hidden, act_fn='silu')."""

    def __init__(self, in_features, hidden):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden)
        self.linear_2 = nn.Linear(hidden, hidden)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class _CombinedEmb(nn.Module):
    """diffusers CombinedTimestep(Guidance)TextProjEmbeddings: Timesteps(256, flip_sin_to_cos=True, shift 0, scale 1)."""

    def __init__(self, dim, pooled_dim, guidance):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedder(256, dim)
        if guidance:
            self.guidance_embedder = _TimestepEmbedder(256, dim)
        self.text_embedder = _TextProj(pooled_dim, dim)
        self.guidance = guidance

    def forward(self, timestep, guidance, pooled):
        t = self.timestep_embedder(timestep_sinusoid(timestep, 256, scale=1.0).to(pooled.dtype))
        if self.guidance:
            t = t + self.guidance_embedder(timestep_sinusoid(guidance, 256, scale=1.0).to(pooled.dtype))
        return t + self.text_embedder(pooled)


class OracleFluxDiT(nn.Module):
    """FluxTransformer2DModel (transformer_flux.py:557-828)."""

    def __init__(self, patch_size=1, in_channels=64, out_channels=None, num_layers=19, num_single_layers=38,
                 attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
                 guidance_embeds=False, axes_dims_rope=(16, 56, 56)):
        super().__init__()
        D = num_attention_heads * attention_head_dim
        self.inner_dim = D
        self.axes_dims_rope = tuple(axes_dims_rope)
        self.guidance_embeds = guidance_embeds
        self.time_text_embed = _CombinedEmb(D, pooled_projection_dim, guidance_embeds)
        self.context_embedder = nn.Linear(joint_attention_dim, D)
        self.x_embedder = nn.Linear(in_channels, D)
        self.transformer_blocks = nn.ModuleList([OracleFluxBlock(D, num_attention_heads, attention_head_dim) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList(
            [OracleFluxSingleBlock(D, num_attention_heads, attention_head_dim) for _ in range(num_single_layers)])
        self.norm_out = _AdaLNContinuous(D)
        self.proj_out = nn.Linear(D, patch_size * patch_size * (out_channels or in_channels), bias=True)

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, img_ids=None,
                txt_ids=None, guidance=None, joint_attention_kwargs=None, return_dict=False, key_mask=None, attention_mask=None):
        if attention_mask is not None:
            return self.forward_multires(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                                         guidance, attention_mask)
        x = self.x_embedder(hidden_states)
        timestep = timestep.to(x.dtype) * 1000           # (:729-730) computed in the model dtype
        if guidance is not None:
            guidance = guidance.to(x.dtype) * 1000
        temb = self.time_text_embed(timestep, guidance, pooled_projections)
        ctx = self.context_embedder(encoder_hidden_states)
        ids = torch.cat((txt_ids, img_ids), dim=0)
        rope = flux_rope_tables(ids, self.axes_dims_rope)
        for blk in self.transformer_blocks:
            ctx, x = blk(x, ctx, temb, rope, key_mask)
        for blk in self.single_transformer_blocks:
            ctx, x = blk(x, ctx, temb, rope, key_mask)
        x = self.norm_out(x, temb)
        return (self.proj_out(x),)


def _forward_multires(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance,
                      attention_mask):
    """Multi-resolution forward of the custom model (src/qflux/models/transformer_flux_custom.py:372-741): bool
    attention_mask [B, T+S_max]; per-sample RoPE with identity rotation on padding (:537-560); additive -inf key mask
    (:585-616); padded image tokens zeroed after x_embedder (:427-442), after EVERY block (:648-660,696-708) and in
    the output (:724-733).  The text stream is never masked."""
    B, S_i = hidden_states.shape[:2]
    T = txt_ids.shape[0]
    mask = attention_mask if attention_mask.dtype == torch.bool else attention_mask > 0
    img_mask = mask[:, T:T + S_i]
    x = self.x_embedder(hidden_states)
    if not img_mask.all():
        x = x.masked_fill(~img_mask.unsqueeze(-1), 0)
    timestep = timestep.to(x.dtype) * 1000
    if guidance is not None:
        guidance = guidance.to(x.dtype) * 1000
    temb = self.time_text_embed(timestep, guidance, pooled_projections)
    ctx = self.context_embedder(encoder_hidden_states)
    if img_ids.ndim == 2:
        img_ids = img_ids.unsqueeze(0).expand(B, -1, -1)
    Dh = sum(self.axes_dims_rope)
    cos = torch.ones(B, T + S_i, Dh)
    sin = torch.zeros(B, T + S_i, Dh)
    for b in range(B):
        n = int(img_mask[b].sum().item())
        c, s_ = flux_rope_tables(torch.cat([txt_ids.float(), img_ids[b, :n].float()], dim=0), self.axes_dims_rope)
        cos[b, : T + n], sin[b, : T + n] = c, s_
    rope = (cos, sin)
    km = torch.zeros(B, T + S_i, dtype=x.dtype).masked_fill(~mask[:, : T + S_i], float("-inf"))
    for blk in list(self.transformer_blocks) + list(self.single_transformer_blocks):
        ctx, x = blk(x, ctx, temb, rope, km)
        if not img_mask.all():
            x = x * img_mask.unsqueeze(-1)
    x = self.norm_out(x, temb)
    out = self.proj_out(x)
    if not img_mask.all():
        out = out.masked_fill(~img_mask.unsqueeze(-1), 0)
    return (out,)


OracleFluxDiT.forward_multires = _forward_multires


def attention_mask_mse_loss(model_pred, target, attention_mask, edit_mask=None, fg=1.0, bg=1.0, eps=1e-12):
    """AttentionMaskMseLoss.forward, reduction='mean' (src/qflux/losses/attention_mask_loss.py:146-226)."""
    el = (model_pred.float() - target.float()) ** 2
    if edit_mask is None:
        ew = torch.ones_like(attention_mask, dtype=torch.float32).unsqueeze(-1)
    else:
        m = edit_mask.float()
        ew = (m * fg + (1.0 - m) * bg).unsqueeze(-1)
    tok = (el * ew * attention_mask.float().unsqueeze(-1)).mean(dim=2)
    nv = attention_mask.sum().to(model_pred.dtype)
    return tok.sum() / (nv + eps)


def flux_compute_loss_multires(dit: nn.Module, samples: list, txt: dict, dtype: torch.dtype, return_pred=False):
    """FluxKontextLoraTrainer._compute_loss_multi_resolution_mode (flux_kontext_trainer.py:579-796) with (noise, t) injected.
    samples[i] = dict(image_latents [n_t,64], control_latents [n_c,64], hw=(h,w), control_hw=[(h,w),...], noise [n_t,64], t scalar);
    txt = dict(text_ids [T,3], pooled_prompt_embeds [B,P], prompt_embeds [B,T,J])."""
    B = len(samples)
    seqs, ids, tmask_len = [], [], []
    for smp in samples:
        h, w = smp["hw"]
        lat_ids = prepare_latent_image_ids(h, w, dtype)
        cids = []
        for j, (ch, cw) in enumerate(smp["control_hw"]):
            ci = prepare_latent_image_ids(ch, cw, dtype)
            ci[..., 0] = j + 1
            cids.append(ci)
        t_ = smp["t"].to(dtype).reshape(1, 1)
        x_t = (1.0 - t_) * smp["image_latents"] + t_ * smp["noise"].to(dtype)
        seqs.append(torch.cat([x_t, smp["control_latents"]], dim=0))
        ids.append(torch.cat([lat_ids] + cids, dim=0))
        tmask_len.append(smp["image_latents"].shape[0])
    S_max = max(s.shape[0] for s in seqs)
    n_t_max = max(tmask_len)
    T = txt["text_ids"].shape[0]
    inp = torch.zeros(B, S_max, 64)
    idb = torch.zeros(B, S_max, 3)
    full = torch.ones(B, T + S_max, dtype=torch.bool)
    lat_mask = torch.zeros(B, n_t_max, dtype=dtype)
    noise_in = torch.zeros(B, n_t_max, 64, dtype=dtype)
    x0_pad = torch.zeros(B, n_t_max, 64)
    for i, (sq, di) in enumerate(zip(seqs, ids)):
        inp[i, : sq.shape[0]] = sq.float()
        idb[i, : di.shape[0]] = di.float()
        full[i, T + sq.shape[0]:] = False
        lat_mask[i, : tmask_len[i]] = 1
        noise_in[i, : tmask_len[i]] = samples[i]["noise"].to(dtype)
        x0_pad[i, : tmask_len[i]] = samples[i]["image_latents"].float()
    timestep = torch.stack([s["t"].to(dtype).reshape(()) for s in samples])
    guidance = torch.ones((B,)).to(dtype) if getattr(dit, "guidance_embeds", False) else None
    pred = dit(hidden_states=inp.to(dtype), timestep=timestep, guidance=guidance, pooled_projections=txt["pooled_prompt_embeds"].to(dtype),
               encoder_hidden_states=txt["prompt_embeds"].to(dtype), txt_ids=txt["text_ids"], img_ids=idb, attention_mask=full,
               return_dict=False)[0]
    pred = pred[:, :n_t_max]
    target = noise_in - x0_pad.to(dtype)
    loss = attention_mask_mse_loss(pred, target, lat_mask)
    return (loss, pred) if return_pred else loss


def prepare_latent_image_ids(height: int, width: int, dtype=torch.float32) -> torch.Tensor:
    """flux_kontext_trainer.py:871-883 (_prepare_latent_image_ids)."""
    ids = torch.zeros(height, width, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(height)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(width)[None, :]
    return ids.reshape(height * width, 3).to(dtype)


def flux_compute_loss(dit: nn.Module, emb: dict, noise: torch.Tensor, t: torch.Tensor, dtype: torch.dtype, return_pred=False):
    """FluxKontextLoraTrainer._compute_loss_shared_mode (flux_kontext_trainer.py:494-577) with (noise, t) injected.
    emb: image_latents [B,S_t,64], control_latents, control_ids [S_c,3], text_ids [T,3], pooled_prompt_embeds,
    prompt_embeds, latent_hw=(h,w) in packed-latent units."""
    x0 = emb["image_latents"]
    with torch.no_grad():
        t_ = t.unsqueeze(1).unsqueeze(1)
        x_t = (1.0 - t_) * x0 + t_ * noise
        h, w = emb["latent_hw"]
        latent_ids = prepare_latent_image_ids(h, w, dtype).to(x0.device)      # (.to(device): the checker also runs on the GPU, tests/test_fulldepth_gpu.py)
        inp = torch.cat([x_t, emb["control_latents"]], dim=1)
        ids = torch.cat([latent_ids, emb["control_ids"].to(dtype)], dim=0)
    guidance = torch.ones((noise.shape[0],), device=x0.device).to(dtype) if getattr(dit, "guidance_embeds", False) else None
    pred = dit(hidden_states=inp.to(dtype), timestep=t.to(dtype), guidance=guidance,
               pooled_projections=emb["pooled_prompt_embeds"].to(dtype), encoder_hidden_states=emb["prompt_embeds"].to(dtype),
               txt_ids=emb["text_ids"], img_ids=ids, joint_attention_kwargs={}, return_dict=False)[0]
    pred = pred[:, : x0.size(1)]
    target = noise - x0
    loss = F.mse_loss(pred, target.to(pred.dtype), reduction="mean")   # MseLoss with weighting=None (mse_loss.py:68-70)
    return (loss, pred) if return_pred else loss


def flux_sample(dit: nn.Module, emb: dict, dtype: torch.dtype):
    """FluxKontextLoraTrainer.sampling_from_embeddings (flux_kontext_trainer.py:902-976) with the initial latents injected; scheduler =
    restated FlowMatchEulerDiscreteScheduler (dynamic exponential shift; base_trainer.py:1009-1043).  Test infrastructure."""
    import math
    steps, cfg = int(emb["num_inference_steps"]), float(emb.get("true_cfg_scale", 1.0))
    do_cfg = cfg > 1.0 and "negative_pooled_prompt_embeds" in emb
    ctrl, latents = emb["control_latents"].to(dtype), emb["latents"].to(dtype)
    ids = torch.cat([emb["latent_ids"], emb["control_ids"]], dim=0)
    B, n = latents.shape[0], latents.shape[1]
    sig = torch.linspace(1.0, 1.0 / steps, steps, dtype=torch.float64)
    m = (1.15 - 0.5) / (4096 - 256)
    mu = n * m + (0.5 - m * 256)
    sig = (math.exp(mu) / (math.exp(mu) + (1.0 / sig - 1.0))).to(torch.float32)
    ts = sig * 1000
    sig = torch.cat([sig, torch.zeros(1)])
    guidance = torch.full([B], float(emb.get("guidance", 1.0)), dtype=torch.float32) if getattr(dit, "guidance_embeds", False) else None
    with torch.no_grad():
        for i, t in enumerate(ts):
            x = torch.cat([latents, ctrl], dim=1)
            tt = t.expand(B).to(dtype) / 1000
            pred = dit(hidden_states=x, timestep=tt, guidance=guidance, pooled_projections=emb["pooled_prompt_embeds"].to(dtype),
                       encoder_hidden_states=emb["prompt_embeds"].to(dtype), txt_ids=emb["text_ids"], img_ids=ids,
                       joint_attention_kwargs={}, return_dict=False)[0][:, :n]
            if do_cfg:
                neg = dit(hidden_states=x, timestep=tt, guidance=guidance, pooled_projections=emb["negative_pooled_prompt_embeds"].to(dtype),
                          encoder_hidden_states=emb["negative_prompt_embeds"].to(dtype), txt_ids=emb["negative_text_ids"], img_ids=ids,
                          joint_attention_kwargs={}, return_dict=False)[0][:, :n]
                pred = neg + cfg * (pred - neg)
            latents = (latents.to(torch.float32) + (float(sig[i + 1]) - float(sig[i])) * pred.to(torch.float32)).to(pred.dtype)
    return latents
