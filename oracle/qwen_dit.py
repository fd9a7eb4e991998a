"""Pure-PyTorch CPU restatement of the Qwen-Image-Edit DiT LoRA training step.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Every class cites the reference lines it follows (paths relative to
/root/reference).  Sub-module / parameter names equal the reference's
state-dict keys so one state dict drives the reference
(through the build-container shim), this oracle and the HIP modules.

Numerics: the module runs in whatever dtype its parameters are in (fp32 for
tight checks, bf16 to emulate the reference's bf16 eager rounding points:
torch CPU bf16 ops accumulate in fp32 and round once per op, exactly like the
reference's eager GPU ops do).
"""
from __future__ import annotations

import math
import re
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# diffusers primitives (third-party, restated)
# ----------------------------------------------------------------------------
def timestep_sinusoid(timesteps: torch.Tensor, dim: int = 256, scale: float = 1000.0,
                      flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0.0,
                      max_period: int = 10000) -> torch.Tensor:
    """diffusers.models.embeddings.get_timestep_embedding as configured at
    src/qflux/models/transformer_qwenimage.py:147 (Timesteps(256, True, 0, scale=1000)).
    Computed in fp32 regardless of the timestep dtype; result fp32."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class OracleRMSNorm(nn.Module):
    """diffusers.models.normalization.RMSNorm (used at transformer_qwenimage.py:549 and
    inside Attention for qk_norm="rms_norm"): fp32 variance, x*rsqrt in fp32, cast to the
    weight dtype when it is half precision, THEN multiply by the weight."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.eps)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        x = x * self.weight
        return x


class _GELUProj(nn.Module):
    """diffusers.models.activations.GELU(approximate="tanh"): Linear then gelu-tanh."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=True)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class OracleFeedForward(nn.Module):
    """diffusers FeedForward(dim, dim_out=dim, activation_fn="gelu-approximate")
    (transformer_qwenimage.py:408,418): net = [GELU(proj), Dropout(0), Linear(4D, D)]."""

    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim, bias=True)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class OracleLoraLinear(nn.Module):
    """peft.tuners.lora.layer.Linear restated (third party, unpinned):
        y = base(x) + lora_B(lora_A(x.to(A.dtype))) * (lora_alpha / r);  y.to(base dtype)
    Adapter weights are fp32 (peft autocast_adapter_dtype=True) even on a bf16 base.
    Parameter names: base_layer.{weight,bias}, lora_A.<adapter>.weight [r,in],
    lora_B.<adapter>.weight [out,r]."""

    def __init__(self, base: nn.Linear, r: int, lora_alpha: float, adapter_name: str,
                 init: str = "gaussian", generator: torch.Generator | None = None):
        super().__init__()
        self.base_layer = base
        self.r = r
        self.scaling = float(lora_alpha) / float(r)
        self.adapter_name = adapter_name
        a = nn.Linear(base.in_features, r, bias=False, dtype=torch.float32)
        b = nn.Linear(r, base.out_features, bias=False, dtype=torch.float32)
        with torch.no_grad():
            if init == "gaussian":
                a.weight.copy_(torch.randn(a.weight.shape, generator=generator) / r)
            else:  # peft default: kaiming_uniform(a=sqrt(5))
                nn.init.kaiming_uniform_(a.weight, a=math.sqrt(5), generator=generator)
            b.weight.zero_()
        self.lora_A = nn.ModuleDict({adapter_name: a})
        self.lora_B = nn.ModuleDict({adapter_name: b})
        for p in self.base_layer.parameters():
            p.requires_grad_(False)

    @property
    def in_features(self):
        return self.base_layer.in_features

    @property
    def out_features(self):
        return self.base_layer.out_features

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        result = self.base_layer(x)
        out_dtype = result.dtype
        a = self.lora_A[self.adapter_name]
        b = self.lora_B[self.adapter_name]
        xa = x.to(a.weight.dtype)
        result = result + b(a(xa)) * self.scaling
        return result.to(out_dtype)


class OracleAttention(nn.Module):
    """Holder equal to diffusers Attention as constructed at transformer_qwenimage.py:394-406."""

    def __init__(self, dim: int, heads: int, dim_head: int, eps: float = 1e-6):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(dim, inner, bias=True)
        self.to_k = nn.Linear(dim, inner, bias=True)
        self.to_v = nn.Linear(dim, inner, bias=True)
        self.add_q_proj = nn.Linear(dim, inner, bias=True)
        self.add_k_proj = nn.Linear(dim, inner, bias=True)
        self.add_v_proj = nn.Linear(dim, inner, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim, bias=True), nn.Dropout(0.0)])
        self.to_add_out = nn.Linear(inner, dim, bias=True)
        self.norm_q = OracleRMSNorm(dim_head, eps)
        self.norm_k = OracleRMSNorm(dim_head, eps)
        self.norm_added_q = OracleRMSNorm(dim_head, eps)
        self.norm_added_k = OracleRMSNorm(dim_head, eps)


def apply_rope_complex(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb_qwen(use_real=False), transformer_qwenimage.py:134-140.
    x [B,S,H,d]; freqs [S,d/2] complex64. Adjacent pairs (2j,2j+1) form the complex number."""
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    out = torch.view_as_real(xc * freqs.unsqueeze(1)).flatten(3)
    return out.type_as(x)


def joint_attention(attn: OracleAttention, img: torch.Tensor, txt: torch.Tensor,
                    rope: tuple[torch.Tensor, torch.Tensor] | None,
                    key_mask: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """QwenDoubleStreamAttnProcessor2_0.__call__ (transformer_qwenimage.py:271-354).
    key_mask: optional additive [B, T+S_i] float mask (multi-resolution path,
    transformer_qwen_custom.py); None on the standard path (the text mask is ignored there)."""
    T = txt.shape[1]
    H = attn.heads
    iq, ik, iv = attn.to_q(img), attn.to_k(img), attn.to_v(img)
    tq, tk, tv = attn.add_q_proj(txt), attn.add_k_proj(txt), attn.add_v_proj(txt)
    iq, ik, iv = (t.unflatten(-1, (H, -1)) for t in (iq, ik, iv))
    tq, tk, tv = (t.unflatten(-1, (H, -1)) for t in (tq, tk, tv))
    iq, ik = attn.norm_q(iq), attn.norm_k(ik)
    tq, tk = attn.norm_added_q(tq), attn.norm_added_k(tk)
    if rope is not None and not isinstance(rope, list):
        img_f, txt_f = rope
        iq, ik = apply_rope_complex(iq, img_f), apply_rope_complex(ik, img_f)
        tq, tk = apply_rope_complex(tq, txt_f), apply_rope_complex(tk, txt_f)
    q = torch.cat([tq, iq], dim=1)  # order [text, image]  (:324-326)
    k = torch.cat([tk, ik], dim=1)
    v = torch.cat([tv, iv], dim=1)
    if isinstance(rope, list):
        # QwenDoubleStreamAttnProcessorPerSample._apply_rope_per_sample (transformer_qwen_custom.py:175-228): per sample the
        # text table covers joint rows [0, txt_len_b) and the image table the rows RIGHT AFTER them, [txt_len_b, txt_len_b +
        # seq_img_b) -- measured from the sample's own text length, not from the padded one; all other rows stay unrotated.
        qo, ko = q.clone(), k.clone()
        for b, (img_f, txt_f) in enumerate(rope):
            st, si = txt_f.shape[0], img_f.shape[0]
            qo[b:b + 1, :st] = apply_rope_complex(q[b:b + 1, :st], txt_f)
            ko[b:b + 1, :st] = apply_rope_complex(k[b:b + 1, :st], txt_f)
            qo[b:b + 1, st:st + si] = apply_rope_complex(q[b:b + 1, st:st + si], img_f)
            ko[b:b + 1, st:st + si] = apply_rope_complex(k[b:b + 1, st:st + si], img_f)
        q, k = qo, ko
    am = None
    if key_mask is not None:
        am = key_mask[:, None, None, :].to(q.dtype)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                       attn_mask=am, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).flatten(2, 3).to(q.dtype)
    t_o, i_o = o[:, :T], o[:, T:]
    i_o = attn.to_out[0](i_o)
    t_o = attn.to_add_out(t_o)
    return i_o, t_o


class OracleQwenBlock(nn.Module):
    """QwenImageTransformerBlock (transformer_qwenimage.py:377-494)."""

    def __init__(self, dim: int, heads: int, dim_head: int, eps: float = 1e-6):
        super().__init__()
        self.img_mod = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim, bias=True))
        self.img_norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.attn = OracleAttention(dim, heads, dim_head, eps)
        self.img_norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.img_mlp = OracleFeedForward(dim)
        self.txt_mod = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim, bias=True))
        self.txt_norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.txt_norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.txt_mlp = OracleFeedForward(dim)

    @staticmethod
    def _modulate(x, mod):
        shift, scale, gate = mod.chunk(3, dim=-1)  # (:422) order shift, scale, gate
        return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1), gate.unsqueeze(1)

    def forward(self, hidden_states, encoder_hidden_states, temb, rope, key_mask=None):
        img_mod1, img_mod2 = self.img_mod(temb).chunk(2, dim=-1)
        txt_mod1, txt_mod2 = self.txt_mod(temb).chunk(2, dim=-1)
        im, ig1 = self._modulate(self.img_norm1(hidden_states), img_mod1)
        tm, tg1 = self._modulate(self.txt_norm1(encoder_hidden_states), txt_mod1)
        i_o, t_o = joint_attention(self.attn, im, tm, rope, key_mask)
        hidden_states = hidden_states + ig1 * i_o
        encoder_hidden_states = encoder_hidden_states + tg1 * t_o
        im2, ig2 = self._modulate(self.img_norm2(hidden_states), img_mod2)
        hidden_states = hidden_states + ig2 * self.img_mlp(im2)
        tm2, tg2 = self._modulate(self.txt_norm2(encoder_hidden_states), txt_mod2)
        encoder_hidden_states = encoder_hidden_states + tg2 * self.txt_mlp(tm2)
        if encoder_hidden_states.dtype == torch.float16:
            encoder_hidden_states = encoder_hidden_states.clip(-65504, 65504)
        if hidden_states.dtype == torch.float16:
            hidden_states = hidden_states.clip(-65504, 65504)
        return encoder_hidden_states, hidden_states


def qwen_rope_tables(img_shapes: Sequence[Sequence[int]], txt_len: int,
                     axes_dim: Sequence[int] = (16, 56, 56), theta: float = 10000.0,
                     scale_rope: bool = True) -> tuple[torch.Tensor, torch.Tensor]:
    """QwenEmbedRope.forward/_compute_video_freqs (transformer_qwenimage.py:159-254) restated
    from positions instead of table slicing.  img_shapes = [(frame,h,w), ...] of ONE sample
    (the reference uses only the first sample's list, :206-207).
    Returns (vid_freqs [S_i, sum(axes)/2] c64, txt_freqs [T, ...] c64)."""

    def axis_freqs(pos: torch.Tensor, dim: int) -> torch.Tensor:
        inv = 1.0 / torch.pow(torch.tensor(float(theta)), torch.arange(0, dim, 2).to(torch.float32).div(dim))
        ang = torch.outer(pos.to(torch.float32), inv)
        return torch.polar(torch.ones_like(ang), ang)

    vids = []
    max_vid_index = 0
    for idx, (frame, height, width) in enumerate(img_shapes):
        fpos = torch.arange(idx, idx + frame)
        if scale_rope:
            hpos = torch.cat([torch.arange(-(height - height // 2), 0), torch.arange(0, height // 2)])
            wpos = torch.cat([torch.arange(-(width - width // 2), 0), torch.arange(0, width // 2)])
        else:
            hpos, wpos = torch.arange(height), torch.arange(width)
        ff = axis_freqs(fpos, axes_dim[0]).view(frame, 1, 1, -1).expand(frame, height, width, -1)
        fh = axis_freqs(hpos, axes_dim[1]).view(1, height, 1, -1).expand(frame, height, width, -1)
        fw = axis_freqs(wpos, axes_dim[2]).view(1, 1, width, -1).expand(frame, height, width, -1)
        vids.append(torch.cat([ff, fh, fw], dim=-1).reshape(frame * height * width, -1))
        if scale_rope:
            max_vid_index = max(height // 2, width // 2, max_vid_index)
        else:
            max_vid_index = max(height, width, max_vid_index)
    tpos = torch.arange(max_vid_index, max_vid_index + txt_len)
    txt = torch.cat([axis_freqs(tpos, d) for d in axes_dim], dim=1)
    return torch.cat(vids, dim=0).contiguous(), txt.contiguous()


class _TimestepEmbedder(nn.Module):
    def __init__(self, in_ch: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_ch, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class _TimeTextEmbed(nn.Module):
    """QwenTimestepProjEmbeddings (transformer_qwenimage.py:143-156)."""

    def __init__(self, dim: int):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedder(256, dim)

    def forward(self, timestep, hidden_states):
        proj = timestep_sinusoid(timestep, 256, scale=1000.0)
        return self.timestep_embedder(proj.to(dtype=hidden_states.dtype))


class _AdaLNContinuous(nn.Module):
    """diffusers AdaLayerNormContinuous(D, D, elementwise_affine=False, eps=1e-6):
    emb = linear(silu(c)); scale, shift = chunk(emb, 2, dim=1)."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.linear = nn.Linear(dim, 2 * dim, bias=True)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)

    def forward(self, x, c):
        emb = self.linear(F.silu(c).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class OracleQwenDiT(nn.Module):
    """QwenImageTransformer2DModel (transformer_qwenimage.py:497-672)."""

    def __init__(self, patch_size=2, in_channels=64, out_channels=16, num_layers=60,
                 attention_head_dim=128, num_attention_heads=24, joint_attention_dim=3584,
                 axes_dims_rope=(16, 56, 56)):
        super().__init__()
        self.cfg = dict(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels,
                        num_layers=num_layers, attention_head_dim=attention_head_dim,
                        num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                        axes_dims_rope=tuple(axes_dims_rope))
        D = num_attention_heads * attention_head_dim
        self.inner_dim = D
        self.axes_dims_rope = tuple(axes_dims_rope)
        self.time_text_embed = _TimeTextEmbed(D)
        self.txt_norm = OracleRMSNorm(joint_attention_dim, eps=1e-6)
        self.img_in = nn.Linear(in_channels, D)
        self.txt_in = nn.Linear(joint_attention_dim, D)
        self.transformer_blocks = nn.ModuleList(
            [OracleQwenBlock(D, num_attention_heads, attention_head_dim) for _ in range(num_layers)])
        self.norm_out = _AdaLNContinuous(D)
        self.proj_out = nn.Linear(D, patch_size * patch_size * (out_channels or in_channels), bias=True)

    def forward(self, hidden_states, encoder_hidden_states=None, encoder_hidden_states_mask=None,
                timestep=None, img_shapes=None, txt_seq_lens=None, guidance=None,
                attention_kwargs=None, return_dict=False, key_mask=None, attention_mask=None):
        batched = isinstance(img_shapes, list) and len(img_shapes) > 0 and isinstance(img_shapes[0], list)
        if attention_mask is not None or (batched and not all(sh == img_shapes[0] for sh in img_shapes)):
            return self.forward_custom(hidden_states, encoder_hidden_states, timestep, img_shapes, txt_seq_lens, attention_mask)
        hidden_states = self.img_in(hidden_states)
        timestep = timestep.to(hidden_states.dtype)  # (:623-624) sigma rounded to the model dtype
        encoder_hidden_states = self.txt_in(self.txt_norm(encoder_hidden_states))
        temb = self.time_text_embed(timestep, hidden_states)
        shapes = img_shapes[0] if isinstance(img_shapes[0], (list, tuple)) and isinstance(
            img_shapes[0][0], (list, tuple)) else img_shapes
        vid_f, txt_f = qwen_rope_tables(shapes, max(txt_seq_lens), self.axes_dims_rope)
        rope = (vid_f.to(hidden_states.device), txt_f.to(hidden_states.device))
        for block in self.transformer_blocks:
            encoder_hidden_states, hidden_states = block(hidden_states, encoder_hidden_states, temb, rope, key_mask)
        hidden_states = self.norm_out(hidden_states, temb)
        return (self.proj_out(hidden_states),)


def _forward_custom(self, hidden_states, encoder_hidden_states, timestep, img_shapes, txt_seq_lens, attention_mask):
    """QwenImageTransformer2DModel.forward of transformer_qwen_custom.py:384-573 (non-checkpointed path): per-sample RoPE when
    the samples' shape lists differ, boolean [B, T+S_max] padding mask -> additive key mask, padded text rows zeroed once
    after txt_in, padded image rows zeroed after img_in, after EVERY block and in the output."""
    hidden_states = self.img_in(hidden_states)
    timestep = timestep.to(hidden_states.dtype)
    encoder_hidden_states = self.txt_in(self.txt_norm(encoder_hidden_states))
    temb = self.time_text_embed(timestep, hidden_states)
    batched = isinstance(img_shapes, list) and len(img_shapes) > 0 and isinstance(img_shapes[0], list)
    if batched:
        B = len(img_shapes)
        lens = txt_seq_lens if isinstance(txt_seq_lens, list) else [txt_seq_lens] * B
        if all(sh == img_shapes[0] for sh in img_shapes):
            vid_f, txt_f = qwen_rope_tables(img_shapes[0], max(lens), self.axes_dims_rope)
            rope = (vid_f, txt_f)
        else:
            rope = [qwen_rope_tables(img_shapes[b], lens[b], self.axes_dims_rope) for b in range(B)]
    else:
        rope = qwen_rope_tables(img_shapes, max(txt_seq_lens) if isinstance(txt_seq_lens, list) else txt_seq_lens, self.axes_dims_rope)
    key_mask = None
    img_mask = None
    if attention_mask is not None:
        mask = attention_mask if attention_mask.dtype == torch.bool else attention_mask > 0
        T, S_i = encoder_hidden_states.shape[1], hidden_states.shape[1]
        img_mask = mask[:, T:T + S_i]
        encoder_hidden_states = encoder_hidden_states.masked_fill(~mask[:, :T].unsqueeze(-1), 0)
        hidden_states = hidden_states.masked_fill(~img_mask.unsqueeze(-1), 0)
        key_mask = torch.zeros(mask.shape, dtype=hidden_states.dtype).masked_fill(~mask, float("-inf"))
    for block in self.transformer_blocks:
        encoder_hidden_states, hidden_states = block(hidden_states, encoder_hidden_states, temb, rope, key_mask)
        if img_mask is not None and not img_mask.all():
            hidden_states = hidden_states * img_mask.unsqueeze(-1)
    out = self.proj_out(self.norm_out(hidden_states, temb))
    if img_mask is not None:
        out = out.masked_fill(~img_mask.unsqueeze(-1), 0)
    return (out,)


OracleQwenDiT.forward_custom = _forward_custom


# ----------------------------------------------------------------------------
# LoRA injection (peft LoraConfig/add_adapter semantics; base_trainer.py:929-941)
# ----------------------------------------------------------------------------
def match_target(name: str, target_modules) -> bool:
    """peft target matching: full-match regex if a string, else suffix match."""
    if isinstance(target_modules, str):
        if target_modules == "all-linear":
            return True
        return re.fullmatch(target_modules, name) is not None
    return any(name == t or name.endswith("." + t) for t in target_modules)


def add_lora(model: nn.Module, r: int = 16, lora_alpha: float = 16,
             target_modules=("to_k", "to_q", "to_v", "to_out.0"), adapter_name: str = "default",
             init: str = "gaussian", seed: int | None = None, wrapper=OracleLoraLinear) -> list[str]:
    """Wrap every nn.Linear whose dotted name matches; freeze everything but names containing
    'lora' (qwen_image_edit_trainer.py:312-318).  Returns the wrapped names in module order."""
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    names = [n for n, m in model.named_modules() if isinstance(m, nn.Linear) and match_target(n, target_modules)
             and "lora_" not in n]
    for n in names:
        parent_name, _, child = n.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        base = getattr(parent, child) if not child.isdigit() else parent[int(child)]
        wrapped = wrapper(base, r, lora_alpha, adapter_name, init, gen)
        if child.isdigit():
            parent[int(child)] = wrapped
        else:
            setattr(parent, child, wrapped)
    for pn, p in model.named_parameters():
        p.requires_grad_("lora" in pn)
    return names


# ----------------------------------------------------------------------------
# Training-step caller + criterion (qwen_image_edit_trainer.py:777-849; losses/mse_loss.py:46-83)
# ----------------------------------------------------------------------------
def flowmatch_sigmas(num_train_timesteps: int = 1000, shift: float = 1.0):
    """FlowMatchEulerDiscreteScheduler tables as built at construction (third party; table comes
    from the model repo's scheduler_config.json: dynamic shifting => unshifted at init)."""
    ts = torch.linspace(1, num_train_timesteps, num_train_timesteps).flip(0)
    sig = ts / num_train_timesteps
    sig = shift * sig / (1 + (shift - 1) * sig)
    return sig * num_train_timesteps, sig


def mse_loss(model_pred, target, weighting=None):
    """MseLoss.forward reduction='mean' (losses/mse_loss.py:66-83)."""
    if weighting is None:
        return F.mse_loss(model_pred, target, reduction="mean")
    el = (model_pred.float() - target.float()) ** 2
    wl = weighting.float() * el
    return torch.mean(wl.reshape(target.shape[0], -1), dim=1).mean()


def qwen_compute_loss(dit: nn.Module, emb: dict, noise: torch.Tensor, u: torch.Tensor,
                      dtype: torch.dtype, return_pred: bool = False):
    """QwenImageEditTrainer._compute_loss (qwen_image_edit_trainer.py:777-849) with the random
    draws (noise, u) injected.  emb keys: image_latents, control_latents, prompt_embeds,
    prompt_embeds_mask, img_shapes."""
    x0 = emb["image_latents"].to(dtype)
    ctrl = emb["control_latents"].to(dtype)
    pe = emb["prompt_embeds"].to(dtype)
    mask = emb["prompt_embeds_mask"].to(torch.int64)
    timesteps_tbl, sigmas_tbl = flowmatch_sigmas()
    with torch.no_grad():
        noise = noise.to(dtype)
        idx = (u.cpu() * 1000).long()                       # the tables live on the host like the scheduler's; results follow the data
        timesteps = timesteps_tbl[idx].to(x0.device)        # (device moves are no-ops on the CPU; the checker also runs on the GPU:
        sig = sigmas_tbl.to(dtype)[idx].flatten().to(x0.device)   # tests/test_fulldepth_gpu.py)
        while sig.ndim < x0.ndim:
            sig = sig.unsqueeze(-1)
        x_t = (1.0 - sig) * x0 + sig * noise
        packed = torch.cat([x_t, ctrl], dim=1)
        txt_seq_lens = mask.sum(dim=1).tolist()
    pred = dit(hidden_states=packed, timestep=timesteps / 1000, guidance=None,
               encoder_hidden_states_mask=mask, encoder_hidden_states=pe,
               img_shapes=emb["img_shapes"], txt_seq_lens=txt_seq_lens, return_dict=False)[0]
    pred = pred[:, : x0.size(1)]
    weighting = torch.ones_like(sig)
    target = noise - x0
    loss = mse_loss(pred, target, weighting)
    return (loss, pred) if return_pred else loss


def mask_edit_loss(model_pred, target, edit_mask=None, fg=2.0, bg=1.0):
    """MaskEditLoss.forward, reduction='mean', weighting=None (src/qflux/losses/edit_mask_loss.py:45-86)."""
    el = (model_pred.float() - target.float()) ** 2
    B, T, _ = model_pred.shape
    m = torch.ones((B, T), dtype=torch.float32, device=model_pred.device) if edit_mask is None else edit_mask.float()
    w = (m * fg + (1 - m) * bg).unsqueeze(-1)
    return torch.mean((el * w).reshape(B, -1), 1).mean()


def map_mask_to_latent(image_mask):
    """src/qflux/losses/edit_mask_loss.py:7-36: 8x8 average pool, 2x2 patch maximum, flatten."""
    B, H, W = image_mask.shape
    lh, lw = H // 8, W // 8
    m = torch.nn.functional.avg_pool2d(image_mask.float().unsqueeze(1), kernel_size=8, stride=8).squeeze(1)
    p = m.reshape(B, lh // 2, 2, lw // 2, 2).permute(0, 1, 3, 2, 4).contiguous().view(B, lh // 2, lw // 2, 4)
    return p.max(dim=-1)[0].view(B, (lh // 2) * (lw // 2))


def qwen_sample(dit: nn.Module, emb: dict, dtype: torch.dtype):
    """QwenImageEditTrainer.sampling_from_embeddings (qwen_image_edit_trainer.py:1116-1289) with the initial latents injected;
    scheduler = restated FlowMatchEulerDiscreteScheduler (dynamic exponential shift; base_trainer.py:1009-1043)."""
    import math
    steps, cfg = int(emb["num_inference_steps"]), float(emb.get("true_cfg_scale", 1.0))
    do_cfg = cfg > 1 and emb.get("negative_prompt_embeds") is not None
    ctrl, pe, mask = emb["control_latents"].to(dtype), emb["prompt_embeds"].to(dtype), emb["prompt_embeds_mask"]
    latents = emb["latents"].to(dtype)
    n = latents.shape[1]
    sig = torch.linspace(1.0, 1.0 / steps, steps, dtype=torch.float64)
    m = (1.15 - 0.5) / (4096 - 256)
    mu = n * m + (0.5 - m * 256)
    sig = (math.exp(mu) / (math.exp(mu) + (1.0 / sig - 1.0))).to(torch.float32)
    ts = sig * 1000
    sig = torch.cat([sig, torch.zeros(1)])
    with torch.no_grad():
        for i, t in enumerate(ts):
            x = torch.cat([latents, ctrl], dim=1)
            tt = t.expand(latents.shape[0]).to(dtype) / 1000
            pred = dit(hidden_states=x, timestep=tt, guidance=None, encoder_hidden_states_mask=mask, encoder_hidden_states=pe,
                       img_shapes=emb["img_shapes"], txt_seq_lens=mask.sum(dim=1).tolist(), return_dict=False)[0][:, :n]
            if do_cfg:
                nm = emb["negative_prompt_embeds_mask"]
                neg = dit(hidden_states=x, timestep=tt, guidance=None, encoder_hidden_states_mask=nm,
                          encoder_hidden_states=emb["negative_prompt_embeds"].to(dtype), img_shapes=emb["img_shapes"],
                          txt_seq_lens=nm.sum(dim=1).tolist(), return_dict=False)[0][:, :n]
                comb = neg + cfg * (pred - neg)
                pred = comb * (torch.norm(pred, dim=-1, keepdim=True) / torch.norm(comb, dim=-1, keepdim=True))
            latents = (latents.to(torch.float32) + (float(sig[i + 1]) - float(sig[i])) * pred.to(torch.float32)).to(pred.dtype)
    return latents
