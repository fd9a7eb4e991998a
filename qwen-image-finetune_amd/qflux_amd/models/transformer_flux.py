"""MI355X-native drop-in for FluxTransformer2DModel on the LoRA-training hot path
(reference: src/qflux/models/transformer_flux.py:557-828; blocks :385-523; attention processor :102-166;
FluxPosEmbed :526-554).  Same constructor arguments, diffusers state-dict keys and forward signature.

Built on the same launch-program machinery as the Qwen model: the double-stream blocks reuse the Qwen emitters
(AdaLayerNormZero == modulation GEMV + ln_modulate with the same (shift, scale, gate) x2 chunk order); the
single-stream blocks run on the joint [text|image] buffer:
    ln_modulate -> grouped q/k/v GEMM (+LoRA K-ext) + proj_mlp GEMM (GELU epilogue) -> qk_norm_rope -> attention ->
    proj_out as a two-segment GEMM  [attn | gelu(mlp)] @ W_out^T  (no concat buffer), epilogue x + gate*y
and backward mirrors it with ONE dX GEMM over K = 3D (dq|dk|dv) + 4D (d mlp) + LoRA extension.
RoPE: real cos/sin from ids (float64 on the host, cached) == the complex rotation the kernels already apply;
q/k norms are torch.nn.RMSNorm (single rounding): norm_flags = 1.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..plan_cache import PlanCache, ladder
from ..modules import LoraStore, QfxLinear, QfxLoraLinear, QfxRMSNorm
from .transformer_qwenimage import (BF, F32, QfxAttention, QfxFeedForward, QwenImageTransformer2DModel, _AdaLNOut, _Cfg, _LinW,
                                    _Prog, _QwenPlan, _TimestepEmbedder, _ceil, _ptr)

lib = L.lib


class _NormLinear(nn.Module):
    """AdaLayerNormZero / -Single holder: only `.linear` carries parameters."""

    def __init__(self, dim, n):
        super().__init__()
        self.linear = QfxLinear(dim, n * dim)


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head):
        super().__init__()
        self.norm1 = _NormLinear(dim, 6)
        self.norm1_context = _NormLinear(dim, 6)
        self.attn = QfxAttention(dim, heads, dim_head, eps=1e-6)
        self.ff = QfxFeedForward(dim)
        self.ff_context = QfxFeedForward(dim)


class _SingleAttn(nn.Module):
    def __init__(self, dim, heads, dim_head):
        super().__init__()
        self.heads = heads
        for n in ("to_q", "to_k", "to_v"):
            setattr(self, n, QfxLinear(dim, heads * dim_head))
        self.norm_q = QfxRMSNorm(dim_head, 1e-6)
        self.norm_k = QfxRMSNorm(dim_head, 1e-6)


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, mlp_ratio=4.0):
        super().__init__()
        self.norm = _NormLinear(dim, 3)
        self.proj_mlp = QfxLinear(dim, int(dim * mlp_ratio))
        self.proj_out = QfxLinear(dim + int(dim * mlp_ratio), dim)
        self.attn = _SingleAttn(dim, heads, dim_head)


class _TextProj(nn.Module):
    def __init__(self, pooled, dim):
        super().__init__()
        self.linear_1 = QfxLinear(pooled, dim)
        self.linear_2 = QfxLinear(dim, dim)


class _CombinedEmb(nn.Module):
    def __init__(self, dim, pooled, guidance):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedder(dim)
        if guidance:
            self.guidance_embedder = _TimestepEmbedder(dim)
        self.text_embedder = _TextProj(pooled, dim)


def flux_joint_rope(ids: torch.Tensor, axes_dim, theta: float = 10000.0) -> torch.Tensor:
    """FluxPosEmbed (transformer_flux.py:533-554) in the kernels' layout [S, dh/2, 2] (cos, sin), float64 math."""
    pos = ids.detach().float().cpu()
    parts = []
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = torch.outer(pos[:, i].to(torch.float64), freqs)
        parts.append(torch.stack([ang.cos(), ang.sin()], dim=-1))
    return torch.cat(parts, dim=1).float().contiguous()


class FluxTransformer2DModel(QwenImageTransformer2DModel):
    """See module docstring."""

    def __init__(self, patch_size: int = 1, in_channels: int = 64, out_channels: int | None = None, num_layers: int = 19,
                 num_single_layers: int = 38, attention_head_dim: int = 128, num_attention_heads: int = 24,
                 joint_attention_dim: int = 4096, pooled_projection_dim: int = 768, guidance_embeds: bool = False,
                 axes_dims_rope=(16, 56, 56)):
        nn.Module.__init__(self)
        if attention_head_dim not in (64, 128):
            raise ValueError("qflux_amd attention kernels support head dims 64 and 128")
        self.config = _Cfg(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                           num_single_layers=num_single_layers, attention_head_dim=attention_head_dim,
                           num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                           pooled_projection_dim=pooled_projection_dim, guidance_embeds=guidance_embeds,
                           axes_dims_rope=tuple(axes_dims_rope))
        self.out_channels = out_channels or in_channels
        self.inner_dim = D = num_attention_heads * attention_head_dim
        self.time_text_embed = _CombinedEmb(D, pooled_projection_dim, guidance_embeds)
        self.context_embedder = QfxLinear(joint_attention_dim, D)
        self.x_embedder = QfxLinear(in_channels, D)
        self.transformer_blocks = nn.ModuleList([FluxTransformerBlock(D, num_attention_heads, attention_head_dim) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(D, num_attention_heads, attention_head_dim) for _ in range(num_single_layers)])
        self.norm_out = _AdaLNOut(D)
        self.proj_out = QfxLinear(D, patch_size * patch_size * self.out_channels)
        self.gradient_checkpointing = False
        self._lora = LoraStore(self)
        self._adapter_name = None
        self._prepared = None
        self._lora_prep = None
        self._plans = PlanCache()
        self._version = 0
        self._adapter_gen = 0      # bumped ONLY by add_adapter / load_lora_adapter / load_state_dict: what a data-parallel resync keys on
        self._rope_cache = {}

    _HEAD_SITES = {"x_embedder": "x_in", "context_embedder": "c_in", "proj_out": "proj_out"}

    _COND_SUFFIXES = ("timestep_embedder.linear_1", "timestep_embedder.linear_2", "guidance_embedder.linear_1", "guidance_embedder.linear_2",
                      "text_embedder.linear_1", "text_embedder.linear_2", "norm1.linear", "norm1_context.linear", "norm.linear",
                      "norm_out.linear")

    def _cond_modules(self):
        te = self.time_text_embed
        mods = [te.timestep_embedder.linear_1, te.timestep_embedder.linear_2, te.text_embedder.linear_1, te.text_embedder.linear_2,
                self.norm_out.linear]
        if self.config.guidance_embeds:
            mods += [te.guidance_embedder.linear_1, te.guidance_embedder.linear_2]
        for blk in self.transformer_blocks:
            mods += [blk.norm1.linear, blk.norm1_context.linear]
        mods += [blk.norm.linear for blk in self.single_transformer_blocks]
        return mods

    def _lora_supported(self, name: str) -> bool:
        if name in self._HEAD_SITES or name.endswith(self._COND_SUFFIXES):
            return True
        if name.startswith("transformer_blocks."):
            return name.endswith(self._LORA_SUFFIXES)
        return name.startswith("single_transformer_blocks.") and name.endswith(("attn.to_q", "attn.to_k", "attn.to_v", "proj_mlp", "proj_out"))

    # ------------------------------------------------------------------ preparation
    def _prepare(self):
        if self._prepared is not None:
            return self._prepared
        assert self.device.type == "cuda", "qflux_amd runs on the GPU only (no CPU fallback)"
        dev = self.device
        P = {"blocks": [], "singles": []}
        for blk in self.transformer_blocks:
            a = blk.attn
            w = {}
            for s, names in (("img", ("to_q", "to_k", "to_v")), ("txt", ("add_q_proj", "add_k_proj", "add_v_proj"))):
                qkv = [_LinW(getattr(a, n), False) for n in names]
                w[s + ".qkv"] = qkv
                w[s + ".qkvT"] = torch.cat([l.W for l in qkv], dim=0).t().contiguous()
            w["img.o"] = _LinW(a.to_out[0], True)
            w["txt.o"] = _LinW(a.to_add_out, True)
            for s, mlp in (("img", blk.ff), ("txt", blk.ff_context)):
                w[s + ".fc1"] = _LinW(mlp.net[0].proj, True)
                w[s + ".fc2"] = _LinW(mlp.net[2], True)
            w["img.mod"] = _LinW(blk.norm1.linear, False)
            w["txt.mod"] = _LinW(blk.norm1_context.linear, False)
            w["norms"] = (a.norm_added_q.weight.data, a.norm_added_k.weight.data, a.norm_q.weight.data, a.norm_k.weight.data)
            P["blocks"].append(w)
        for blk in self.single_transformer_blocks:
            a = blk.attn
            w = {"qkv": [_LinW(getattr(a, n), False) for n in ("to_q", "to_k", "to_v")]}
            w["qkvT"] = torch.cat([l.W for l in w["qkv"]], dim=0).t().contiguous()       # [D, 3D]
            w["mlp"] = _LinW(blk.proj_mlp, True)                                           # WT [D, 4D]
            w["out"] = _LinW(blk.proj_out, True)                                           # W [D, 5D], WT [5D, D]
            w["mod"] = _LinW(blk.norm.linear, False)
            w["norms"] = (a.norm_q.weight.data, a.norm_k.weight.data)
            w["B2"] = w["mlp"].WT                                                          # replaced by [WmlpT | WeT] when LoRA'd
            P["singles"].append(w)
        te = self.time_text_embed
        lin = {"x_in": self.x_embedder, "c_in": self.context_embedder, "t1": te.timestep_embedder.linear_1,
               "t2": te.timestep_embedder.linear_2, "p1": te.text_embedder.linear_1, "p2": te.text_embedder.linear_2,
               "norm_out": self.norm_out.linear}
        if self.config.guidance_embeds:
            lin["g1"], lin["g2"] = te.guidance_embedder.linear_1, te.guidance_embedder.linear_2
        for k, m in lin.items():
            P[k] = _LinW(m, False)
            P[k + "_Wp"] = torch.tensor([P[k].W.data_ptr()], dtype=torch.int64, device=dev)
            P[k + "_bp"] = torch.tensor([P[k].b.data_ptr()], dtype=torch.int64, device=dev)
        P["proj_out"] = _LinW(self.proj_out, True)
        Jd = self.config.joint_attention_dim
        if Jd % 64:   # the GEMM contracts in 64-wide K tiles: zero-pad the context embedder's K once
            Jp = _ceil(Jd, 64)
            wpad = torch.zeros(self.inner_dim, Jp, dtype=BF, device=dev)
            wpad[:, :Jd].copy_(P["c_in"].W)
            P["c_in"].W, P["c_in"].K = wpad, Jp
        mods = [w[s + ".mod"] for w in P["blocks"] for s in ("img", "txt")]
        P["mod_W"] = torch.tensor([m.W.data_ptr() for m in mods], dtype=torch.int64, device=dev)
        P["mod_b"] = torch.tensor([m.b.data_ptr() for m in mods], dtype=torch.int64, device=dev)
        P["smod_W"] = torch.tensor([w["mod"].W.data_ptr() for w in P["singles"]] or [0], dtype=torch.int64, device=dev)
        P["smod_b"] = torch.tensor([w["mod"].b.data_ptr() for w in P["singles"]] or [0], dtype=torch.int64, device=dev)
        self._prepared = P
        return P

    def _prepare_lora(self):
        if self._lora_prep is not None:
            return self._lora_prep
        P = self._prepare()
        self._ensure_lora_store()
        dev, D = self.device, self.inner_dim
        descs = []
        max_dim = 1
        for w, blk in zip(P["blocks"], self.transformer_blocks):
            max_dim = max(max_dim, self._prep_double_lora(w, blk.attn, descs))
        for w, blk in zip(P["singles"], self.single_transformer_blocks):
            a = blk.attn

            # adapters on proj_mlp / proj_out (configs/face_seg_flux_kontext_fp16.yaml:11) are stand-alone sites; proj_mlp's
            # backward K-extension joins segment 2 of the block's dX GEMM: [W_mlp^T | WeT(q) WeT(k) WeT(v) | WeT(mlp)]
            max_dim = max(max_dim, self._prep_site_lora(w["mlp"], descs), self._prep_site_lora(w["out"], descs))
            kx_mlp = w["mlp"].lora.Kext if w["mlp"].lora is not None else 0

            def make_wet(Kext, w=w, kx_mlp=kx_mlp):
                # dX B operand of the single block: [W_mlp^T | WeT(q) WeT(k) WeT(v)] so the LoRA K-extension rides in segment 2
                b2 = torch.zeros(D, 4 * D + 3 * Kext + kx_mlp, dtype=BF, device=dev)
                b2[:, : 4 * D].copy_(w["mlp"].WT)
                w["B2"] = b2
                return b2[:, 4 * D: 4 * D + 3 * Kext]

            max_dim = max(max_dim, self._prep_qkv_lora(w, "", [a.to_q, a.to_k, a.to_v], descs, WeT=make_wet))
            if kx_mlp:
                if w["qkv_lora"] is None:
                    make_wet(0)
                # the packed WeT of proj_mlp must live inside B2: re-point the adapter's WeT (and its pack descriptor) there
                lo = w["mlp"].lora
                kq = 3 * w["qkv_lora"]["Kext"] if w["qkv_lora"] is not None else 0
                lo.WeT = w["B2"][:, 4 * D + kq: 4 * D + kq + kx_mlp]
                for d_ in descs:
                    if d_.A == lo.mod.A.data_ptr():
                        d_.WeT, d_.ld_wet = lo.WeT.data_ptr(), lo.WeT.stride(0)
        max_dim = max(max_dim, self._prep_head_lora(P, descs))
        prep = dict(n=len(descs), max_dim=max_dim, descs=None)
        if descs:
            arr = (L.LoraPackArgs * len(descs))(*descs)
            prep["descs"] = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._lora_prep = prep
        return prep

    # ------------------------------------------------------------------ forward
    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, img_ids=None, txt_ids=None,
                guidance=None, joint_attention_kwargs=None, controlnet_block_samples=None, controlnet_single_block_samples=None,
                return_dict=True, controlnet_blocks_repeat=False, attention_mask=None):
        if controlnet_block_samples is not None or controlnet_single_block_samples is not None:
            raise NotImplementedError("controlnet residuals are not part of the LoRA training path")
        if self.config.guidance_embeds and guidance is None:
            raise ValueError("guidance_embeds=True requires `guidance`")
        B, S_i, T = hidden_states.shape[0], hidden_states.shape[1], encoder_hidden_states.shape[1]
        if attention_mask is not None:
            m = attention_mask if attention_mask.dtype == torch.bool else attention_mask > 0
            if m.dim() != 2 or m.shape[1] < T + S_i:
                raise ValueError("attention_mask must have shape (batch, total_sequence_length).")
            if not bool(m.all()):
                plan = self.get_plan_multires(B, S_i, T, img_ids, m[:, T:T + S_i].sum(dim=1).tolist())
            else:
                attention_mask = None
        if attention_mask is None:
            plan = self.get_plan(B, S_i, T, img_ids[0] if img_ids.ndim == 3 else img_ids, txt_ids)
        from .transformer_qwenimage import _QwenDiTFn
        out = _QwenDiTFn.apply(self, plan, (hidden_states, pooled_projections, guidance), encoder_hidden_states, timestep,
                               *self.lora_parameters())
        return (out,) if not return_dict else _Cfg(sample=out)

    def get_plan(self, B, S_i, T, img_ids, txt_ids):
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.shape[0] != S_i or txt_ids.shape[0] != T:
            raise ValueError("img_ids / txt_ids do not match the sequence lengths")
        ids = torch.cat((txt_ids.float().cpu(), img_ids.float().cpu()), dim=0)
        rkey = (tuple(ids.shape), hash(ids.numpy().tobytes()))
        self._prepare()
        self._prepare_lora()
        key = (B, S_i, T, rkey, self._version)
        return self._plans.get_or_build(key, lambda: _FluxPlan(self, B, S_i, T, ids))


def _get_plan_multires(self, B, S_i, T, img_ids, valid_lens):
    """Multi-resolution plan: one per LADDER size of the padded shape, in an LRU-bounded cache (plan_cache.py); rows between the
    batch maximum and the ladder size are further padded rows.  RoPE / masks are refreshed every call."""
    if img_ids.ndim == 2:
        img_ids = img_ids.unsqueeze(0).expand(B, -1, -1)
    self._prepare()
    self._prepare_lora()
    S_plan = ladder(S_i)
    key = (B, S_plan, T, "multires", self._version)
    plan = self._plans.get_or_build(key, lambda: _FluxPlan(self, B, S_plan, T, None, multires=True))
    plan.set_multires(img_ids, valid_lens)
    return plan


FluxTransformer2DModel.get_plan_multires = _get_plan_multires


class _FluxPlan(_QwenPlan):
    NORM_FLAGS = 1  # torch.nn.RMSNorm rounding

    def __init__(self, model: FluxTransformer2DModel, B, S_i, T, ids, multires=False):
        self._setup(model, B, S_i, T)
        self.multires = multires
        cfg = model.config
        D, S, H, dh, S_pad = self.D, self.S, self.H, self.dh, self.S_pad
        buf, rows = self.buf, self.rows
        Ld, Ls = cfg.num_layers, cfg.num_single_layers
        Cin, Cout, Jd, Pd = cfg.in_channels, model.proj_out.out_features, cfg.joint_attention_dim, cfg.pooled_projection_dim
        P = model._prepared
        A = self.A
        if multires:
            # dynamic per-step buffers: per-sample RoPE, additive key mask, padded-row masks (set_multires fills them)
            A["rope_b"] = buf(B, S, dh // 2, 2, dtype=F32)
            A["kmask"] = buf(B, S, dtype=F32, zero=True)
            A["rm_img"] = buf(B * S_i, dtype=F32)
            A["rm_joint"] = buf(B * S, dtype=F32)
            self.rope, self.rope_bs = A["rope_b"], S * (dh // 2) * 2
            self.kmask = A["kmask"]
            self.rmask = {"img": A["rm_img"], "txt": None, "joint": A["rm_joint"]}
        else:
            self.rope = flux_joint_rope(ids, cfg.axes_dims_rope).to(model.device)
            assert self.rope.shape == (S, dh // 2, 2)
        A["in_img"] = buf(B * S_i, Cin); A["in_txt"] = buf(B * T, P["c_in"].K, zero=True); A["pooled"] = buf(B, Pd)
        A["t"] = buf(B, dtype=F32); A["gd"] = buf(B, dtype=F32)
        for k in ("tproj", "gproj"):
            A[k] = buf(B, 256)
        for k in ("t1", "t2", "g1", "g2", "p1", "p2", "temb"):
            A[k] = buf(1, B, D)
        A["X"] = {s: [buf(rows[s], D) for _ in range(Ld + 1)] for s in ("img", "txt")}
        A["J"] = [buf(B * S, D) for _ in range(Ls + 1)]
        A["mods"] = buf(max(2 * Ld, 1), B, 6 * D); A["smods"] = buf(max(Ls, 1), B, 3 * D); A["mod_out"] = buf(1, B, 2 * D)
        A["xn_out"] = buf(B * S_i, D); A["out"] = buf(B * S_i, Cout)
        A["dpred"] = buf(B * S_i, Cout); A["dxn"] = buf(B * S_i, D)
        A["blk"] = [self._alloc_double_block(w) for w in P["blocks"]]
        self._alloc_double_scratch(P["blocks"])
        # single-stream blocks
        mpj = _ceil(B * S, 128)
        A["sblk"] = []
        kext_s, rp_s = 0, 0
        kext_m = 0
        for w in P["singles"]:
            b = dict(qkv=buf(B, S, 3 * D), sqk=buf(B, S, 2 * D), lse=buf(B, H, S_pad, dtype=F32, zero=True), h=buf(B * S, 4 * D))
            if w["out"].lora is not None or getattr(self.model, "_quant", None):
                # adapter on proj_out: its input [attn | gelu(mlp)] is kept as ONE row-major buffer (attention and the GELU epilogue
                # write straight into its two column ranges), so that u = cat A^T and dA = v^T cat are single rank-r launches.
                # MX-FP8 trunk: the same layout makes proj_out ONE contraction over K = 5D, which the block-scaled kernel takes
                # (its two-segment form -- different operands per segment -- exists in bf16 only)
                b["cat"] = buf(B * S, 5 * D)
                b["ao"] = b["cat"][:, :D]
            else:
                b["ao"] = buf(B * S, D)
            grp = w["qkv_lora"]
            if grp is not None or w["mlp"].lora is not None:
                b["xm"] = buf(B * S, D)
            if grp is not None:
                b["Uqkv"] = (buf(3 * grp["Rp"], mpj, zero=True), buf(3 * grp["Rp"], mpj, zero=True))
                kext_s, rp_s = max(kext_s, grp["Kext"]), max(rp_s, grp["Rp"])
            b["site_mlp"] = self._site_alloc(w["mlp"], B * S)
            b["site_out"] = self._site_alloc(w["out"], B * S)
            if w["mlp"].lora is not None:
                kext_m = max(kext_m, w["mlp"].lora.Kext)
            A["sblk"].append(b)
        A["xm_j"] = buf(B * S, D); A["g_j"] = buf(B * S, 4 * D)
        A["A2"] = buf(B * S, 4 * D + 3 * kext_s + kext_m, zero=True)   # [d(mlp pre-act) | LoRA v ext (q,k,v) | LoRA v ext (proj_mlp)] : A operand, segment 2 of the dX GEMM
        self.cond = model.cond_lora
        if self.cond:
            from ..cond_hip import CondHeadHip
            A["dmods"] = buf(max(2 * Ld, 1), B, 6 * D, dtype=F32, zero=True)
            A["dsmods"] = buf(max(Ls, 1), B, 3 * D, dtype=F32, zero=True)
            A["dmod_out"] = buf(1, B, 2 * D, dtype=F32, zero=True)
            te = model.time_text_embed
            chains = [(te.timestep_embedder.linear_1, te.timestep_embedder.linear_2, A["tproj"], A["t1"], A["t2"])]
            if cfg.guidance_embeds:
                chains.append((te.guidance_embedder.linear_1, te.guidance_embedder.linear_2, A["gproj"], A["g1"], A["g2"]))
            chains.append((te.text_embedder.linear_1, te.text_embedder.linear_2, A["pooled"], A["p1"], A["p2"]))
            banks = []
            if Ld:
                banks.append(([m for blk in model.transformer_blocks for m in (blk.norm1.linear, blk.norm1_context.linear)],
                              A["mods"], A["dmods"]))
            if Ls:
                banks.append(([blk.norm.linear for blk in model.single_transformer_blocks], A["smods"], A["dsmods"]))
            banks.append(([model.norm_out.linear], A["mod_out"], A["dmod_out"]))
            self.cond_head = CondHeadHip(model, B, D, chains=chains, temb=A["temb"], banks=banks, buf=buf)
            for bb in A["blk"]:
                bb["y1"] = {s: buf(rows[s], D) for s in ("img", "txt")}
                bb["y2"] = {s: buf(rows[s], D) for s in ("img", "txt")}
            for bb in A["sblk"]:
                bb["y"] = buf(B * S, D)
        A["site"] = {"x_in": self._site_alloc(P["x_in"], rows["img"]), "c_in": self._site_alloc(P["c_in"], rows["txt"]),
                     "proj_out": self._site_alloc(P["proj_out"], rows["img"])}
        self.in_grad = P["x_in"].lora is not None or P["c_in"].lora is not None
        self.full_bwd = self.in_grad or self.cond
        if self.full_bwd and Ld == 0:
            raise NotImplementedError("LoRA on the embedders / conditioning head of a FLUX model without double-stream blocks")
        if kext_s:
            A["ext3_j"] = buf(B * S, 3 * kext_s, zero=True)
            A["Vt_j"] = (buf(3 * rp_s, mpj, zero=True), buf(3 * rp_s, mpj, zero=True))
        A["dJ"] = [buf(B * S, D, zero=True), buf(B * S, D, zero=True)]
        A["dyg_j"] = buf(B * S, D, zero=True)
        A["dxm_j"] = buf(B * S, D)
        self.kext_s = kext_s
        self.fwd = _Prog()
        self.bwd = _Prog()
        self._build_forward(P)
        self._build_backward(P)

    # ------------------------------------------------------------------ forward
    def _gemv(self, p, P, key, x, K, N, silu, out):
        p.c(lib.qfx_mod_gemv, _ptr(x), self.B, K, _ptr(P[key + "_Wp"]), _ptr(P[key + "_bp"]), 1, N, silu, _ptr(out))

    def _build_forward(self, P):
        A, B, D, S, H, dh, T, S_i = self.A, self.B, self.D, self.S, self.H, self.dh, self.T, self.S_i
        S_pad = self.S_pad
        p = self.fwd
        model = self.model
        cfg = model.config
        Ld, Ls = cfg.num_layers, cfg.num_single_layers
        rows, rpb, off = self.rows, self.rpb, self.off
        eps = 1e-6
        # ---- temb = time_emb(+ guidance_emb) + pooled text emb   (CombinedTimestep(Guidance)TextProjEmbeddings)
        p.c(lib.qfx_timestep_embed, _ptr(A["t"]), B, 256, 1.0, 1000.0, _ptr(A["tproj"]))
        if self.cond:   # adapters on the conditioning head: base GEMVs + the banks' rank-r launches (cond_hip.py)
            if cfg.guidance_embeds:
                p.c(lib.qfx_timestep_embed, _ptr(A["gd"]), B, 256, 1.0, 1000.0, _ptr(A["gproj"]))
            self.cond_head.emit_forward(p)
        else:
            self._cond_hip(p, P)
        self._build_forward_body(P)

    def _cond_hip(self, p, P):
        A, B, D = self.A, self.B, self.D
        cfg = self.model.config
        Ld, Ls = cfg.num_layers, cfg.num_single_layers
        self._gemv(p, P, "t1", A["tproj"], 256, D, 0, A["t1"])
        self._gemv(p, P, "t2", A["t1"], D, D, 1, A["t2"])
        self._gemv(p, P, "p1", A["pooled"], cfg.pooled_projection_dim, D, 0, A["p1"])
        self._gemv(p, P, "p2", A["p1"], D, D, 1, A["p2"])
        n = B * D
        if cfg.guidance_embeds:
            p.c(lib.qfx_timestep_embed, _ptr(A["gd"]), B, 256, 1.0, 1000.0, _ptr(A["gproj"]))
            self._gemv(p, P, "g1", A["gproj"], 256, D, 0, A["g1"])
            self._gemv(p, P, "g2", A["g1"], D, D, 1, A["g2"])
            p.c(lib.qfx_add3_bf16, _ptr(A["t2"]), _ptr(A["g2"]), _ptr(A["p2"]), _ptr(A["temb"]), n)
        else:
            p.c(lib.qfx_add3_bf16, _ptr(A["t2"]), _ptr(A["p2"]), None, _ptr(A["temb"]), n)
        if Ld:
            p.c(lib.qfx_mod_gemv, _ptr(A["temb"]), B, D, _ptr(P["mod_W"]), _ptr(P["mod_b"]), 2 * Ld, 6 * D, 1, _ptr(A["mods"]))
        if Ls:
            p.c(lib.qfx_mod_gemv, _ptr(A["temb"]), B, D, _ptr(P["smod_W"]), _ptr(P["smod_b"]), Ls, 3 * D, 1, _ptr(A["smods"]))
        self._gemv(p, P, "norm_out", A["temb"], D, 2 * D, 1, A["mod_out"])

    def _build_forward_body(self, P):
        A, B, D, S, H, dh, T, S_i = self.A, self.B, self.D, self.S, self.H, self.dh, self.T, self.S_i
        S_pad = self.S_pad
        p = self.fwd
        model = self.model
        cfg = model.config
        Ld, Ls = cfg.num_layers, cfg.num_single_layers
        rows, rpb, off = self.rows, self.rpb, self.off
        eps = 1e-6
        # ---- embedders; with no double blocks the embeddings go straight into the joint buffer
        if Ld == 0 and Ls == 0:
            raise NotImplementedError("FLUX model without any transformer block")
        first_out = {s: ((A["X"][s][0], (0, 0)) if Ld else (A["J"][0], (S, off[s]))) for s in ("img", "txt")}
        kw = self._site_fwd(p, P["x_in"], A["site"]["x_in"], A["in_img"], cfg.in_channels, rows["img"])
        self._gemm(p, A1=A["in_img"], lda1=cfg.in_channels, B1=P["x_in"].W, K1=cfg.in_channels, M=rows["img"], N=D,
                   C_=first_out["img"][0], ldc=D, bias=P["x_in"].b, rpb=rpb["img"], c_map=first_out["img"][1], row_mask=self.rmask["img"], **kw)
        kw = self._site_fwd(p, P["c_in"], A["site"]["c_in"], A["in_txt"], P["c_in"].K, rows["txt"])
        self._gemm(p, A1=A["in_txt"], lda1=P["c_in"].K, B1=P["c_in"].W, K1=P["c_in"].K, M=rows["txt"], N=D,
                   C_=first_out["txt"][0], ldc=D, bias=P["c_in"].b, rpb=rpb["txt"], c_map=first_out["txt"][1], **kw)
        self.attn_args = []
        for i in range(Ld):
            mods = {"img": A["mods"][2 * i], "txt": A["mods"][2 * i + 1]}
            to_joint = (i + 1 == Ld) and Ls > 0
            x_out = {s: ((A["J"][0], (S, off[s])) if to_joint else (A["X"][s][i + 1], (0, 0))) for s in ("img", "txt")}
            self._emit_double_fwd(p, P["blocks"][i], A["blk"][i], mods, {s: A["X"][s][i] for s in ("img", "txt")}, x_out,
                                  last=(i + 1 == Ld and Ls == 0), norm_flags=self.NORM_FLAGS)
        self.sattn_args = []
        for i in range(Ls):
            self._emit_single_fwd(p, P["singles"][i], A["sblk"][i], A["smods"][i], A["J"][i], A["J"][i + 1])
        # ---- norm_out + proj_out on the image rows of the joint buffer (per sample: contiguous row ranges)
        mo = A["mod_out"][0]
        if Ls:
            JL = A["J"][Ls]
            for b in range(B):
                p.c(lib.qfx_ln_modulate_fwd, _ptr(JL[b * S + T:]), _ptr(mo[b:b + 1, D:2 * D]), _ptr(mo[b:b + 1, 0:D]), 2 * D,
                    _ptr(A["xn_out"][b * S_i:]), S_i, D, S_i, eps)
        else:
            p.c(lib.qfx_ln_modulate_fwd, _ptr(A["X"]["img"][Ld]), _ptr(mo[:, D:2 * D]), _ptr(mo[:, 0:D]), 2 * D, _ptr(A["xn_out"]),
                rows["img"], D, rpb["img"], eps)
        po = P["proj_out"]
        kw = self._site_fwd(p, po, A["site"]["proj_out"], A["xn_out"], D, rows["img"])
        self._gemm(p, A1=A["xn_out"], lda1=D, B1=po.W, K1=D, M=rows["img"], N=po.N, C_=A["out"], ldc=po.N, bias=po.b,
                   row_mask=self.rmask["img"], **kw)

    def _emit_single_fwd(self, p, w, bb, mod, x, x_next):
        """FluxSingleTransformerBlock.forward (transformer_flux.py:407-436) on the joint buffer; mod [B,3D] = shift|scale|gate."""
        A, B, D, S, H, dh, T = self.A, self.B, self.D, self.S, self.H, self.dh, self.T
        S_pad = self.S_pad
        eps = 1e-6
        M = B * S
        grp = w["qkv_lora"]
        xm = bb.get("xm", A["xm_j"])
        q2 = bb["qkv"].view(M, 3 * D)
        sqk2 = bb["sqk"].view(M, 2 * D)
        fused = grp is not None and self._ln_down(p, [(self._ln_fwd_args(x, mod[:, 0:D], mod[:, D:2 * D], 3 * D, xm, M, D, S, eps),
                                                       dict(W_hi=grp["A_hi"], W_lo=grp["A_lo"], W_fr=grp.get("A_fr"), ldw=D, R=3 * grp["Rp"], Ut=bb["Uqkv"],
                                                            ext=A["ext3_j"], ld_ext=A["ext3_j"].stride(0), group_R=grp["Rp"],
                                                            group_stride=grp["Kext"]))])
        if not fused:
            p.c(lib.qfx_ln_modulate_fwd, _ptr(x), _ptr(mod[:, 0:D]), _ptr(mod[:, D:2 * D]), 3 * D, _ptr(xm), M, D, S, eps)
            if grp is not None:
                self._down(p, X=xm, ldx=D, M=M, K=D, W_hi=grp["A_hi"], W_lo=grp["A_lo"], ldw=D, R=3 * grp["Rp"], Ut=bb["Uqkv"],
                           ext=A["ext3_j"], ld_ext=A["ext3_j"].stride(0), group_R=grp["Rp"], group_stride=grp["Kext"])
        groups = []
        for sec in range(3):
            lw = w["qkv"][sec]
            kw = {}
            if lw.lora is not None:
                kw = dict(A2=A["ext3_j"][:, sec * grp["Kext"]:], lda2=A["ext3_j"].stride(0), B2=lw.lora.We, ldb2=lw.lora.We.stride(0),
                          K2=lw.lora.Kext)
            c_, ldc = (sqk2[:, sec * D:], 2 * D) if sec < 2 else (q2[:, 2 * D:], 3 * D)    # q,k: see the Qwen double block
            groups.append(self._gargs(A1=xm, lda1=D, B1=lw.W, K1=D, M=M, N=D, C_=c_, ldc=ldc, bias=lw.b, **kw))
        self._gemm_group(p, groups)
        ml, wo = w["mlp"], w["out"]
        cat = bb.get("cat")                       # [M, 5D] = [attn | gelu(mlp)] when proj_out carries an adapter
        gact, ldg = (cat[:, D:], 5 * D) if cat is not None else (A["g_j"], 4 * D)
        kw = self._site_fwd(p, ml, bb["site_mlp"], xm, D, M)
        self._gemm(p, A1=xm, lda1=D, B1=ml.W, K1=D, M=M, N=4 * D, C_=bb["h"], ldc=4 * D, bias=ml.b, epi=L.EPI_GELU, C2=gact, ldc2=ldg, **kw)
        nq, nk = w["norms"]
        p.c(lib.qfx_qk_norm_rope_fwd, _ptr(bb["qkv"]), _ptr(bb["sqk"]), _ptr(self.rope), _ptr(nq), _ptr(nk), _ptr(nq), _ptr(nk),
            B, S, T, H, dh, eps, self.NORM_FLAGS | 2, self.rope_bs)
        a = L.AttnArgs()
        a.B, a.S, a.S_pad, a.H, a.dh, a.scale = B, S, S_pad, H, dh, 1.0 / math.sqrt(dh)
        a.Q, a.K, a.V = _ptr(q2[:, 0:]), _ptr(q2[:, D:]), _ptr(q2[:, 2 * D:])
        a.ldq = a.ldk = a.ldv = 3 * D
        a.O, a.ldo, a.lse2 = _ptr(bb["ao"]), bb["ao"].stride(0), _ptr(bb["lse"])
        a.key_mask = _ptr(self.kmask)
        a.dsum = _ptr(A["dsum"])
        a.dO, a.lddo = _ptr(A["dao"]), D
        dq2 = A["dqkv"].view(M, 3 * D)
        a.dQ, a.dK, a.dV = _ptr(dq2[:, 0:]), _ptr(dq2[:, D:]), _ptr(dq2[:, 2 * D:])
        a.lddq = a.lddk = a.lddv = 3 * D
        self._fuse_qk_bwd(a, bb["sqk"], (nq, nk, nq, nk), self.NORM_FLAGS, eps)
        self.sattn_args.append(a)
        p.c(lib.qfx_attn_fwd, C.byref(a))
        if cat is not None:
            # adapted proj_out: ONE contraction over the kept [attn | gelu(mlp)] buffer + the LoRA K-extension (base rounded first)
            kw = self._site_fwd(p, wo, bb["site_out"], cat, 5 * D, M)
            if "y" in bb:
                kw.update(C2=bb["y"], ldc2=D)
            self._gemm(p, A1=cat, lda1=5 * D, B1=wo.W, ldb1=5 * D, K1=5 * D, M=M, N=D, C_=x_next, ldc=D, bias=wo.b, epi=L.EPI_GATE_RES,
                       aux=x, ldaux=D, gate=mod[:, 2 * D:3 * D], gate_bs=3 * D, rpb=S, row_mask=self.rmask["joint"], **kw)
            return
        # proj_out([attn | gelu(mlp)]) as a two-segment contraction, epilogue x + gate * y
        kw = dict(C2=bb["y"], ldc2=D) if "y" in bb else {}
        self._gemm(p, A1=bb["ao"].view(M, D), lda1=D, B1=wo.W, ldb1=5 * D, K1=D, A2=A["g_j"], lda2=4 * D, B2=wo.W[:, D:], ldb2=5 * D,
                   K2=4 * D, M=M, N=D, C_=x_next, ldc=D, bias=wo.b, epi=L.EPI_GATE_RES, aux=x, ldaux=D, gate=mod[:, 2 * D:3 * D],
                   gate_bs=3 * D, rpb=S, seg2_plain=1, row_mask=self.rmask["joint"], **kw)

    # ------------------------------------------------------------------ backward
    def _build_backward(self, P):
        A, B, D, S, H, dh, T, S_i = self.A, self.B, self.D, self.S, self.H, self.dh, self.T, self.S_i
        p = self.bwd
        cfg = self.model.config
        Ld, Ls = cfg.num_layers, cfg.num_single_layers
        rows, rpb, off = self.rows, self.rpb, self.off
        eps = 1e-6
        po = P["proj_out"]
        kw = self._site_bwd(p, po, A["site"]["proj_out"], A["dpred"], po.N, rows["img"], A["xn_out"], D)
        self._gemm(p, A1=A["dpred"], lda1=po.N, B1=po.WT, K1=po.N, M=rows["img"], N=D, C_=A["dxn"], ldc=D, **kw)
        mo = A["mod_out"][0]
        cur = 0
        if self.cond:
            for k in ("dmods", "dsmods", "dmod_out"):
                p.py(A[k].zero_)
            dmo = A["dmod_out"][0]
        if Ls:
            # tail LayerNorm backward per sample into the joint gradient; text rows of d(joint) are zero (dead text tail)
            dJ, dyg = A["dJ"][cur], A["dyg_j"]
            gl = A["smods"][Ls - 1]
            p.py(dJ.view(B, S, D)[:, :T].zero_)
            p.py(dyg.view(B, S, D)[:, :T].zero_)
            for b in range(B):
                r0 = b * S + T
                if self.cond:
                    self._mod_grad(p, dy=A["dxn"][b * S_i:], x=A["J"][Ls][r0:], rows=S_i, rpb=S_i, dshift=dmo[b:b + 1, D:2 * D],
                                   dscale=dmo[b:b + 1, 0:D], out_bs=2 * D,
                                   row_mask=self.rmask["img"][b * S_i:] if self.rmask["img"] is not None else None)
                p.c(lib.qfx_ln_modulate_bwd, _ptr(A["dxn"][b * S_i:]), _ptr(A["J"][Ls][r0:]), _ptr(mo[b:b + 1, 0:D]), 2 * D, None,
                    _ptr(gl[b:b + 1, 2 * D:3 * D]), 3 * D, _ptr(dJ[r0:]), _ptr(dyg[r0:]), S_i, D, S_i, eps,
                    _ptr(self.rmask["img"][b * S_i:]) if self.rmask["img"] is not None else None)
            for i in range(Ls - 1, -1, -1):
                nxt = cur ^ 1
                self._emit_single_bwd(p, P["singles"][i], A["sblk"][i], self.sattn_args[i], A["smods"][i], A["J"][i],
                                      dJ_out=A["dJ"][cur], dJ_in=A["dJ"][nxt], i=i, Ld=Ld)
                p.mark(f"single_transformer_blocks.{i}.")
                cur = nxt
            if Ld == 0:
                return
            dcur = 0   # single block 0 wrote the per-stream gradients into A["dX"][s][0] / A["dyg2"][s]
        else:
            dcur = 0
            modL = A["mods"][2 * (Ld - 1)]
            if self.cond:
                self._mod_grad(p, dy=A["dxn"], x=A["X"]["img"][Ld], rows=rows["img"], rpb=rpb["img"], dshift=dmo[:, D:2 * D],
                               dscale=dmo[:, 0:D], out_bs=2 * D, row_mask=self.rmask["img"])
            p.c(lib.qfx_ln_modulate_bwd, _ptr(A["dxn"]), _ptr(A["X"]["img"][Ld]), _ptr(mo[:, 0:D]), 2 * D, None,
                _ptr(modL[:, 5 * D:6 * D]), 6 * D, _ptr(A["dX"]["img"][dcur]), _ptr(A["dyg2"]["img"]), rows["img"], D, rpb["img"], eps, None)
        for i in range(Ld - 1, -1, -1):
            nxt = dcur ^ 1
            mods = {"img": A["mods"][2 * i], "txt": A["mods"][2 * i + 1]}
            gate_prev = None if i == 0 else {"img": A["mods"][2 * (i - 1)][:, 5 * D:6 * D], "txt": A["mods"][2 * (i - 1) + 1][:, 5 * D:6 * D]}
            self._emit_double_bwd(p, P["blocks"][i], A["blk"][i], self.attn_args[i], mods, {s: A["X"][s][i] for s in ("img", "txt")},
                                  dx2={s: A["dX"][s][dcur] for s in ("img", "txt")}, out_dx={s: A["dX"][s][nxt] for s in ("img", "txt")},
                                  gate_prev=gate_prev, last=(i + 1 == Ld and Ls == 0), first=(i == 0 and not self.full_bwd),
                                  norm_flags=self.NORM_FLAGS,
                                  dmods=({"img": A["dmods"][2 * i], "txt": A["dmods"][2 * i + 1]} if self.cond else None))
            p.mark(f"transformer_blocks.{i}.")
            dcur = nxt
        if self.in_grad:   # the embedders' adapters: d(block-0 input) = A["dX"][s][dcur]; their own inputs carry no gradient
            self._site_bwd(p, P["x_in"], A["site"]["x_in"], A["dX"]["img"][dcur], D, rows["img"], A["in_img"], cfg.in_channels)
            self._site_bwd(p, P["c_in"], A["site"]["c_in"], A["dX"]["txt"][dcur], D, rows["txt"], A["in_txt"], P["c_in"].K)
        if self.cond:
            self.cond_head.emit_backward(p)

    def _emit_single_bwd(self, p, w, bb, a, mod, x, dJ_out, dJ_in, i, Ld):
        """In: dJ_out = d(block output) [B*S,D], A["dyg_j"] = gate*dJ_out.  Out: dJ_in and A["dyg_j"] = gate_prev*dJ_in
        (or, for the first single block, the per-stream gradients of the last double block)."""
        A, B, D, S, H, dh, T, S_i = self.A, self.B, self.D, self.S, self.H, self.dh, self.T, self.S_i
        S_pad = self.S_pad
        rows, rpb, off = self.rows, self.rpb, self.off
        eps = 1e-6
        M = B * S
        wo, ml = w["out"], w["mlp"]
        grp = w["qkv_lora"]
        ldA2 = A["A2"].stride(0)
        dao2 = A["dao"].view(M, D)
        dq2 = A["dqkv"].view(M, 3 * D)
        # d[attn | mlp] = (gate*dx) W_out : attention part -> dO, mlp part through gelu' -> A2[:, :4D]   (+ proj_out's adapter)
        kwa, kwm = {}, {}
        # the block's weight-gradient problems (up to 8) go out as batched launches right before the LayerNorm backward overwrites
        # dyg_j -- their operands (dyg_j, dqkv, A2, v^T scratch, the block's kept buffers) stay intact until then; one launch each
        # they were 8 x 38 latency-bound launches on the main stream (the FLUX programs keep their gradients there)
        gl = [] if os.environ.get("QFX_FLUX_SINGLE_BATCH", "1") != "0" else None     # None: one launch per problem (A/B switch)
        if wo.lora is not None:
            kwo = self._site_bwd(p, wo, bb["site_out"], A["dyg_j"], D, M, bb["cat"], 5 * D, defer=gl)
            kwa = dict(kwo, B2=wo.lora.WeT[:D])
            kwm = dict(kwo, B2=wo.lora.WeT[D:])
        self._gemm(p, A1=A["dyg_j"], lda1=D, B1=wo.WT, K1=D, M=M, N=D, C_=dao2, ldc=D, **kwa)
        self._gemm(p, A1=A["dyg_j"], lda1=D, B1=wo.WT[D:], K1=D, M=M, N=4 * D, C_=A["A2"], ldc=ldA2, epi=L.EPI_DGELU, aux=bb["h"],
                   ldaux=4 * D, **kwm)
        q2 = bb["qkv"].view(M, 3 * D)
        ops.emit_attn_backward(p, a, A)      # two-pass pair, or the one-pass kernel (QFX_ATTN_BWD)
        if not a.qk_saved:
            nq, nk = w["norms"]
            p.c(lib.qfx_qk_norm_rope_bwd, _ptr(A["dqkv"]), _ptr(bb["sqk"]), _ptr(self.rope), _ptr(nq), _ptr(nk), _ptr(nq), _ptr(nk),
                B, S, T, H, dh, eps, self.NORM_FLAGS, self.rope_bs)
        K2 = 4 * D
        if grp is not None:
            Rp, Kext = grp["Rp"], grp["Kext"]
            Vth, Vtl = A["Vt_j"]
            Uth, Utl = bb["Uqkv"]
            dl = [] if gl is not None else None     # the q / k / v down projections of dqkv: one launch
            for sec in range(3):
                lo = w["qkv"][sec].lora
                if lo is None:
                    continue
                sl = slice(sec * Rp, (sec + 1) * Rp)
                self._down(p, X=dq2[:, sec * D:], ldx=3 * D, M=M, K=D, W_hi=lo.Bt_hi, W_lo=lo.Bt_lo, ldw=lo.Bt_hi.stride(0), R=Rp,
                           Ut=(Vth[sl], Vtl[sl]), ext=A["A2"][:, 4 * D + sec * Kext:], ld_ext=ldA2, defer=dl)
                self._grad(p, Vt=(Uth[sl], Utl[sl]), R=Rp, r_valid=lo.r, X=dq2[:, sec * D:], ldx=3 * D, M=M, K=D, G=lo.gB, g_sr=1,
                           g_sc=lo.r, out_scale=lo.scale, defer=gl)
            if dl:
                self._flush_batch(p, dl, L.LoraDownArgs, lib.qfx_lora_down_batch)
            los = [w["qkv"][sec].lora for sec in range(3)]
            if all(l is not None for l in los):
                self._grad(p, Vt=(Vth[:3 * Rp], Vtl[:3 * Rp]), R=3 * Rp, r_valid=los[0].r, group_R=Rp, X=bb["xm"], ldx=D, M=M, K=D,
                           G=[l.gA for l in los], g_sr=D, g_sc=1, defer=gl)
            else:
                for sec, lo in enumerate(los):
                    if lo is not None:
                        sl = slice(sec * Rp, (sec + 1) * Rp)
                        self._grad(p, Vt=(Vth[sl], Vtl[sl]), R=Rp, r_valid=lo.r, X=bb["xm"], ldx=D, M=M, K=D, G=lo.gA, g_sr=D, g_sc=1,
                                   defer=gl)
            K2 = 4 * D + 3 * Kext
        if ml.lora is not None:
            # proj_mlp's adapter: v = d(mlp pre-act) (sB)^T goes into the last K-extension columns of A2; dB / dA as for any site
            lo = ml.lora
            sb = bb["site_mlp"]
            self._down(p, X=A["A2"], ldx=ldA2, M=M, K=4 * D, W_hi=lo.Bt_hi, W_lo=lo.Bt_lo, ldw=lo.Bt_hi.stride(0), R=lo.Rp, Ut=sb["V"],
                       ext=A["A2"][:, K2:], ld_ext=ldA2)
            self._grad(p, Vt=sb["U"], R=lo.Rp, r_valid=lo.r, X=A["A2"], ldx=ldA2, M=M, K=4 * D, G=lo.gB, g_sr=1, g_sc=lo.r, out_scale=lo.scale,
                       defer=gl)
            self._grad(p, Vt=sb["V"], R=lo.Rp, r_valid=lo.r, X=bb["xm"], ldx=D, M=M, K=D, G=lo.gA, g_sr=D, g_sc=1, defer=gl)
            K2 += lo.Kext
        # d(norm_x) = [dq|dk|dv] Wqkv + [d mlp | LoRA v] [W_mlp ; A]
        if getattr(self.model, "_quant", None) == "mxfp8-fb" and D % 128 == 0 and D >= 1024:
            # low-precision trunk: the four frozen contractions as one MX-FP8 GEMM over K = 3D + 4D, adapters as its bf16 K-extension
            self._gemm_mxfp8_cat(p, [(dq2, 3 * D, 3 * D, w["qkvT"]), (A["A2"], ldA2, 4 * D, w["B2"])], M=M, N=D, C_=A["dxm_j"], ldc=D,
                                 ext=(A["A2"][:, 4 * D:], ldA2, w["B2"][:, 4 * D:], w["B2"].stride(0), K2 - 4 * D))
        else:
            self._gemm(p, A1=dq2, lda1=3 * D, B1=w["qkvT"], K1=3 * D, A2=A["A2"], lda2=ldA2, B2=w["B2"], ldb2=w["B2"].stride(0), K2=K2,
                       M=M, N=D, C_=A["dxm_j"], ldc=D, seg2_plain=1)
        if self.cond:   # d(shift, scale, gate) of the single block's AdaLayerNormZeroSingle
            dm = A["dsmods"][i]
            self._mod_grad(p, dy=A["dxm_j"], x=x, rows=M, rpb=S, dshift=dm[:, 0:D], dscale=dm[:, D:2 * D], dgate=dm[:, 2 * D:3 * D],
                           dxo=dJ_out, y=bb["y"], out_bs=3 * D, row_mask=self.rmask["joint"])
        if gl:
            self._flush_batch(p, gl, L.LoraGradArgs, lib.qfx_lora_grad_batch)
        if i > 0:
            gp = A["smods"][i - 1][:, 2 * D:3 * D]
            p.c(lib.qfx_ln_modulate_bwd, _ptr(A["dxm_j"]), _ptr(x), _ptr(mod[:, D:2 * D]), 3 * D, _ptr(dJ_out), _ptr(gp), 3 * D,
                _ptr(dJ_in), _ptr(A["dyg_j"]), M, D, S, eps, _ptr(self.rmask["joint"]))
        elif Ld > 0:
            # hand the gradient over to the last double block: per (sample, stream) row ranges of the joint buffer
            mods_prev = {"img": A["mods"][2 * (Ld - 1)], "txt": A["mods"][2 * (Ld - 1) + 1]}
            for b in range(B):
                for s in ("txt", "img"):
                    r0, n, c0 = b * S + off[s], rpb[s], b * rpb[s]
                    gp = mods_prev[s][b:b + 1, 5 * D:6 * D]
                    p.c(lib.qfx_ln_modulate_bwd, _ptr(A["dxm_j"][r0:]), _ptr(x[r0:]), _ptr(mod[b:b + 1, D:2 * D]), 3 * D,
                        _ptr(dJ_out[r0:]), _ptr(gp), 6 * D, _ptr(A["dX"][s][0][c0:]), _ptr(A["dyg2"][s][c0:]), n, D, n, eps,
                        _ptr(self.rmask[s][c0:]) if self.rmask[s] is not None else None)

    def set_multires(self, img_ids_b: torch.Tensor, valid_lens):
        """Per-step dynamic state of the multi-resolution path (transformer_flux_custom.py:499-616): per-sample RoPE with
        identity rotation on padding, additive key mask (0 / -inf), row masks of the padded image tokens."""
        cfg = self.model.config
        B, S, T, S_i = self.B, self.S, self.T, self.S_i
        dev = self.model.device
        rope = torch.zeros(B, S, self.dh // 2, 2)
        rope[..., 0] = 1.0
        km = torch.zeros(B, S)
        rm = torch.zeros(B, S_i)
        txt = torch.zeros(T, 3)
        for b in range(B):
            n = int(valid_lens[b])
            ids = torch.cat([txt, img_ids_b[b, :n].float().cpu()], dim=0)
            rope[b, : T + n] = flux_joint_rope(ids, cfg.axes_dims_rope)
            km[b, T + n:] = float("-inf")
            rm[b, :n] = 1.0
        A = self.A
        A["rope_b"].copy_(rope.to(dev, non_blocking=True))
        A["kmask"].copy_(km.to(dev, non_blocking=True))
        A["rm_img"].copy_(rm.reshape(-1).to(dev, non_blocking=True))
        rj = torch.ones(B, S)
        rj[:, T:] = rm
        A["rm_joint"].copy_(rj.reshape(-1).to(dev, non_blocking=True))

    # ------------------------------------------------------------------ execution
    def run_forward(self, inputs, encoder_hidden_states, timestep):
        hidden_states, pooled, guidance = inputs
        A = self.A
        self._copy_rows(A["in_img"].view(self.B, self.S_i, -1), hidden_states)
        A["in_txt"].view(self.B, self.T, -1)[:, :, : encoder_hidden_states.shape[-1]].copy_(encoder_hidden_states)
        A["pooled"].copy_(pooled)
        A["t"].copy_(timestep.reshape(self.B).to(F32))
        if guidance is not None:
            A["gd"].copy_(guidance.reshape(self.B).to(F32))
        self.model.refresh_lora_operands()
        self.fwd.run()
        return A["out"].view(self.B, self.S_i, -1)
