"""MI355X-native drop-in for the reference's QwenImageTransformer2DModel
(src/qflux/models/transformer_qwenimage.py:497-672) on the LoRA-training hot path.

Same constructor arguments, state-dict keys, forward signature and return value; `add_adapter`
reproduces peft's naming (X.base_layer / X.lora_A.<name> / X.lora_B.<name>).  The forward and the
backward of the whole DiT are ONE autograd node: for a given shape signature a *launch program*
(flat list of libqfx C-ABI calls with pre-built argument structs over a persistent HBM arena) is
built once and replayed every step -- no per-op autograd graph, no gradient checkpointing (288 GB),
frozen base weights never get a dW, LoRA dA/dB accumulate straight into the flat gradient buffer.

Numerics follow the reference's bf16 eager graph at every rounding point (see csrc/*.hip).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..dp import DataParallelMixin
from ..plan_cache import PlanCache, ladder
from ..modules import (LoraConfig, LoraStore, QfxLinear, QfxLoraLinear, QfxRMSNorm, init_lora_, match_target)
from ..rope import QwenEmbedRope, normalize_img_shapes, qwen_joint_rope

lib = L.lib
BF = torch.bfloat16
F32 = torch.float32


# ----------------------------------------------------------------------------------------------
# holder modules (names == reference state-dict keys)
# ----------------------------------------------------------------------------------------------
class _GELUProj(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = QfxLinear(dim_in, dim_out)


class QfxFeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, dim * mult), nn.Identity(), QfxLinear(dim * mult, dim)])


class QfxAttention(nn.Module):
    def __init__(self, dim, heads, dim_head, eps=1e-6):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj"):
            setattr(self, n, QfxLinear(dim, inner))
        self.to_out = nn.ModuleList([QfxLinear(inner, dim), nn.Identity()])
        self.to_add_out = QfxLinear(inner, dim)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            setattr(self, n, QfxRMSNorm(dim_head, eps))


class QwenImageTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, eps=1e-6):
        super().__init__()
        self.img_mod = nn.Sequential(nn.Identity(), QfxLinear(dim, 6 * dim))   # index 1 == the Linear (index 0 is SiLU)
        self.attn = QfxAttention(dim, num_attention_heads, attention_head_dim, eps)
        self.img_mlp = QfxFeedForward(dim)
        self.txt_mod = nn.Sequential(nn.Identity(), QfxLinear(dim, 6 * dim))
        self.txt_mlp = QfxFeedForward(dim)


class _TimestepEmbedder(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear_1 = QfxLinear(256, dim)
        self.linear_2 = QfxLinear(dim, dim)


class _TimeTextEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedder(dim)


class _AdaLNOut(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear = QfxLinear(dim, 2 * dim)


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# ----------------------------------------------------------------------------------------------
# prepared device-side weights
# ----------------------------------------------------------------------------------------------
class _LoraW:
    __slots__ = ("mod", "r", "Rp", "Kext", "scale", "A_hi", "A_lo", "Bt_hi", "Bt_lo", "We", "WeT", "gA", "gB", "A_hl", "Bt_hl", "hl_dh", "A_fr", "fr_row0", "fr_nf")


class _LinW:
    __slots__ = ("W", "b", "WT", "lora", "N", "K", "mod")

    def __init__(self, mod, need_T: bool):
        self.mod = mod
        base = mod.base_layer if isinstance(mod, QfxLoraLinear) else mod
        self.W = base.weight.data
        assert self.W.is_contiguous() and self.W.dtype == BF
        self.b = base.bias.data if base.bias is not None else None
        self.N, self.K = self.W.shape
        self.WT = self.W.t().contiguous() if need_T else None
        self.lora = None


def _ceil(a, b):
    return (a + b - 1) // b * b


class _Prog:
    """A flat launch program: list of (callable, args); C calls get the stream appended."""

    def __init__(self):
        self.calls = []
        self.keep = []
        self.marks = []   # (call index, parameter-name prefix): every LoRA gradient under `prefix` is final after calls[:index]

    def c(self, fn, *args):
        self.calls.append((fn, args))

    def c_side(self, fn, *args):
        """C call issued on the program's side stream (leaf work that overlaps the main stream; joined by explicit events)."""
        self.calls.append((fn, args, True))

    def c_on(self, stream, fn, *args):
        """C call issued on `stream` (a torch.cuda.Stream kept alive by the caller; forked / joined by explicit events)."""
        self.calls.append((fn, args, stream))

    def py(self, fn):
        self.calls.append((None, fn))

    def mark(self, prefix: str):
        self.marks.append((len(self.calls), prefix))

    side = None   # torch.cuda.Stream for c_side calls (set by the plan that uses them)

    def run(self, start: int = 0, end: int | None = None):
        st = torch.cuda.current_stream().cuda_stream
        for ent in self.calls[start:end]:
            fn, args = ent[0], ent[1]
            if fn is None:
                args()
            else:
                rc = fn(*args, st if len(ent) < 3 else (self.side if ent[2] is True else ent[2]).cuda_stream)
                if rc != 0:
                    raise L.QfxError(f"{fn.__name__} failed with code {rc}")


def _ptr(t):
    if t is None:
        return None
    return t.data_ptr() if isinstance(t, torch.Tensor) else t


# ----------------------------------------------------------------------------------------------
class QwenImageTransformer2DModel(DataParallelMixin, nn.Module):
    """See module docstring.  Reference: transformer_qwenimage.py:497-672."""

    _supports_gradient_checkpointing = True

    def __init__(self, patch_size: int = 2, in_channels: int = 64, out_channels: int | None = 16, num_layers: int = 60,
                 attention_head_dim: int = 128, num_attention_heads: int = 24, joint_attention_dim: int = 3584,
                 guidance_embeds: bool = False, axes_dims_rope=(16, 56, 56)):
        super().__init__()
        if attention_head_dim not in (64, 128):
            raise ValueError("qflux_amd attention kernels support head dims 64 and 128")
        self.config = _Cfg(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                           attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads,
                           joint_attention_dim=joint_attention_dim, guidance_embeds=guidance_embeds,
                           axes_dims_rope=tuple(axes_dims_rope))
        self.out_channels = out_channels or in_channels
        self.inner_dim = num_attention_heads * attention_head_dim
        D = self.inner_dim
        self.pos_embed = QwenEmbedRope(theta=10000, axes_dim=list(axes_dims_rope), scale_rope=True)   # :546 (no parameters)
        self.time_text_embed = _TimeTextEmbed(D)
        self.txt_norm = QfxRMSNorm(joint_attention_dim, eps=1e-6)
        self.img_in = QfxLinear(in_channels, D)
        self.txt_in = QfxLinear(joint_attention_dim, D)
        self.transformer_blocks = nn.ModuleList(
            [QwenImageTransformerBlock(D, num_attention_heads, attention_head_dim) for _ in range(num_layers)])
        self.norm_out = _AdaLNOut(D)
        self.proj_out = QfxLinear(D, patch_size * patch_size * self.out_channels)
        self.gradient_checkpointing = False
        self._lora = LoraStore(self)
        self._adapter_name = None
        self._prepared = None      # prepared weights
        self._lora_prep = None     # packed-operand buffers + descriptors
        self._plans = PlanCache()
        self._version = 0
        self._adapter_gen = 0      # bumped ONLY by add_adapter / load_lora_adapter / load_state_dict: what a data-parallel resync keys on

    # ------------------------------------------------------------------ reference-surface methods
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: str | None = None, torch_dtype=BF, device_map=None,
                        variant: str | None = None, use_safetensors: bool = True, **unused):
        """diffusers ModelMixin.from_pretrained for a LOCAL checkpoint folder (the reference's loader calls it with
        subfolder="transformer", torch_dtype=weight_dtype, device_map="cpu": src/qflux/models/load_model.py:34-47,
        flux_kontext_loader.py:145-181): `config.json` -> constructor arguments (unknown keys ignored), weights from
        `diffusion_pytorch_model[.<variant>].safetensors` or its sharded form (`...safetensors.index.json` -> weight_map).
        Hub ids are not resolved here (no network on the training nodes: pass the snapshot directory).  Extra keyword
        arguments of the reference's call sites (attn_implementation, ...) are accepted and ignored."""
        import inspect
        import json
        import os
        from safetensors import safe_open
        root = str(pretrained_model_name_or_path)
        if subfolder:
            root = os.path.join(root, subfolder)
        if not os.path.isdir(root):
            raise FileNotFoundError(f"{root}: from_pretrained needs a local checkpoint directory (hub ids are not resolved offline)")
        if not use_safetensors:
            raise NotImplementedError("only safetensors checkpoints are read")
        with open(os.path.join(root, "config.json")) as f:
            cfg = json.load(f)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        kwargs = {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in accepted}
        dtype = torch_dtype or BF
        if dtype != BF:
            raise NotImplementedError("the MI355X path keeps the frozen trunk in bf16 (weight_dtype of the reference's configs)")
        device = torch.device("cpu")
        if isinstance(device_map, (str, torch.device)) and str(device_map) not in ("cpu", "auto"):
            device = torch.device(device_map)
        with torch.device(device):
            model = cls(**kwargs)
        stem = "diffusion_pytorch_model" + (f".{variant}" if variant else "")
        index = os.path.join(root, stem + ".safetensors.index.json")
        if os.path.exists(index):
            with open(index) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
        else:
            files = [stem + ".safetensors"]
        own = dict(model.named_parameters())
        seen = set()
        with torch.no_grad():
            for fn in files:
                with safe_open(os.path.join(root, fn), framework="pt", device=str(device)) as sf:
                    for key in sf.keys():
                        if key not in own:
                            raise KeyError(f"unexpected key {key!r} in {fn}")
                        t = sf.get_tensor(key)
                        if tuple(t.shape) != tuple(own[key].shape):
                            raise ValueError(f"{key}: checkpoint shape {tuple(t.shape)} != model shape {tuple(own[key].shape)}")
                        own[key].copy_(t.to(own[key].dtype))
                        seen.add(key)
        missing = sorted(set(own) - seen)
        if missing:
            raise KeyError(f"{len(missing)} parameters missing from the checkpoint, e.g. {missing[:3]}")
        model._invalidate()
        return model

    @property
    def device(self):
        return self.norm_out.linear.weight.device if isinstance(self.norm_out.linear, QfxLinear) else next(self.parameters()).device

    @property
    def dtype(self):
        return BF      # the frozen trunk is bf16 (adapters are fp32)

    def enable_gradient_checkpointing(self):
        """Accepted for API parity (base_trainer.py:324-325); a no-op: activations stay resident in HBM."""
        self.gradient_checkpointing = True

    def _invalidate(self):
        self._prepared = None
        self._lora_prep = None
        self._plans = PlanCache()
        self.__dict__.pop("_wq_cache", None)      # quantised copies of the frozen weights (low-precision trunk)
        self._version += 1

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._invalidate()
        self._adapter_gen += 1
        return r

    def add_adapter(self, adapter_config, adapter_name: str = "default", generator: torch.Generator | None = None):
        """peft add_adapter look-alike (base_trainer.py:939): wrap matching Linears, freeze all but 'lora' params."""
        cfg = adapter_config
        names = [n for n, m in self.named_modules() if isinstance(m, QfxLinear) and ".base_layer" not in n
                 and not n.endswith("base_layer") and match_target(n, cfg.target_modules)]
        for n in names:
            if not self._lora_supported(n):
                raise NotImplementedError(
                    f"LoRA target '{n}': the fused path covers the attention projections, the feed-forward linears, the embedders "
                    f"and the output projection (FLUX: also the single-block proj_mlp / proj_out) and the conditioning head "
                    f"(timestep / guidance / text embedders, AdaLN modulation linears)")
        for n in names:
            parent_name, _, child = n.rpartition(".")
            parent = self.get_submodule(parent_name)
            base = parent[int(child)] if child.isdigit() else getattr(parent, child)
            wrapped = QfxLoraLinear(base, cfg.r, cfg.lora_alpha, adapter_name)
            init_lora_(wrapped, cfg.init_lora_weights, generator)
            if child.isdigit():
                parent[int(child)] = wrapped
            else:
                setattr(parent, child, wrapped)
        self._adapter_name = adapter_name
        for pn, p in self.named_parameters():
            p.requires_grad_("lora" in pn)
        # diffusers' PeftAdapterMixin.add_adapter keeps the config under `peft_config[adapter_name]`: peft's
        # get_peft_model_state_dict(model, adapter_name=...) -- what BaseTrainer.save_lora calls on the unwrapped DiT
        # (base_trainer.py:870-872) -- reads `.peft_type`, `.bias`, `.use_dora`, `.target_modules` ... from it and then filters
        # model.state_dict() by "lora_" + adapter name.  A real peft.LoraConfig passes through untouched; the stand-in
        # (modules.LoraConfig) carries the same fields.
        if not isinstance(getattr(self, "peft_config", None), dict):
            self.peft_config = {}
        self.peft_config[adapter_name] = cfg
        self._hf_peft_config_loaded = True
        self._invalidate()
        self._adapter_gen += 1
        # The reference wraps its LoRA container in DDP right after this call (base_trainer.py:384-393); the kernels write dA / dB
        # straight into the flat gradient buffer, so the model exchanges them itself.  Under an initialised multi-rank process
        # group that happens without an extra line in the trainer (no-op otherwise; call enable_data_parallel(...) again to
        # choose a process group / bucket size).
        # Opt-out: QFX_AUTO_DP=0 (a trainer that wraps / exchanges the LoRA gradients itself).
        if (self._dp is None and os.environ.get("QFX_AUTO_DP", "1") != "0" and dist.is_available() and dist.is_initialized()
                and dist.get_world_size() > 1):
            self.enable_data_parallel()
        return names

    _LORA_SUFFIXES = ("attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out.0", "attn.add_q_proj", "attn.add_k_proj",
                      "attn.add_v_proj", "attn.to_add_out",
                      "img_mlp.net.0.proj", "img_mlp.net.2", "txt_mlp.net.0.proj", "txt_mlp.net.2",      # Qwen feed-forwards
                      "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2")           # FLUX double blocks

    # embedders / output projection: plain GEMM sites outside the blocks (K-extension + the two rank-r gradient launches)
    _HEAD_SITES = {"img_in": "img_in", "txt_in": "txt_in", "proj_out": "proj_out"}      # module name -> key in the prepared weights

    # conditioning head (M = batch rows): timestep embedder, AdaLN modulation linears -- cond_hip.py when adapted
    _COND_SUFFIXES = ("timestep_embedder.linear_1", "timestep_embedder.linear_2", "img_mod.1", "txt_mod.1", "norm_out.linear")

    def _lora_supported(self, name: str) -> bool:
        if name in self._HEAD_SITES or name.endswith(self._COND_SUFFIXES):
            return True
        return name.startswith("transformer_blocks.") and name.endswith(self._LORA_SUFFIXES)

    def _cond_modules(self):
        te = self.time_text_embed.timestep_embedder
        mods = [te.linear_1, te.linear_2, self.norm_out.linear]
        for blk in self.transformer_blocks:
            mods += [blk.img_mod[1], blk.txt_mod[1]]
        return mods

    @property
    def cond_lora(self) -> bool:
        """True when a linear of the conditioning head carries an adapter (the head is then emitted by cond_hip.CondHeadHip)."""
        return any(isinstance(m, QfxLoraLinear) for m in self._cond_modules())

    def set_adapter(self, adapter_name):
        """PeftAdapterMixin.set_adapter (base_trainer.py:940-941): make `adapter_name` the active adapter of every wrapped linear.
        Plans bake adapter pointers, rank and scaling in at build time, so a change of the active adapter drops them."""
        if isinstance(adapter_name, (list, tuple)):
            if len(adapter_name) != 1:
                raise NotImplementedError("one active adapter at a time")
            adapter_name = adapter_name[0]
        wrapped = [m for m in self.modules() if isinstance(m, QfxLoraLinear)]
        if wrapped and not any(adapter_name in m.lora_A for m in wrapped):
            raise ValueError(f"Adapter {adapter_name!r} not found (known: {sorted({k for m in wrapped for k in m.lora_A})})")
        changed = self._adapter_name != adapter_name
        for m in wrapped:
            if adapter_name in m.lora_A and m.active_adapter != adapter_name:
                m.active_adapter, changed = adapter_name, True
        self._adapter_name = adapter_name
        if changed:
            self._invalidate()

    def merge_adapter(self):
        """PeftAdapterMixin.merge_adapter as BaseTrainer.merge_lora calls it (base_trainer.py:413-416): fold every adapter into its
        base weight; the LoRA segment of the GEMMs then carries a zero scale (forward == merged base layer alone)."""
        for m in self.modules():
            if isinstance(m, QfxLoraLinear):
                m.merge()
        self._merged = True
        self._invalidate()

    def unmerge_adapter(self):
        for m in self.modules():
            if isinstance(m, QfxLoraLinear):
                m.unmerge()
        self._merged = False
        self._invalidate()

    def save_lora_weights(self, save_folder, style="diffusers"):
        """pytorch_lora_weights.safetensors as BaseTrainer.save_lora writes it (base_trainer.py:858-875)."""
        from ..lora_io import save_lora_weights
        return save_lora_weights(self, save_folder, style)

    def load_lora_adapter(self, path, adapter_name="default", lora_alpha=None):
        """DIFFUSERS- or PEFT-style LoRA file/folder (base_trainer.py:977-999)."""
        from ..lora_io import load_lora_adapter
        names = load_lora_adapter(self, path, adapter_name, lora_alpha)
        self._invalidate()
        self._adapter_gen += 1
        return names

    def quantize_trunk(self, mode: str | None = "mxfp8"):
        """Low-precision trunk switch (the reference's `model.quantize: true`, base_trainer.py:617-621,919-927 ->
        quantize_model_to_fp8): "mxfp8" runs the forward GEMMs of the block linears on the block-scaled FP8 MFMA
        (OCP MX-FP8: e4m3 elements, E8M0 scale per 32 K elements; weights quantised once, activations per launch);
        "mxfp8-fb" also the dX GEMMs of the backward (dY and the transposed weight copy quantised along the contraction);
        None restores the bf16 trunk.  Adapters, biases, norms, attention, the rank-r gradient kernels stay bf16 / fp32."""
        if mode not in (None, "mxfp8", "mxfp8-fb"):
            raise ValueError(f"unknown trunk quantisation {mode!r} (supported: 'mxfp8' = forward GEMMs, 'mxfp8-fb' = forward + dX GEMMs)")
        self._quant = mode
        self.__dict__.pop("_wq_cache", None)
        self._plans = PlanCache()
        self._version += 1
        return self

    def lora_parameters(self):
        return [p for n, p in self.named_parameters() if "lora_" in n]

    @property
    def lora_store(self) -> LoraStore:
        self._ensure_lora_store()
        return self._lora

    # ------------------------------------------------------------------ preparation
    def _ensure_lora_store(self):
        if not self._lora.is_consistent(self.device):
            self._lora.rebuild(self.device)
            self._lora_prep = None
            self._plans = PlanCache()
        self._lora.ensure_grads()

    def _prepare(self):
        """Contiguous bf16 weights + resident transposed copies for the dX GEMMs (2x weight memory, by design)."""
        if self._prepared is not None:
            return self._prepared
        assert self.device.type == "cuda", "qflux_amd runs on the GPU only (no CPU fallback)"
        P = {"blocks": []}
        for blk in self.transformer_blocks:
            a = blk.attn
            w = {}
            for s, names in (("img", ("to_q", "to_k", "to_v")), ("txt", ("add_q_proj", "add_k_proj", "add_v_proj"))):
                qkv = [_LinW(getattr(a, n), False) for n in names]
                w[s + ".qkv"] = qkv
                w[s + ".qkvT"] = torch.cat([l.W for l in qkv], dim=0).t().contiguous()       # [D, 3D]
            w["img.o"] = _LinW(a.to_out[0], True)
            w["txt.o"] = _LinW(a.to_add_out, True)
            for s, mlp in (("img", blk.img_mlp), ("txt", blk.txt_mlp)):
                w[s + ".fc1"] = _LinW(mlp.net[0].proj, True)
                w[s + ".fc2"] = _LinW(mlp.net[2], True)
            w["img.mod"] = _LinW(blk.img_mod[1], False)
            w["txt.mod"] = _LinW(blk.txt_mod[1], False)
            w["norms"] = (a.norm_added_q.weight.data, a.norm_added_k.weight.data, a.norm_q.weight.data, a.norm_k.weight.data)
            P["blocks"].append(w)
        P["img_in"] = _LinW(self.img_in, False)
        P["txt_in"] = _LinW(self.txt_in, False)
        P["t1"] = _LinW(self.time_text_embed.timestep_embedder.linear_1, False)
        P["t2"] = _LinW(self.time_text_embed.timestep_embedder.linear_2, False)
        P["norm_out"] = _LinW(self.norm_out.linear, False)
        P["proj_out"] = _LinW(self.proj_out, True)
        dev = self.device
        mods = [w[s + ".mod"] for w in P["blocks"] for s in ("img", "txt")]
        P["mod_W"] = torch.tensor([m.W.data_ptr() for m in mods], dtype=torch.int64, device=dev)
        P["mod_b"] = torch.tensor([m.b.data_ptr() for m in mods], dtype=torch.int64, device=dev)
        for key in ("t1", "t2", "norm_out"):
            P[key + "_Wp"] = torch.tensor([P[key].W.data_ptr()], dtype=torch.int64, device=dev)
            P[key + "_bp"] = torch.tensor([P[key].b.data_ptr()], dtype=torch.int64, device=dev)
        self._prepared = P
        return P

    def _prepare_lora(self):
        """Packed bf16 operand buffers for every adapter (+ the device descriptor array for qfx_lora_pack)."""
        if self._lora_prep is not None:
            return self._lora_prep
        P = self._prepare()
        self._ensure_lora_store()
        dev = self.device
        D = self.inner_dim
        descs = []
        keep = []
        max_dim = 1
        for w, blk in zip(P["blocks"], self.transformer_blocks):
            max_dim = max(max_dim, self._prep_double_lora(w, blk.attn, descs))
        max_dim = max(max_dim, self._prep_head_lora(P, descs))
        prep = dict(n=len(descs), max_dim=max_dim, descs=None)
        if descs:
            arr = (L.LoraPackArgs * len(descs))(*descs)
            prep["descs"] = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._lora_prep = prep
        return prep

    def _prep_site_lora(self, lw, descs):
        """Operand buffers of ONE adapted linear that is not part of a q/k/v group (own A_hi/A_lo and WeT)."""
        m = lw.mod
        if not isinstance(m, QfxLoraLinear):
            lw.lora = None
            return 1
        if m.in_features % 32:
            raise NotImplementedError(f"LoRA on a linear with in_features={m.in_features}: the rank-r kernels contract in 32-wide steps")
        dev = self.device
        r = m.r[m.active_adapter]
        Rp, Kext = _ceil(r, 16), _ceil(3 * _ceil(r, 16), 64)
        K = m.in_features
        lw.lora = self._make_lora(m, Rp, Kext, torch.zeros(Rp, K, dtype=BF, device=dev), torch.zeros(Rp, K, dtype=BF, device=dev),
                                  torch.zeros(K, Kext, dtype=BF, device=dev), dev)
        descs.append(self._pack_desc(lw.lora))
        return max(lw.N, K)

    def _prep_head_lora(self, P, descs):
        return max(self._prep_site_lora(P[key], descs) for key in self._HEAD_SITES.values())

    def _prep_qkv_lora(self, w, prefix, mods, descs, WeT=None, hl=None):
        """LoRA operand buffers of a q/k/v projection group sharing one input: concatenated A_hi/A_lo [3Rp,D] (one fused
        down-projection) and WeT [D, 3*Kext] (one K-extension of the dX GEMM).  WeT may be a column slice of a bigger B2."""
        dev, D = self.device, self.inner_dim
        lmods = [m for m in mods if isinstance(m, QfxLoraLinear)]
        w[prefix + "qkv_lora"] = None
        if not lmods:
            return 1
        r = lmods[0].r[lmods[0].active_adapter]
        Rp, Kext = _ceil(r, 16), _ceil(3 * _ceil(r, 16), 64)
        A_hi = torch.zeros(3 * Rp, D, dtype=BF, device=dev)
        A_lo = torch.zeros_like(A_hi)
        if WeT is None:
            WeT = torch.zeros(D, 3 * Kext, dtype=BF, device=dev)
        else:
            WeT = WeT(Kext)
        # the same rows in MFMA-fragment order for the fused LayerNorm + down projection (qfx_ln_down_args.W_fr): [hi image | lo image]
        A_fr = torch.zeros(2 * 3 * Rp * D, dtype=BF, device=dev) if (D % 32 == 0 and os.environ.get("QFX_LN_DOWN_FRAG", "1") != "0") else None
        w[prefix + "qkv_lora"] = dict(Rp=Rp, Kext=Kext, A_hi=A_hi, A_lo=A_lo, A_fr=A_fr, WeT=WeT, present=[isinstance(m, QfxLoraLinear) for m in mods])
        md = 1
        for sec, (m, lw) in enumerate(zip(mods, w[prefix + "qkv"])):
            if not isinstance(m, QfxLoraLinear):
                continue
            lo = self._make_lora(m, Rp, Kext, A_hi[sec * Rp:(sec + 1) * Rp], A_lo[sec * Rp:(sec + 1) * Rp],
                                 WeT[:, sec * Kext:(sec + 1) * Kext], dev, hl=hl)
            lo.A_fr, lo.fr_row0, lo.fr_nf = A_fr, sec * Rp, 3 * Rp // 16
            lw.lora = lo
            descs.append(self._pack_desc(lo))
            md = max(md, lw.N, lw.K)
        return md

    def _prep_double_lora(self, w, a, descs):
        dev = self.device
        md = 1
        # head-fragment weight images for the attention epilogues' rank-r projections (qfx_head_lora, ABI 6)
        hl = os.environ.get("QFX_FUSE_HEAD_LORA", "1") != "0" and self.config.attention_head_dim % 32 == 0
        for s, names in (("img", ("to_q", "to_k", "to_v")), ("txt", ("add_q_proj", "add_k_proj", "add_v_proj"))):
            md = max(md, self._prep_qkv_lora(w, s + ".", [getattr(a, n) for n in names], descs, hl="Bt" if hl else None))
        for key in ("img.o", "txt.o", "img.fc1", "img.fc2", "txt.fc1", "txt.fc2"):
            m = w[key].mod
            if isinstance(m, QfxLoraLinear):
                r = m.r[m.active_adapter]
                Rp, Kext = _ceil(r, 16), _ceil(3 * _ceil(r, 16), 64)
                lw = w[key]
                lo = self._make_lora(m, Rp, Kext, torch.zeros(Rp, lw.K, dtype=BF, device=dev),
                                     torch.zeros(Rp, lw.K, dtype=BF, device=dev),
                                     torch.zeros(lw.K, Kext, dtype=BF, device=dev), dev, hl="A" if (hl and key.endswith(".o")) else None)
                lw.lora = lo
                descs.append(self._pack_desc(lo))
                md = max(md, lw.N, lw.K)
        return md

    def _make_lora(self, m: QfxLoraLinear, Rp, Kext, A_hi, A_lo, WeT, dev, hl=None):
        lo = _LoraW()
        name = m.active_adapter
        lo.mod, lo.r, lo.Rp, lo.Kext, lo.scale = m, m.r[name], Rp, Kext, m.scaling[name]
        N, K = m.out_features, m.in_features
        lo.A_hi, lo.A_lo, lo.WeT = A_hi, A_lo, WeT
        lo.Bt_hi = torch.zeros(Rp, N, dtype=BF, device=dev)
        lo.Bt_lo = torch.zeros(Rp, N, dtype=BF, device=dev)
        lo.We = torch.zeros(N, Kext, dtype=BF, device=dev)
        lo.A_hl = torch.zeros(2 * Rp * K, dtype=BF, device=dev) if hl == "A" else None
        lo.Bt_hl = torch.zeros(2 * Rp * N, dtype=BF, device=dev) if hl == "Bt" else None
        lo.hl_dh = self.config.attention_head_dim
        lo.A_fr, lo.fr_row0, lo.fr_nf = None, 0, 0
        st = self._lora
        oa, ob = st.offset_of(m.A), st.offset_of(m.B)
        lo.gA = st.gflat[oa:oa + m.A.numel()]
        lo.gB = st.gflat[ob:ob + m.B.numel()]
        return lo

    @staticmethod
    def _pack_desc(lo: _LoraW):
        d = L.LoraPackArgs()
        m = lo.mod
        d.A, d.B, d.r, d.K, d.N = m.A.data_ptr(), m.B.data_ptr(), lo.r, m.in_features, m.out_features
        d.scale = 0.0 if m.merged else lo.scale      # merged adapter: the base weight already holds scale * B A
        d.A_hi, d.A_lo, d.ld_a = lo.A_hi.data_ptr(), lo.A_lo.data_ptr(), lo.A_hi.stride(0)
        d.Bt_hi, d.Bt_lo, d.ld_bt = lo.Bt_hi.data_ptr(), lo.Bt_lo.data_ptr(), lo.Bt_hi.stride(0)
        d.We, d.ld_we = lo.We.data_ptr(), lo.We.stride(0)
        d.WeT, d.ld_wet = lo.WeT.data_ptr(), lo.WeT.stride(0)
        d.Rp, d.Kext = lo.Rp, lo.Kext
        d.A_hl = lo.A_hl.data_ptr() if lo.A_hl is not None else None
        d.Bt_hl = lo.Bt_hl.data_ptr() if lo.Bt_hl is not None else None
        d.hl_dh = lo.hl_dh
        d.A_fr = lo.A_fr.data_ptr() if lo.A_fr is not None else None
        d.fr_row0, d.fr_nf = lo.fr_row0, lo.fr_nf
        return d

    def refresh_lora_operands(self):
        """Re-split the fp32 adapter weights into the bf16 hi/lo MFMA operands (one launch for all adapters)."""
        prep = self._prepare_lora()
        if prep["n"]:
            rc = lib.qfx_lora_pack(prep["descs"].data_ptr(), prep["n"], prep["max_dim"], torch.cuda.current_stream().cuda_stream)
            L.check(rc, "qfx_lora_pack")

    # ------------------------------------------------------------------ forward
    @contextlib.contextmanager
    def cache_context(self, name: str):
        """diffusers CacheMixin.cache_context("cond"|"uncond") (qwen_image_edit_trainer.py:1237,1255): no feature caches exist
        here, the context is accepted for call-site compatibility."""
        yield

    def forward(self, hidden_states, encoder_hidden_states=None, encoder_hidden_states_mask=None, timestep=None,
                img_shapes=None, txt_seq_lens=None, guidance=None, attention_kwargs=None, return_dict=True, attention_mask=None):
        """attention_mask: optional bool [B, T+S_max] padding mask of the multi-resolution model (transformer_qwen_custom.py:384-396);
        with it, or with per-sample img_shapes that differ, the masked / per-sample-RoPE launch program runs."""
        if encoder_hidden_states is None:
            raise ValueError("QwenImageTransformer2DModel requires encoder_hidden_states (text stream)")
        if guidance is not None:
            raise NotImplementedError("guidance embeddings are not part of the Qwen-Image-Edit training path")
        B, S_i, T = hidden_states.shape[0], hidden_states.shape[1], encoder_hidden_states.shape[1]
        batched = isinstance(img_shapes, list) and len(img_shapes) > 0 and isinstance(img_shapes[0], list)
        ragged = batched and not all(sh == img_shapes[0] for sh in img_shapes)
        if attention_mask is not None or ragged:
            if attention_mask is not None:
                if attention_mask.dim() != 2:
                    raise ValueError("attention_mask must have shape (batch, total_sequence_length).")
                if attention_mask.shape[1] < T + S_i:
                    raise ValueError(f"attention_mask length {attention_mask.shape[1]} is smaller than expected sequence length {T + S_i}.")
            plan = self.get_plan_multires(B, S_i, T, img_shapes, txt_seq_lens, attention_mask)
            out = _QwenDiTFn.apply(self, plan, hidden_states, encoder_hidden_states, timestep, *self.lora_parameters())
            return (out,) if not return_dict else _Cfg(sample=out)
        plan = self.get_plan(B, S_i, T, img_shapes, txt_seq_lens)
        out = _QwenDiTFn.apply(self, plan, hidden_states, encoder_hidden_states, timestep, *self.lora_parameters())
        if not return_dict:
            return (out,)
        return _Cfg(sample=out)

    def get_plan(self, B, S_i, T, img_shapes, txt_seq_lens):
        shapes = normalize_img_shapes(img_shapes)
        if sum(f * h * w for f, h, w in shapes) != S_i:
            raise ValueError(f"img_shapes {shapes} do not cover the {S_i} image tokens")
        if txt_seq_lens is not None and max(txt_seq_lens) != T:
            raise ValueError("max(txt_seq_lens) must equal the text sequence length (reference RoPE broadcast)")
        self._prepare()
        self._prepare_lora()
        key = (B, S_i, T, shapes, self._version)
        return self._plans.get_or_build(key, lambda: _QwenPlan(self, B, S_i, T, shapes))

    def get_plan_multires(self, B, S_i, T, img_shapes, txt_seq_lens, attention_mask):
        """One plan per LADDER size of the padded shape (B, ladder(S_max), T) in an LRU-bounded cache (plan_cache.py): the rows
        between S_max and the ladder size are further padded rows of the masked program.  The per-batch contents (per-sample
        RoPE, key mask, row masks) are refreshed on every call."""
        self._prepare()
        self._prepare_lora()
        S_plan = ladder(S_i)
        key = ("multires", B, S_plan, T, self._version)
        plan = self._plans.get_or_build(key, lambda: _QwenPlan(self, B, S_plan, T, None, multires=True))
        plan.set_multires(img_shapes, txt_seq_lens, attention_mask, S_in=S_i)
        return plan


# ----------------------------------------------------------------------------------------------
class _QwenPlan:
    """Launch programs (forward, backward) + persistent arena for one shape signature."""

    side_grads = False        # see _init_side_grads (plans that do not call it keep every launch on the main stream)
    _side_q = ()
    _side_late = False
    _pending_side = None
    _ncopy = 2                # copies of the side launches' scratch operands (see _init_side_grads)

    def __init__(self, model, B: int, S_i: int, T: int, shapes, multires: bool = False):
        self._setup(model, B, S_i, T)
        cfg = model.config
        D, S = self.D, self.S
        buf = self.buf
        rows = self.rows
        Lyr = cfg.num_layers
        Cin, Cout, Jd = cfg.in_channels, model.proj_out.out_features, cfg.joint_attention_dim
        P = model._prepared
        A = self.A
        self.multires = multires
        self.rm_txt0 = None
        if multires:
            # per-batch contents, filled by set_multires(): per-sample RoPE (identity on unrotated rows), additive key mask,
            # row masks of the padded image tokens (every block) and of the padded text tokens (once, after txt_in)
            A["rope_b"] = buf(B, S, self.dh // 2, 2, dtype=F32)
            A["kmask"] = buf(B, S, dtype=F32, zero=True)
            A["rm_img"] = buf(B * S_i, dtype=F32)
            A["rm_txt0"] = buf(B * T, dtype=F32)
            self.rope, self.rope_bs = A["rope_b"], S * (self.dh // 2) * 2
            self.kmask = A["kmask"]
            self.rmask = {"img": A["rm_img"], "txt": None, "joint": None}
            self.rm_txt0 = A["rm_txt0"]
        else:
            self.rope = qwen_joint_rope(shapes, T, cfg.axes_dims_rope).to(model.device)
            assert self.rope.shape == (S, self.dh // 2, 2)
        A["in_img"] = buf(B * S_i, Cin); A["in_txt"] = buf(B * T, Jd); A["t"] = buf(B, dtype=F32)
        A["tproj"] = buf(B, 256); A["t1"] = buf(1, B, D); A["temb"] = buf(1, B, D)
        A["txt_n"] = buf(B * T, Jd)
        A["X"] = {s: [buf(rows[s], D) for _ in range(Lyr + 1)] for s in ("img", "txt")}
        A["mods"] = buf(2 * Lyr, B, 6 * D); A["mod_out"] = buf(1, B, 2 * D)
        A["xn_out"] = buf(B * S_i, D); A["out"] = buf(B * S_i, Cout)
        A["dpred"] = buf(B * S_i, Cout); A["dxn"] = buf(B * S_i, D)
        A["blk"] = [self._alloc_double_block(w) for w in P["blocks"]]
        self._alloc_double_scratch(P["blocks"])
        self.cond = model.cond_lora
        if self.cond:
            from ..cond_hip import CondHeadHip
            A["dmods"] = buf(2 * Lyr, B, 6 * D, dtype=F32, zero=True)
            A["dmod_out"] = buf(1, B, 2 * D, dtype=F32, zero=True)
            te = model.time_text_embed.timestep_embedder
            self.cond_head = CondHeadHip(
                model, B, D, chains=[(te.linear_1, te.linear_2, A["tproj"], A["t1"], A["temb"])], temb=A["temb"],
                banks=[([m for blk in model.transformer_blocks for m in (blk.img_mod[1], blk.txt_mod[1])], A["mods"], A["dmods"]),
                       ([model.norm_out.linear], A["mod_out"], A["dmod_out"])], buf=buf)
            for bb in A["blk"]:    # pre-gate outputs of the two gated linears of a block (d gate = sum_rows dx_out * y)
                bb["y1"] = {s: buf(rows[s], D) for s in ("img", "txt")}
                bb["y2"] = {s: buf(rows[s], D) for s in ("img", "txt")}
        A["site"] = {"img_in": self._site_alloc(P["img_in"], rows["img"]), "txt_in": self._site_alloc(P["txt_in"], rows["txt"]),
                     "proj_out": self._site_alloc(P["proj_out"], rows["img"])}
        self.in_grad = P["img_in"].lora is not None or P["txt_in"].lora is not None   # the gradient must reach the block-0 inputs
        self.full_bwd = self.in_grad or self.cond      # block 0 runs its complete backward (d xm1 feeds d shift1 / d scale1)
        if multires and P["txt_in"].lora is not None:
            raise NotImplementedError("LoRA on txt_in with the multi-resolution Qwen model (the reference has no training caller for it)")
        self.fwd = _Prog()
        self.bwd = _Prog()
        self._init_side_grads(P["blocks"], type(self) is _QwenPlan)
        self._build_forward(P)
        self._build_backward(P)

    def _init_side_grads(self, blocks, allowed):
        """LoRA weight gradients are leaves of the backward: the batched lora_grad launch of block i goes to a low-priority side
        stream and overlaps the first two GEMMs of block i-1 (persistent 240-block grids leave 16 CUs idle); the main stream joins
        it right before block i-1 first overwrites one of its operands (dyg1).  With feed-forward adapters dh and their v^T scratch
        alternate by block parity too, the join sits at the top of the block (dh is the first thing a block's backward overwrites),
        and the launches that read dyg2 stay on the main stream.  QFX_SIDE_GRADS=0 keeps everything on the main stream."""
        import os
        ff = any(w[s + k].lora is not None for w in blocks for s in ("img", "txt") for k in (".fc1", ".fc2"))
        self._ff_side = bool(ff)      # feed-forward adapters: dh / v^T(fc1, fc2) join the parity-alternated scratch, the join moves to the block top
        self.side_grads = bool(allowed and self.has_lora and os.environ.get("QFX_SIDE_GRADS", "1") != "0"
                               and (not ff or os.environ.get("QFX_SIDE_GRADS_FF", "1") != "0"))
        self._side_q = []              # (event, prefix) of the blocks whose gradient launches are in flight on the side stream
        # QFX_SIDE_COPIES=3 (round-6 lever): three copies of the side launches' scratch operands instead of two, so that the join with the
        # launch of block i+2 can sit right in front of the fork of block i (two adjacent barrier packets instead of two separate bubbles)
        self._ncopy = 3 if (self.side_grads and os.environ.get("QFX_SIDE_COPIES", "2") == "3") else 2
        # QFX_SIDE_AT_ATTN=1 (round-6 lever): the gradient launches of block i go out in front of block i-1's ATTENTION backward (whose
        # launches leave CUs idle in their last round) instead of in front of its feed-forward GEMMs (which they slow down by 18 us)
        self._side_late = bool(self.side_grads and self._ncopy == 2 and os.environ.get("QFX_SIDE_AT_ATTN", "0") == "1")
        self._pending_side = None
        if self.side_grads:
            dev = self.model.device
            from .. import ops
            # QFX_SIDE_CUS=16 confines the side stream to two CUs per XCD (qfx_stream_create_cu_masked).  Measured: the mask works
            # (tools/cu_mask_probe.py) but the whole step slows from 99.7 to 127.9 ms with such a queue alive -> default: no mask
            self.bwd.side = ops.side_stream(dev, int(os.environ.get("QFX_SIDE_CUS", "0")))
            self._ev_fork = ops.Event()
            # the scratch operands of those launches (dyg1, dqkv, v^T) alternate between two copies by block parity, so a launch
            # has a whole block of main-stream work to hide under (on the 16 idle CUs it runs ~5x longer than alone)
            A = self.A
            for name in ("dyg1", "dqkv", "Vt", "VtO") + (("dh", "VtF1", "VtF2") if ff else ()):
                src = A[name]
                for c in range(1, self._ncopy):
                    if isinstance(src, dict):
                        A[name + f"#{c}"] = {s: (tuple(torch.zeros_like(t) for t in v) if isinstance(v, tuple) else torch.zeros_like(v))
                                             for s, v in src.items()}
                    else:
                        A[name + f"#{c}"] = torch.zeros_like(src)

    def _fuse_qk_bwd(self, a, sqk, norms, norm_flags, eps):
        """Backward of the QK RMSNorm + RoPE in the epilogues of qfx_attn_bwd_dq / _dkv (one pass over dqkv and a launch less per
        block; QFX_FUSE_QKNORM_BWD=0 keeps the separate qfx_qk_norm_rope_bwd launch)."""
        if os.environ.get("QFX_FUSE_QKNORM_BWD", "1") == "0":
            return
        nq_t, nk_t, nq_i, nk_i = norms
        a.qk_saved, a.ld_saved = _ptr(sqk), 2 * self.D
        a.rope, a.rope_bstride = _ptr(self.rope), self.rope_bs
        a.wq_txt, a.wk_txt, a.wq_img, a.wk_img = _ptr(nq_t), _ptr(nk_t), _ptr(nq_i), _ptr(nk_i)
        a.T, a.norm_flags, a.norm_eps = self.T, norm_flags & 1, eps

    def _head_lora_slots(self, a, w, live):
        """Fill the qfx_head_lora slots of a block's attention arguments (forward slot 0: out-projection adapter; backward slots
        1-3: q / k / v adapters, which need the fused QK-norm backward).  Returns {stream: H} for the streams whose out-projection
        down projection now rides in qfx_attn_fwd; the backward streams are recorded in a._hl_qkv."""
        a._hl_qkv = {}
        # the fused projections pick the text / image adapter by `row >= a.T`: set it HERE, not only in _fuse_qk_bwd (which returns
        # early under QFX_FUSE_QKNORM_BWD=0 and used to leave T = 0: every text row then took the image adapter -- ADVICE r4)
        a.T = self.T
        if not getattr(self, "head_lora", False):
            return {}
        A, B, S, H = self.A, self.B, self.S, self.H
        out = {}
        los = {s: w[s + ".o"].lora for s in live}
        rps = {lo.Rp for lo in los.values() if lo is not None}
        if len(rps) == 1:
            Rp = rps.pop()
            hl = a.hl[0]
            hl.part, hl.part_hstride, hl.ld_part, hl.c0, hl.R = _ptr(A["hl_o"]), B * S * A["hl_o"].shape[2], A["hl_o"].shape[2], 0, Rp
            for si, s in enumerate(("img", "txt")):
                lo = los.get(s)
                if lo is not None and lo.A_hl is not None:
                    hl.w_pk[si] = _ptr(lo.A_hl)
                    out[s] = H
        if a.qk_saved:
            grp = {s: w[s + ".qkv_lora"] for s in ("img", "txt")}
            full = {s: g for s, g in grp.items() if g is not None and all(
                w[s + ".qkv"][sec].lora is not None and w[s + ".qkv"][sec].lora.Bt_hl is not None for sec in range(3))}
            rps = {g["Rp"] for g in full.values()}
            if len(rps) == 1:
                Rp = rps.pop()
                ld = A["hl_qkv"].shape[2]
                for sec in range(3):
                    hl = a.hl[1 + sec]
                    hl.part, hl.part_hstride, hl.ld_part, hl.c0, hl.R = _ptr(A["hl_qkv"]), B * S * ld, ld, sec * Rp, Rp
                    for si, s in enumerate(("img", "txt")):
                        if s in full:
                            lo = w[s + ".qkv"][sec].lora
                            hl.w_pk[si] = _ptr(lo.Bt_hl)
                a._hl_qkv = {s: Rp for s in full}
        return out

    def _head_reduce_args(self, part, H, R, M, rpb, off, ext, Ut, group_R, group_stride):
        r = L.LoraHeadReduceArgs()
        r.part, r.part_hstride, r.ld_part, r.H = _ptr(part), part.shape[1] * part.shape[2], part.shape[2], H
        r.M, r.R = M, R
        r.ext, r.ld_ext = _ptr(ext), ext.stride(0)
        r.Ut_hi, r.Ut_lo, r.ld_ut = _ptr(Ut[0]), _ptr(Ut[1]), Ut[0].stride(0)
        r.group_R, r.group_stride = group_R, group_stride
        r.rows_per_batch, r.x_batch_rows, r.x_row_off = rpb, self.S, off
        return r

    @staticmethod
    def _flush_head_reduce(prog, pending):
        if pending:
            arr = (L.LoraHeadReduceArgs * len(pending))(*pending)
            prog.keep.append(arr)
            prog.c(lib.qfx_lora_head_reduce, arr, len(pending))
            pending.clear()

    def _sb(self, name, par):
        """Scratch buffer `name` of block parity `par` (second copies exist only with side-stream gradient launches)."""
        return self.A[name + f"#{par}"] if (par and self.side_grads) else self.A.get(name)   # v^T scratch exists only with adapters

    def _side_fork(self):
        self._ev_fork.record(torch.cuda.current_stream())
        self._ev_fork.wait(self.bwd.side)

    def _side_join(self, p, keep=0):
        """Emit the joins with the oldest in-flight side-stream gradient launches until at most `keep` stay in flight, each
        followed by the mark that its block's gradients are final."""
        while len(self._side_q) > keep:
            ev, prefix = self._side_q.pop(0)
            p.py(lambda ev=ev: ev.wait(torch.cuda.current_stream()))
            if prefix is not None:
                p.mark(prefix)

    def _setup(self, model, B, S_i, T):
        self.model = model
        cfg = model.config
        dev = model.device
        self.B, self.S_i, self.T = B, S_i, T
        self.H, self.dh = cfg.num_attention_heads, cfg.attention_head_dim
        self.D = self.H * self.dh
        self.S = T + S_i
        self.S_pad = _ceil(self.S, 64)
        if B > 8:
            raise NotImplementedError("per-GPU batch > 8 (modulation GEMV holds <= 8 rows in LDS)")

        def buf(*shape, dtype=BF, zero=False):
            return (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=dev)

        self.buf = buf
        self.rows = {"img": B * S_i, "txt": B * T}
        self.rpb = {"img": S_i, "txt": T}
        self.off = {"img": T, "txt": 0}
        self.A = {}
        self.rope_bs = 0          # per-sample RoPE stride (multi-resolution plans set it)
        self.rmask = {"img": None, "txt": None, "joint": None}   # fp32 row masks of padded tokens (multi-resolution)
        self.kmask = None         # additive fp32 key mask [B,S] (multi-resolution)

    def _alloc_double_block(self, w):
        """Per-block saved activations of a double-stream block."""
        buf, rows, B, S, D, H, S_pad = self.buf, self.rows, self.B, self.S, self.D, self.H, self.S_pad
        b = dict(qkv=buf(B, S, 3 * D), sqk=buf(B, S, 2 * D), ao=buf(B, S, D), lse=buf(B, H, S_pad, dtype=F32, zero=True),
                 x1={s: buf(rows[s], D) for s in ("img", "txt")}, h={s: buf(rows[s], 4 * D) for s in ("img", "txt")})
        for s in ("img", "txt"):
            grp = w[s + ".qkv_lora"]
            mp = _ceil(rows[s], 128)
            if grp is not None:
                b["xm1." + s] = buf(rows[s], D)
                b["Uqkv." + s] = (buf(3 * grp["Rp"], mp, zero=True), buf(3 * grp["Rp"], mp, zero=True))   # u^T hi/lo
            if w[s + ".o"].lora is not None:
                rp_o = w[s + ".o"].lora.Rp
                b["Uo." + s] = (buf(rp_o, mp, zero=True), buf(rp_o, mp, zero=True))
            # feed-forward adapters: their inputs (LN2 output / GELU output) are scratch otherwise and must be kept for dA
            if w[s + ".fc1"].lora is not None:
                rp = w[s + ".fc1"].lora.Rp
                b["xm2." + s] = buf(rows[s], D)
                b["Uf1." + s] = (buf(rp, mp, zero=True), buf(rp, mp, zero=True))
            if w[s + ".fc2"].lora is not None:
                rp = w[s + ".fc2"].lora.Rp
                b["g." + s] = buf(rows[s], 4 * D)
                b["Uf2." + s] = (buf(rp, mp, zero=True), buf(rp, mp, zero=True))
        return b

    def _alloc_double_scratch(self, blocks):
        """Scratch shared by all double-stream blocks (forward + backward)."""
        buf, rows, B, S, D, H, dh, S_pad, A = self.buf, self.rows, self.B, self.S, self.D, self.H, self.dh, self.S_pad, self.A
        A["xm"] = {s: buf(rows[s], D) for s in ("img", "txt")}
        A["g"] = {s: buf(rows[s], 4 * D) for s in ("img", "txt")}
        kext_max = 0
        rp_max = 0
        for w in blocks:   # LoRA scratch (pad columns stay zero forever)
            for s in ("img", "txt"):
                if w[s + ".qkv_lora"] is not None:
                    kext_max = max(kext_max, w[s + ".qkv_lora"]["Kext"]); rp_max = max(rp_max, w[s + ".qkv_lora"]["Rp"])
                for key in (".o", ".fc1", ".fc2"):
                    if w[s + key].lora is not None:
                        kext_max = max(kext_max, w[s + key].lora.Kext); rp_max = max(rp_max, w[s + key].lora.Rp)
        self.has_lora = kext_max > 0
        if self.has_lora:
            A["ext3"] = {s: buf(rows[s], 3 * kext_max, zero=True) for s in ("img", "txt")}
            A["ext1"] = {s: buf(rows[s], kext_max, zero=True) for s in ("img", "txt")}
            for name in ("VtO", "VtF1", "VtF2"):   # one v^T scratch per adapter site: the deferred dA launches read them at block end
                A[name] = {s: (buf(rp_max, _ceil(rows[s], 128), zero=True), buf(rp_max, _ceil(rows[s], 128), zero=True))
                           for s in ("img", "txt")}
            A["Vt"] = {s: (buf(3 * rp_max, _ceil(rows[s], 128), zero=True), buf(3 * rp_max, _ceil(rows[s], 128), zero=True))
                       for s in ("img", "txt")}   # v^T hi/lo scratch (pad columns stay zero)
        # Round 4 (ABI 6): the rank-r down projections whose input an attention kernel holds in registers -- u = ao A_o^T in the
        # forward, v = d(pre-norm q | k) , dV times (sB)^T in the backward -- ride in that kernel's epilogue as per-head partial sums
        # (qfx_head_lora) and a small reduce launch writes what qfx_lora_down wrote: 2 of the 3 qfx_lora_down launches per block go.
        # Needs T % 16 == 0 (a 16-row fragment is all text or all image), rank <= 32, the bf16 trunk (the MX-FP8 trunk's down
        # projections also quantise their input), and -- backward -- the QK-norm backward fused into the same epilogues.
        self.head_lora = (self.has_lora and os.environ.get("QFX_FUSE_HEAD_LORA", "1") != "0" and self.T % 16 == 0 and 0 < rp_max <= 32
                          and getattr(self.model, "_quant", None) is None)
        if self.head_lora:
            A["hl_o"] = buf(H, B * S, rp_max, dtype=F32, zero=True)
            A["hl_qkv"] = buf(H, B * S, 3 * rp_max, dtype=F32, zero=True)
        A["dX"] = {s: [buf(rows[s], D), buf(rows[s], D)] for s in ("img", "txt")}
        A["dyg2"] = {s: buf(rows[s], D) for s in ("img", "txt")}
        A["dyg1"] = {s: buf(rows[s], D) for s in ("img", "txt")}
        A["dx1"] = {s: buf(rows[s], D) for s in ("img", "txt")}
        A["dh"] = {s: buf(rows[s], 4 * D) for s in ("img", "txt")}
        A["dxm"] = {s: buf(rows[s], D) for s in ("img", "txt")}
        A["dao"] = buf(B, S, D, zero=True)
        A["dsum"] = buf(B, H, S_pad, dtype=F32, zero=True)
        A["dqkv"] = buf(B, S, 3 * D)

    # ------------------------------------------------------------------ emit helpers
    @staticmethod
    def _gargs(*, A1, lda1, B1, K1, M, N, C_, ldc, ldb1=None, bias=None, A2=None, lda2=0, B2=None, ldb2=0, K2=0,
               epi=L.EPI_NONE, C2=None, ldc2=0, aux=None, ldaux=0, gate=None, gate_bs=0, rpb=None, a_map=(0, 0), c_map=(0, 0),
               seg2_plain=0, aux_unmapped=0, row_mask=None):
        g = L.GemmArgs()
        g.A1, g.B1, g.lda1, g.ldb1, g.K1 = _ptr(A1), _ptr(B1), lda1, (K1 if ldb1 is None else ldb1), K1
        if K2:
            g.A2, g.B2, g.lda2, g.ldb2, g.K2 = _ptr(A2), _ptr(B2), lda2, ldb2, K2
        g.M, g.N = M, N
        g.bias = _ptr(bias)
        g.C, g.ldc = _ptr(C_), ldc
        g.C2, g.ldc2 = _ptr(C2), ldc2
        g.aux, g.ldaux = _ptr(aux), ldaux
        g.gate, g.gate_bstride = _ptr(gate), gate_bs
        g.rows_per_batch = M if rpb is None else rpb
        g.a_batch_rows, g.a_row_off = a_map
        g.c_batch_rows, g.c_row_off = c_map
        g.epi = epi
        g.seg2_plain = seg2_plain
        g.aux_unmapped = aux_unmapped
        g.row_mask = _ptr(row_mask)
        g._src = (A1, lda1, a_map, g.rows_per_batch, B1)     # python-side operand identities (low-precision forward re-targets them)
        return g

    def _gemm(self, prog, **kw):
        self._gemm_group(prog, [self._gargs(**kw)])

    def _preq_out(self, prog, out, ld, M, N, tag, keep_bf16=True, a_map=(0, 0)):
        """Scratch (fp8 bytes [M, N], tile-major scales [N/128, M, 4]) for the MX-FP8 image of `out` that its PRODUCER writes on
        the fly; registered so that the MX-FP8 GEMM group that comes next skips its quantisation pass for this operand (the
        registration lives until that next group).  None when the trunk is not quantised in this direction / shape."""
        q = getattr(self.model, "_quant", None)
        if (not q or (prog is self.bwd and q != "mxfp8-fb") or N % 128 or N < 1024 or out is None
                or os.environ.get("QFX_FP8_FUSED_QUANT", "1") == "0"):
            return None
        scratch = self.__dict__.setdefault("_q8", {})
        slot = ("pre", M, N, tag)
        if slot not in scratch:
            scratch[slot] = (self.buf(M, N, dtype=torch.uint8), self.buf(N // 128, M, 4, dtype=torch.uint8))
        oq, osc = scratch[slot]
        self.__dict__.setdefault("_preq", {})[(out.data_ptr(), ld, tuple(a_map), M)] = (oq, osc, not keep_bf16)
        return oq, osc

    def _gemm_group(self, prog, groups):
        """One grid for several independent GEMMs with the same epilogue (image+text streams, q/k/v)."""
        try:
            return self._gemm_group_impl(prog, groups)
        finally:
            pq = self.__dict__.setdefault("_preq", {})
            pq.clear()      # on-the-fly quantised operands are for the GEMM group that follows their producer
            pq.update(self.__dict__.pop("_preq_next", None) or [])

    def _gemm_group_impl(self, prog, groups):
        q = getattr(self.model, "_quant", None)
        if q and (prog is self.fwd or (q == "mxfp8-fb" and prog is self.bwd)) and all(self._fp8_ok(g) for g in groups):
            return self._gemm_group_mxfp8(prog, groups)
        for g in groups:      # an operand that left its producer as MX-FP8 only (no bf16 copy) must not reach a bf16 GEMM
            A1, lda1, a_map, rpb, B1 = g._src
            ent = self.__dict__.get("_preq", {}).get((A1.data_ptr(), lda1, a_map, g.M)) if isinstance(A1, torch.Tensor) else None
            if ent is not None and ent[2]:
                raise RuntimeError("internal: an operand that exists only as MX-FP8 is consumed by a bf16 GEMM")
        if len(groups) == 1:
            prog.keep.append(groups[0])
            prog.c(lib.qfx_gemm_bf16, C.byref(groups[0]))
            return
        arr = (L.GemmArgs * len(groups))(*groups)
        prog.keep.append(arr)
        prog.c(lib.qfx_gemm_grouped, arr, len(groups))

    # ------------------------------------------------------------------ low-precision trunk (MX-FP8 forward GEMMs)
    @staticmethod
    def _fp8_ok(g):
        A1, lda1, a_map, rpb, B1 = g._src
        # a weight [N, K1] or the first N rows of a taller contiguous one (the FLUX single block's proj_out^T is contracted in two row
        # ranges): MX blocks run along K, so a row range quantises to the same bytes as the rows of the whole matrix
        return (isinstance(B1, torch.Tensor) and B1.dim() == 2 and B1.is_contiguous() and g.K1 % 128 == 0 and g.K1 >= 1024
                and g.N >= 1024 and B1.shape[1] == g.K1 and B1.shape[0] >= g.N and g.ldb1 == g.K1 and not g.seg2_plain)

    def _gemm_group_mxfp8(self, prog, groups):
        """Forward GEMMs of the block linears on the block-scaled FP8 MFMA (model.quantize = "mxfp8", the MI355X analogue of
        the reference's quantized trunk, src/qflux/models/quantize.py): the frozen weight is quantised ONCE (cached on the model),
        the activation operand once per distinct input of the group; bias, bf16 mid-rounding, the bf16 LoRA K-extension and the
        epilogue are unchanged.  The backward stays on the bf16 operands (dX = dY W in bf16, adapters on the bf16 activations)."""
        from .. import ops
        cache = self.model.__dict__.setdefault("_wq_cache", {})
        scratch = self.__dict__.setdefault("_q8", {})
        preq = self.__dict__.setdefault("_preq", {})      # operands that left their producer's epilogue already quantised
        quantised = {}
        fp8 = []
        produced = []       # operands this group's epilogues quantise for the NEXT group (registered after this one is emitted)
        tiles = sum(((g.M + 255) // 256) * ((g.N + 127) // 128) for g in groups)
        persistent = tiles >= 160 and len(groups) <= 6
        for gi_, g in enumerate(groups):
            A1, lda1, a_map, rpb, B1 = g._src
            key = (B1.data_ptr(), (g.N, g.K1))
            if key not in cache:
                cache[key] = ops.quant_mxfp8(B1[:g.N])
            wq, ws = cache[key]
            akey = (A1.data_ptr(), lda1, a_map, g.M)
            if akey in preq:
                quantised[akey] = preq[akey][:2]
            if akey not in quantised:
                slot = (g.M, g.K1, len(quantised))
                if slot not in scratch:
                    scratch[slot] = (self.buf(g.M, g.K1, dtype=torch.uint8), self.buf(g.K1 // 128, g.M, 4, dtype=torch.uint8))
                xq, xs = scratch[slot]
                qa = L.QuantArgs()
                qa.X, qa.ldx, qa.M, qa.K = _ptr(A1), lda1, g.M, g.K1
                qa.Q, qa.ldq, qa.S, qa.lds = _ptr(xq), g.K1, _ptr(xs), 0
                qa.rows_per_batch, qa.x_batch_rows, qa.x_row_off = rpb, a_map[0], a_map[1]
                prog.keep.append(qa)
                prog.c(lib.qfx_quant_mxfp8, C.byref(qa))
                quantised[akey] = (xq, xs)
            xq, xs = quantised[akey]
            f = L.GemmFp8Args()
            C.memmove(C.byref(f.g), C.byref(g), C.sizeof(L.GemmArgs))
            f.g.A1, f.g.lda1, f.g.a_batch_rows, f.g.a_row_off = _ptr(xq), g.K1, 0, 0
            f.g.B1, f.g.ldb1 = _ptr(wq), g.K1
            f.sa, f.ldsa, f.sb, f.ldsb = _ptr(xs), 0, _ptr(ws), 0
            # quantising epilogue: gelu(h) (forward fc1 -> fc2) / dh (backward fc2-dX -> fc1-dX) leave the producer as MX-FP8 when the
            # consumer is an MX-FP8 GEMM too; the stand-alone quantisation pass of that operand (60 MB read per block) disappears,
            # and so does the bf16 copy when nothing else reads it (`nxt` = (tensor, row stride, bf16 copy still needed))
            nxt = getattr(g, "_next", None)
            if (nxt is not None and persistent and g.N % 128 == 0 and g.N >= 1024 and g.c_batch_rows == 0
                    and os.environ.get("QFX_FP8_FUSED_QUANT", "1") != "0"):
                out, ld_out, keep_bf16 = nxt
                slot = ("pre", g.M, g.N, gi_)
                if slot not in scratch:
                    scratch[slot] = (self.buf(g.M, g.N, dtype=torch.uint8), self.buf(g.N // 128, g.M, 4, dtype=torch.uint8))
                oq, osc = scratch[slot]
                f.cq, f.cs, f.ldcq, f.cq_rows, f.cq_only = _ptr(oq), _ptr(osc), g.N, g.M, 0 if keep_bf16 else 1
                produced.append(((out.data_ptr(), ld_out, (0, 0), g.M), (oq, osc, not keep_bf16)))
            fp8.append(f)
            prog.keep.append((wq, ws))
        # one persistent grid for the whole group (image + text stream, q/k/v) when it is large enough, else one launch each
        if persistent:
            arr = (L.GemmFp8Args * len(fp8))(*fp8)
            prog.keep.append(arr)
            prog.c(lib.qfx_gemm_mxfp8_grouped, arr, len(fp8))
        else:
            for f in fp8:
                prog.keep.append(f)
                prog.c(lib.qfx_gemm_mxfp8, C.byref(f))
        self._preq_next = produced

    def _gemm_mxfp8_cat(self, prog, parts, *, M, N, C_, ldc, ext=None):
        """C = sum_i X_i W_i^T (+ the bf16 LoRA K-extension `ext` = (A2, lda2, B2, ldb2, K2)) as ONE MX-FP8 contraction over the
        concatenated K of `parts` = [(X_i [M, K_i] bf16, row stride, K_i, W_i^T as [N, >= K_i] bf16)]: the operands are quantised
        side by side into one byte buffer / one tile-major scale array (K_i % 128 == 0: MX blocks and scale tiles never straddle a
        seam), the weights once (cached on the model).  Used for dX contractions that sum several frozen linears ("mxfp8-fb")."""
        from .. import ops
        Kt = sum(K for _, _, K, _ in parts)
        cache = self.model.__dict__.setdefault("_wq_cache", {})
        key = ("cat", N) + tuple((Wt.data_ptr(), K) for _, _, K, Wt in parts)
        if key not in cache:
            cache[key] = ops.quant_mxfp8(torch.cat([Wt[:N, :K] for _, _, K, Wt in parts], dim=1).contiguous())
        wq, ws = cache[key]
        scratch = self.__dict__.setdefault("_q8", {})
        slot = ("cat", M, Kt)
        if slot not in scratch:
            scratch[slot] = (self.buf(M, Kt, dtype=torch.uint8), self.buf(Kt // 128, M, 4, dtype=torch.uint8))
        xq, xs = scratch[slot]
        col = 0
        for X, ldx, K, _ in parts:
            assert K % 128 == 0
            qa = L.QuantArgs()
            qa.X, qa.ldx, qa.M, qa.K = _ptr(X), ldx, M, K
            qa.Q, qa.ldq, qa.S, qa.lds = xq.data_ptr() + col, Kt, xs.data_ptr() + (col // 128) * M * 4, 0
            qa.rows_per_batch, qa.x_batch_rows, qa.x_row_off = M, 0, 0
            prog.keep.append(qa)
            prog.c(lib.qfx_quant_mxfp8, C.byref(qa))
            col += K
        kw = {}
        if ext is not None and ext[4] > 0:
            kw = dict(A2=ext[0], lda2=ext[1], B2=ext[2], ldb2=ext[3], K2=ext[4])
        g = self._gargs(A1=xq, lda1=Kt, B1=wq, K1=Kt, M=M, N=N, C_=C_, ldc=ldc, **kw)
        f = L.GemmFp8Args()
        C.memmove(C.byref(f.g), C.byref(g), C.sizeof(L.GemmArgs))
        f.sa, f.ldsa, f.sb, f.ldsb = _ptr(xs), 0, _ptr(ws), 0
        prog.keep.append((f, wq, ws))
        prog.c(lib.qfx_gemm_mxfp8, C.byref(f))

    def _down(self, prog, *, X, ldx, M, K, W_hi, W_lo, ldw, R, U=None, ldu=0, ext=None, ld_ext=0, Ut=None, group_R=None,
              group_stride=0, rpb=None, x_map=(0, 0), defer=None, xq=None):
        a = L.LoraDownArgs()
        if xq is not None:      # (bytes, scales, row stride, scale rows, first 32-column block): X's MX-FP8 image rides along
            a.xq, a.xs, a.ldxq, a.xs_rows, a.xq_kb0 = xq
        a.X, a.ldx, a.M, a.K = _ptr(X), ldx, M, K
        a.W_hi, a.W_lo, a.ldw, a.R = _ptr(W_hi), _ptr(W_lo), ldw, R
        a.U, a.ldu = _ptr(U), ldu
        a.ext, a.ld_ext = _ptr(ext), ld_ext
        if Ut is not None:
            a.Ut_hi, a.Ut_lo, a.ld_ut = _ptr(Ut[0]), _ptr(Ut[1]), Ut[0].stride(0)
        a.group_R = R if group_R is None else group_R
        a.group_stride = group_stride
        a.rows_per_batch = M if rpb is None else rpb
        a.x_batch_rows, a.x_row_off = x_map
        if defer is not None:
            defer.append(a)
            return
        prog.keep.append(a)
        prog.c(lib.qfx_lora_down, C.byref(a))

    @staticmethod
    def _ln_fwd_args(x, shift, scale, mod_bs, y, rows, D, rpb, eps):
        a = L.LnFwdArgs()
        a.x, a.shift, a.scale, a.mod_bstride, a.y = _ptr(x), _ptr(shift), _ptr(scale), mod_bs, _ptr(y)
        a.rows, a.D, a.rows_per_batch, a.eps = rows, D, rpb, eps
        return a

    @staticmethod
    def _ln_bwd_args(dy, x, scale, mod_bs, dres, gate, gate_bs, dx, dyg, rows, D, rpb, eps, row_mask):
        a = L.LnBwdArgs()
        a.dy, a.x, a.scale, a.mod_bstride = _ptr(dy), _ptr(x), _ptr(scale), mod_bs
        a.dres, a.gate, a.gate_bstride, a.dx, a.dyg = _ptr(dres), _ptr(gate), gate_bs, _ptr(dx), _ptr(dyg)
        a.row_mask, a.rows, a.D, a.rows_per_batch, a.eps = _ptr(row_mask), rows, D, rpb, eps
        return a

    def _ln_down(self, prog, entries):
        """LayerNorm+modulate of up to two streams with the LoRA down projection of the adapted ones fused in (qfx_ln_down_fwd).
        entries: [(LnFwdArgs, None | dict(W_hi, W_lo, ldw, R, Ut, ext, ld_ext, group_R, group_stride))].  Returns False when the
        shape is outside the fused kernel's range (caller falls back to the two separate launches)."""
        import os
        D = entries[0][0].D
        rs = {d["R"] for _, d in entries if d is not None}
        # (streams without adapters take the same kernel as plain LayerNorm rows, so that a zero adapter reproduces the frozen
        # model bit for bit: one LayerNorm arithmetic per site, adapted or not)
        if os.environ.get("QFX_FUSE_LN_DOWN", "1") == "0" or len(rs) > 1 or (rs and max(rs) > 48) or D % 256 or D > 3072 or len(entries) > 2:
            return False
        arr = (L.LnDownArgs * len(entries))()
        for i, (ln, d) in enumerate(entries):
            C.memmove(C.byref(arr[i].ln), C.byref(ln), C.sizeof(L.LnFwdArgs))
            if d is not None:
                a = arr[i]
                a.W_hi, a.W_lo, a.ldw, a.R = _ptr(d["W_hi"]), _ptr(d["W_lo"]), d["ldw"], d["R"]
                a.W_fr = _ptr(d.get("W_fr"))
                a.ext, a.ld_ext = _ptr(d["ext"]), d["ld_ext"]
                a.Ut_hi, a.Ut_lo, a.ld_ut = _ptr(d["Ut"][0]), _ptr(d["Ut"][1]), d["Ut"][0].stride(0)
                a.group_R, a.group_stride = d.get("group_R", d["R"]), d.get("group_stride", 0)
        prog.keep.append(arr)
        prog.c(lib.qfx_ln_down_fwd, arr, len(entries))
        return True

    @staticmethod
    def _flush_ln(prog, pending, struct, fn):
        """One launch for the LayerNorm problems of both streams (ragged row counts go last: only the last problem of a batch
        may have rows % 4 != 0)."""
        pend = sorted(pending, key=lambda a: (a.rows % 4 != 0))
        while pend:
            chunk = []
            while pend and len(chunk) < L.MAX_LN_BATCH:
                chunk.append(pend.pop(0))
                if chunk[-1].rows % 4:
                    break
            arr = (struct * len(chunk))(*chunk)
            prog.keep.append(arr)
            prog.c(fn, arr, len(chunk))
        pending.clear()

    @staticmethod
    def _flush_batch(prog, pending, struct, fn, side=False):
        """Emit deferred skinny-kernel problems as batched launches: same R per launch, at most QFX_MAX_BATCH each."""
        by_r = {}
        for a in pending:
            by_r.setdefault(a.R, []).append(a)
        order = sorted(by_r, reverse=True) if (side and os.environ.get("QFX_SIDE_ORDER", "0") == "wide_first") else list(by_r)
        for r_ in order:      # (QFX_SIDE_ORDER=wide_first, round-6 lever: the widest rank class of the side-stream launches first)
            lst = by_r[r_]
            for i in range(0, len(lst), L.MAX_BATCH):
                chunk = lst[i:i + L.MAX_BATCH]
                arr = (struct * len(chunk))(*chunk)
                prog.keep.append(arr)
                (prog.c_side if side else prog.c)(fn, arr, len(chunk))
        pending.clear()

    def _grad(self, prog, *, Vt, R, r_valid, X, ldx, M, K, G, g_sr, g_sc, group_R=None, rpb=None, x_map=(0, 0), out_scale=1.0,
              defer=None):
        a = L.LoraGradArgs()
        Gs = G if isinstance(G, (tuple, list)) else (G,)
        a.Vt_hi, a.Vt_lo, a.ldvt, a.R, a.r_valid = _ptr(Vt[0]), _ptr(Vt[1]), Vt[0].stride(0), R, r_valid
        a.group_R = R // len(Gs) if group_R is None else group_R
        a.X, a.ldx, a.M, a.K = _ptr(X), ldx, M, K
        a.G = _ptr(Gs[0])
        a.G1 = _ptr(Gs[1]) if len(Gs) > 1 else None
        a.G2 = _ptr(Gs[2]) if len(Gs) > 2 else None
        a.g_sr, a.g_sc = g_sr, g_sc
        a.rows_per_batch = M if rpb is None else rpb
        a.x_batch_rows, a.x_row_off = x_map
        a.out_scale = out_scale
        # ABI 7: chunk partials through a scratch of the problem's own, added up in chunk order by the last block to arrive -- the flat
        # LoRA gradient is bit-reproducible (QFX_GRAD_DET=0: the fp32 atomics of rounds 1-5).  One scratch per problem of the plan: launches
        # on the main and the side stream may overlap, 288 GB make sharing pointless (~10 MB per block).
        if os.environ.get("QFX_GRAD_DET", "1") != "0":
            nfl = int(lib.qfx_lora_grad_ws_floats(M, K, R))
            ws = torch.empty(max(nfl, 4), dtype=F32, device=X.device)
            cnt = torch.zeros((K + 127) // 128, dtype=torch.int32, device=X.device)
            a.ws, a.ws_count, a.ws_floats = _ptr(ws), _ptr(cnt), ws.numel()
            prog.keep.append((ws, cnt))
        if defer is not None:
            defer.append(a)
            return
        prog.keep.append(a)
        prog.c(lib.qfx_lora_grad, C.byref(a))

    def _mod_grad(self, prog, *, dy, x, rows, rpb, dshift, dscale, out_bs, dgate=None, dxo=None, y=None, row_mask=None, ld=None,
                  defer=None):
        """defer: list collecting the problem for ONE batched launch (_flush_mod_grad) -- the image and text stream of a block."""
        a = L.ModGradArgs()
        D = self.D
        ld = D if ld is None else ld
        a.dy, a.ld_dy, a.x, a.ld_x = _ptr(dy), ld, _ptr(x), ld
        a.dxo, a.ld_dxo, a.y, a.ld_y = _ptr(dxo), ld, _ptr(y), ld
        a.dshift, a.dscale, a.dgate, a.out_bstride = _ptr(dshift), _ptr(dscale), _ptr(dgate), out_bs
        a.row_mask, a.rows, a.D, a.rows_per_batch, a.eps = _ptr(row_mask), rows, D, rpb, 1e-6
        if defer is not None:
            defer.append(a)
            return
        prog.keep.append(a)
        prog.c(lib.qfx_mod_grad, C.byref(a))

    @staticmethod
    def _flush_mod_grad(prog, pending):
        for i in range(0, len(pending), L.MAX_LN_BATCH):
            chunk = pending[i:i + L.MAX_LN_BATCH]
            arr = (L.ModGradArgs * len(chunk))(*chunk)
            prog.keep.append(arr)
            prog.c(lib.qfx_mod_grad_batch, arr, len(chunk))
        pending.clear()

    # ------------------------------------------------------------------ stand-alone adapted linears (embedders, output projection ...)
    def _site_alloc(self, lw, M):
        """Private rank-r buffers of one adapted linear with M input rows: K-extension images of the forward (ext) and backward
        (extb) GEMM, transposed hi/lo splits of u = x A^T (kept for dB) and v = dy (sB)^T (for dA)."""
        if lw.lora is None:
            return None
        lo, buf, mp = lw.lora, self.buf, _ceil(M, 128)
        return dict(ext=buf(M, lo.Kext, zero=True), extb=buf(M, lo.Kext, zero=True),
                    U=(buf(lo.Rp, mp, zero=True), buf(lo.Rp, mp, zero=True)), V=(buf(lo.Rp, mp, zero=True), buf(lo.Rp, mp, zero=True)))

    def _site_fwd(self, p, lw, sb, X, ldx, M, rpb=None, x_map=(0, 0)):
        """u = x A^T (fp32-accurate), returns the K-extension arguments of the site's GEMM."""
        if lw.lora is None:
            return {}
        lo = lw.lora
        self._down(p, X=X, ldx=ldx, M=M, K=lo.A_hi.shape[1], W_hi=lo.A_hi, W_lo=lo.A_lo, ldw=lo.A_hi.stride(0), R=lo.Rp, Ut=sb["U"],
                   ext=sb["ext"], ld_ext=sb["ext"].stride(0), rpb=rpb, x_map=x_map)
        return dict(A2=sb["ext"], lda2=sb["ext"].stride(0), B2=lo.We, ldb2=lo.We.stride(0), K2=lo.Kext)

    def _site_bwd(self, p, lw, sb, dY, ldy, M, Xin, ldxin, rpb=None, dy_map=(0, 0), x_map=(0, 0), WeT=None, defer=None):
        """v = dy (sB)^T, dB += dy^T u, dA += v^T x; returns the K-extension arguments of the site's dX GEMM.  defer: list that
        collects the two weight-gradient problems for a batched launch by the caller (who keeps dY / Xin / the site buffers intact
        until it flushes)."""
        if lw.lora is None:
            return {}
        lo = lw.lora
        self._down(p, X=dY, ldx=ldy, M=M, K=lw.N, W_hi=lo.Bt_hi, W_lo=lo.Bt_lo, ldw=lo.Bt_hi.stride(0), R=lo.Rp, Ut=sb["V"],
                   ext=sb["extb"], ld_ext=sb["extb"].stride(0), rpb=rpb, x_map=dy_map)
        self._grad(p, Vt=sb["U"], R=lo.Rp, r_valid=lo.r, X=dY, ldx=ldy, M=M, K=lw.N, G=lo.gB, g_sr=1, g_sc=lo.r, out_scale=lo.scale,
                   rpb=rpb, x_map=dy_map, defer=defer)
        self._grad(p, Vt=sb["V"], R=lo.Rp, r_valid=lo.r, X=Xin, ldx=ldxin, M=M, K=lo.A_hi.shape[1], G=lo.gA, g_sr=lo.A_hi.shape[1], g_sc=1,
                   rpb=rpb, x_map=x_map, defer=defer)
        WeT = lo.WeT if WeT is None else WeT
        return dict(A2=sb["extb"], lda2=sb["extb"].stride(0), B2=WeT, ldb2=WeT.stride(0), K2=lo.Kext)

    # ------------------------------------------------------------------ forward program
    def _build_forward(self, P):
        A, B, D, S, H, dh, T, S_i = self.A, self.B, self.D, self.S, self.H, self.dh, self.T, self.S_i
        S_pad = self.S_pad
        p = self.fwd
        model = self.model
        cfg = model.config
        Lyr = cfg.num_layers
        Jd = cfg.joint_attention_dim
        rows, rpb, off = self.rows, self.rpb, self.off
        eps = 1e-6
        # head: timestep embedding -> temb ; img_in ; txt_norm + txt_in ; all modulation vectors in one GEMV launch
        p.c(lib.qfx_timestep_embed, _ptr(A["t"]), B, 256, 1000.0, 1.0, _ptr(A["tproj"]))
        if self.cond:   # adapters on the conditioning head: base GEMVs + the banks' rank-r launches (cond_hip.py)
            self.cond_head.emit_forward(p)
        else:
            p.c(lib.qfx_mod_gemv, _ptr(A["tproj"]), B, 256, _ptr(P["t1_Wp"]), _ptr(P["t1_bp"]), 1, D, 0, _ptr(A["t1"]))
            p.c(lib.qfx_mod_gemv, _ptr(A["t1"]), B, D, _ptr(P["t2_Wp"]), _ptr(P["t2_bp"]), 1, D, 1, _ptr(A["temb"]))
            # Round 6 lever (QFX_SIDE_MOD=1, default OFF): the modulation GEMVs of all blocks are ONE pass over 13.6 GB of frozen weights at
            # HBM speed (2.2 ms of a 93 ms step with the matrix pipes idle).  With the lever only block 0's two matrices stay on the main
            # stream, the rest go out on the side stream in three launches under the first blocks' GEMMs, each joined in front of the first
            # block that reads its rows.  Measured (profiles/r06_step_side_mod.json): 95.35 vs 95.31 ms -- nothing.  The persistent GEMM
            # blocks (12 waves x ~160 registers) leave no register file for a second kernel's waves on their CUs, so the side launches only
            # run in the seams between GEMM launches: the same wall the side-stream gradient launches hit.
            mod_joins = {}
            if Lyr > 4 and os.environ.get("QFX_SIDE_MOD", "0") == "1":
                side = ops.side_stream(model.device, 0)
                p.side = side

                def gemv(i0, i1, on_side):
                    (p.c_side if on_side else p.c)(lib.qfx_mod_gemv, _ptr(A["temb"]), B, D, _ptr(P["mod_W"]) + 8 * i0, _ptr(P["mod_b"]) + 8 * i0,
                                                   i1 - i0, 6 * D, 1, A["mods"][i0].data_ptr())
                gemv(0, 2, False)
                ev_fork = ops.Event()
                p.keep.append(ev_fork)

                def fork(ev=ev_fork, side=side):
                    ev.record(torch.cuda.current_stream())
                    ev.wait(side)
                p.py(fork)
                for b0, b1 in ((1, 3), (3, 9), (9, Lyr)):
                    if b0 >= Lyr:
                        break
                    b1 = min(b1, Lyr)
                    gemv(2 * b0, 2 * b1, True)
                    ev = ops.Event()
                    p.keep.append(ev)
                    p.py(lambda ev=ev, side=side: ev.record(side))
                    mod_joins[b0] = ev
            else:
                p.c(lib.qfx_mod_gemv, _ptr(A["temb"]), B, D, _ptr(P["mod_W"]), _ptr(P["mod_b"]), 2 * Lyr, 6 * D, 1, _ptr(A["mods"]))
            p.c(lib.qfx_mod_gemv, _ptr(A["temb"]), B, D, _ptr(P["norm_out_Wp"]), _ptr(P["norm_out_bp"]), 1, 2 * D, 1, _ptr(A["mod_out"]))
        kw = self._site_fwd(p, P["img_in"], A["site"]["img_in"], A["in_img"], cfg.in_channels, rows["img"])
        self._gemm(p, A1=A["in_img"], lda1=cfg.in_channels, B1=P["img_in"].W, K1=cfg.in_channels, M=rows["img"], N=D,
                   C_=A["X"]["img"][0], ldc=D, bias=P["img_in"].b, row_mask=self.rmask["img"], **kw)
        p.c(lib.qfx_rmsnorm_fwd, _ptr(A["in_txt"]), _ptr(model.txt_norm.weight.data), _ptr(A["txt_n"]), rows["txt"], Jd, eps)
        kw = self._site_fwd(p, P["txt_in"], A["site"]["txt_in"], A["txt_n"], Jd, rows["txt"])
        self._gemm(p, A1=A["txt_n"], lda1=Jd, B1=P["txt_in"].W, K1=Jd, M=rows["txt"], N=D, C_=A["X"]["txt"][0], ldc=D,
                   bias=P["txt_in"].b, row_mask=self.rm_txt0, **kw)
        self.attn_args = []
        for i in range(Lyr):
            if not self.cond and i in mod_joins:      # this block's modulation rows come from the side stream
                p.py(lambda ev=mod_joins[i]: ev.wait(torch.cuda.current_stream()))
            mods = {"img": A["mods"][2 * i], "txt": A["mods"][2 * i + 1]}   # [B, 6D]: shift1 scale1 gate1 shift2 scale2 gate2
            self._emit_double_fwd(p, P["blocks"][i], A["blk"][i], mods, {s: A["X"][s][i] for s in ("img", "txt")},
                                  {s: (A["X"][s][i + 1], (0, 0)) for s in ("img", "txt")}, last=(i == Lyr - 1), norm_flags=0, par=i % self._ncopy)
        mo = A["mod_out"][0]  # [B, 2D]: scale | shift  (AdaLayerNormContinuous chunk order)
        p.c(lib.qfx_ln_modulate_fwd, _ptr(A["X"]["img"][Lyr]), _ptr(mo[:, D:2 * D]), _ptr(mo[:, 0:D]), 2 * D, _ptr(A["xn_out"]),
            rows["img"], D, rpb["img"], eps)
        po = P["proj_out"]
        kw = self._site_fwd(p, po, A["site"]["proj_out"], A["xn_out"], D, rows["img"])
        self._gemm(p, A1=A["xn_out"], lda1=D, B1=po.W, K1=D, M=rows["img"], N=po.N, C_=A["out"], ldc=po.N, bias=po.b,
                   row_mask=self.rmask["img"], **kw)

    def _emit_double_fwd(self, p, w, bb, mods, x_in, x_out, last, norm_flags, par=0):
        """One double-stream block (reference: transformer_qwenimage.py:425-494; FLUX: transformer_flux.py:467-523).
        x_in[s]: [rows_s, D] block input; x_out[s] = (tensor, c_map): where the block output goes (possibly a joint buffer)."""
        A, B, D, S, H, dh, T = self.A, self.B, self.D, self.S, self.H, self.dh, self.T
        S_pad = self.S_pad
        rows, rpb, off = self.rows, self.rpb, self.off
        eps = 1e-6
        scale = 1.0 / math.sqrt(dh)
        STREAMS = (("img", 0), ("txt", 1))
        if True:
            qkv = bb["qkv"]
            q2 = qkv.view(B * S, 3 * D)
            sqk2 = bb["sqk"].view(B * S, 2 * D)
            ao2 = bb["ao"].view(B * S, D)
            # ---- LN1 + modulate, LoRA down-projections, then ONE grouped launch for the 6 q/k/v projections
            groups = []
            lnl = []
            ents = []
            for s, sidx in STREAMS:
                mod = mods[s]
                grp = w[s + ".qkv_lora"]
                xm1 = bb["xm1." + s] if grp is not None else A["xm"][s]
                ln = self._ln_fwd_args(x_in[s], mod[:, 0:D], mod[:, D:2 * D], 6 * D, xm1, rows[s], D, rpb[s], eps)
                pq_ = self._preq_out(p, xm1, D, rows[s], D, s)       # MX-FP8 trunk: the q/k/v GEMMs take xm1 quantised by its producer
                if pq_ is not None:
                    ln.yq, ln.ys, ln.ldyq, ln.ys_rows = _ptr(pq_[0]), _ptr(pq_[1]), D, rows[s]
                lnl.append(ln)
                ents.append((ln, None if grp is None else dict(W_hi=grp["A_hi"], W_lo=grp["A_lo"], W_fr=grp.get("A_fr"), ldw=D, R=3 * grp["Rp"],
                                                               Ut=bb["Uqkv." + s], ext=A["ext3"][s], ld_ext=A["ext3"][s].stride(0),
                                                               group_R=grp["Rp"], group_stride=grp["Kext"])))
            # LayerNorm+modulate and the q/k/v down projection of its output in ONE pass over the row block (qfx_ln_down_fwd)
            fused = self._ln_down(p, ents)
            if fused:
                lnl.clear()
            else:
                self._flush_ln(p, lnl, L.LnFwdArgs, lib.qfx_ln_modulate_fwd_batch)
            for s, sidx in STREAMS:
                mod = mods[s]
                x = x_in[s]
                grp = w[s + ".qkv_lora"]
                xm1 = bb["xm1." + s] if grp is not None else A["xm"][s]
                if grp is not None and not fused:
                    self._down(p, X=xm1, ldx=D, M=rows[s], K=D, W_hi=grp["A_hi"], W_lo=grp["A_lo"], ldw=D, R=3 * grp["Rp"],
                               Ut=bb["Uqkv." + s], ext=A["ext3"][s], ld_ext=A["ext3"][s].stride(0),
                               group_R=grp["Rp"], group_stride=grp["Kext"])
                for sec in range(3):
                    lw = w[s + ".qkv"][sec]
                    kw = {}
                    if lw.lora is not None:
                        kw = dict(A2=A["ext3"][s][:, sec * grp["Kext"]:], lda2=A["ext3"][s].stride(0), B2=lw.lora.We,
                                  ldb2=lw.lora.We.stride(0), K2=lw.lora.Kext)
                    # q and k go straight into the block's saved pre-norm copy (what the backward of the QK norm needs); the
                    # norm+RoPE pass reads them there and writes the joint buffer (out-of-place mode: no copy pass)
                    c_, ldc = (sqk2[:, sec * D:], 2 * D) if sec < 2 else (q2[:, 2 * D:], 3 * D)
                    groups.append(self._gargs(A1=xm1, lda1=D, B1=lw.W, K1=D, M=rows[s], N=D, C_=c_, ldc=ldc,
                                              bias=lw.b, rpb=rpb[s], c_map=(S, off[s]), **kw))
            self._gemm_group(p, groups)
            nq_t, nk_t, nq_i, nk_i = w["norms"]
            p.c(lib.qfx_qk_norm_rope_fwd, _ptr(qkv), _ptr(bb["sqk"]), _ptr(self.rope), _ptr(nq_t), _ptr(nk_t), _ptr(nq_i), _ptr(nk_i),
                B, S, T, H, dh, eps, norm_flags | 2, self.rope_bs)
            a = L.AttnArgs()
            a.B, a.S, a.S_pad, a.H, a.dh, a.scale = B, S, S_pad, H, dh, scale
            a.Q, a.K, a.V = _ptr(q2[:, 0:]), _ptr(q2[:, D:]), _ptr(q2[:, 2 * D:])
            a.ldq = a.ldk = a.ldv = 3 * D
            a.O, a.ldo, a.lse2 = _ptr(bb["ao"]), D, _ptr(bb["lse"])   # no transposed copies: the kernels use LDS transpose reads
            a.key_mask = _ptr(self.kmask)
            # backward fields (same struct reused by the backward program)
            a.dsum = _ptr(A["dsum"])
            a.dO, a.lddo = _ptr(A["dao"]), D
            dq2 = self._sb("dqkv", par).view(B * S, 3 * D)
            a.dQ, a.dK, a.dV = _ptr(dq2[:, 0:]), _ptr(dq2[:, D:]), _ptr(dq2[:, 2 * D:])
            a.lddq = a.lddk = a.lddv = 3 * D
            self._fuse_qk_bwd(a, bb["sqk"], (nq_t, nk_t, nq_i, nk_i), norm_flags, eps)
            # the text stream of the last block never reaches the output (:661-663): dead compute, skipped
            live = [(s, sidx) for s, sidx in STREAMS if not (last and s == "txt")]
            hl_o = self._head_lora_slots(a, w, [s for s, _ in live])
            self.attn_args.append(a)
            p.c(lib.qfx_attn_fwd, C.byref(a))
            groups = []
            dfo = []
            dho = []
            for s, sidx in live:
                lw = w[s + ".o"]
                kw = {}
                if lw.lora is not None and s in hl_o:
                    # u = ao A_o^T left the attention epilogue as per-head partial sums: reduce + pack (what qfx_lora_down wrote)
                    dho.append(self._head_reduce_args(A["hl_o"], hl_o[s], lw.lora.Rp, rows[s], rpb[s], off[s], A["ext1"][s], bb["Uo." + s],
                                                      lw.lora.Rp, 0))
                    kw = dict(A2=A["ext1"][s], lda2=A["ext1"][s].stride(0), B2=lw.lora.We, ldb2=lw.lora.We.stride(0), K2=lw.lora.Kext)
                elif lw.lora is not None:
                    # MX-FP8 trunk: the down projection reads every attention-output row of this stream anyway and leaves its
                    # MX-FP8 image for the out-projection GEMM
                    pq_ = self._preq_out(p, ao2, D, rows[s], D, "ao." + s, a_map=(S, off[s]))
                    self._down(p, X=ao2, ldx=D, M=rows[s], K=D, W_hi=lw.lora.A_hi, W_lo=lw.lora.A_lo, ldw=D, R=lw.lora.Rp,
                               Ut=bb["Uo." + s], ext=A["ext1"][s], ld_ext=A["ext1"][s].stride(0),
                               rpb=rpb[s], x_map=(S, off[s]), defer=dfo,
                               xq=None if pq_ is None else (_ptr(pq_[0]), _ptr(pq_[1]), D, rows[s], 0))
                    kw = dict(A2=A["ext1"][s], lda2=A["ext1"][s].stride(0), B2=lw.lora.We, ldb2=lw.lora.We.stride(0), K2=lw.lora.Kext)
                if "y1" in bb:
                    kw.update(C2=bb["y1"][s], ldc2=D)
                groups.append(self._gargs(A1=ao2, lda1=D, B1=lw.W, K1=D, M=rows[s], N=D, C_=bb["x1"][s], ldc=D, bias=lw.b,
                                          epi=L.EPI_GATE_RES, aux=x_in[s], ldaux=D, gate=mods[s][:, 2 * D:3 * D], gate_bs=6 * D,
                                          rpb=rpb[s], a_map=(S, off[s]), **kw))
            self._flush_batch(p, dfo, L.LoraDownArgs, lib.qfx_lora_down_batch)
            self._flush_head_reduce(p, dho)
            self._gemm_group(p, groups)
            groups = []
            # feed-forward (+ LoRA on net.0.proj / net.2: the adapter's input is then kept per block instead of in scratch)
            xm2 = {s: (bb["xm2." + s] if w[s + ".fc1"].lora is not None else A["xm"][s]) for s, _ in live}
            gact = {s: (bb["g." + s] if w[s + ".fc2"].lora is not None else A["g"][s]) for s, _ in live}
            lnl = [self._ln_fwd_args(bb["x1"][s], mods[s][:, 3 * D:4 * D], mods[s][:, 4 * D:5 * D], 6 * D, xm2[s], rows[s], D, rpb[s], eps)
                   for s, sidx in live]
            for ln, (s, sidx) in zip(lnl, live):
                pq_ = self._preq_out(p, xm2[s], D, rows[s], D, s)    # ... and fc1 takes xm2
                if pq_ is not None:
                    ln.yq, ln.ys, ln.ldyq, ln.ys_rows = _ptr(pq_[0]), _ptr(pq_[1]), D, rows[s]
            self._flush_ln(p, lnl, L.LnFwdArgs, lib.qfx_ln_modulate_fwd_batch)

            def lora_ext(s, lw, X, ldx, ukey):
                """Down-projection of a single adapted linear; returns the K-extension arguments of its GEMM."""
                if lw.lora is None:
                    return {}
                lo, e1 = lw.lora, A["ext1"][s]
                self._down(p, X=X, ldx=ldx, M=rows[s], K=lw.K, W_hi=lo.A_hi, W_lo=lo.A_lo, ldw=lo.A_hi.stride(0), R=lo.Rp,
                           Ut=bb[ukey + s], ext=e1, ld_ext=e1.stride(0), defer=dfw)
                return dict(A2=e1, lda2=e1.stride(0), B2=lo.We, ldb2=lo.We.stride(0), K2=lo.Kext)

            dfw = []      # the image- and text-stream down projections of one site go out as ONE batched launch
            for s, sidx in live:
                f1 = w[s + ".fc1"]
                kw = lora_ext(s, f1, xm2[s], D, "Uf1.")
                groups.append(self._gargs(A1=xm2[s], lda1=D, B1=f1.W, K1=D, M=rows[s], N=4 * D, C_=bb["h"][s], ldc=4 * D,
                                          bias=f1.b, epi=L.EPI_GELU, C2=gact[s], ldc2=4 * D, **kw))
                groups[-1]._next = (gact[s], 4 * D, w[s + ".fc2"].lora is not None)    # gelu(h) feeds fc2 (and its adapter's dA, if any)
            self._flush_batch(p, dfw, L.LoraDownArgs, lib.qfx_lora_down_batch)
            self._gemm_group(p, groups)
            groups = []
            for s, sidx in live:
                f2 = w[s + ".fc2"]
                kw = lora_ext(s, f2, gact[s], 4 * D, "Uf2.")
                if "y2" in bb:
                    kw.update(C2=bb["y2"][s], ldc2=D)
                groups.append(self._gargs(A1=gact[s], lda1=4 * D, B1=f2.W, K1=4 * D, M=rows[s], N=D, C_=x_out[s][0], ldc=D,
                                          bias=f2.b, epi=L.EPI_GATE_RES, aux=bb["x1"][s], ldaux=D, gate=mods[s][:, 5 * D:6 * D],
                                          gate_bs=6 * D, rpb=rpb[s], c_map=x_out[s][1], aux_unmapped=1, row_mask=self.rmask[s], **kw))
            self._flush_batch(p, dfw, L.LoraDownArgs, lib.qfx_lora_down_batch)
            self._gemm_group(p, groups)

    # ------------------------------------------------------------------ backward program
    def _build_backward(self, P):
        A, B, D, S, H, dh, T, S_i = self.A, self.B, self.D, self.S, self.H, self.dh, self.T, self.S_i
        S_pad = self.S_pad
        p = self.bwd
        cfg = self.model.config
        Lyr = cfg.num_layers
        rows, rpb, off = self.rows, self.rpb, self.off
        eps = 1e-6
        po = P["proj_out"]
        # tail: proj_out dX (+ its adapter), norm_out LN backward (+ gate2 of the last block folded in)
        kw = self._site_bwd(p, po, A["site"]["proj_out"], A["dpred"], po.N, rows["img"], A["xn_out"], D)
        self._gemm(p, A1=A["dpred"], lda1=po.N, B1=po.WT, K1=po.N, M=rows["img"], N=D, C_=A["dxn"], ldc=D,
                   row_mask=self.rmask["img"], **kw)   # backward of the output masked_fill: no gradient enters through padded rows
        mo = A["mod_out"][0]
        modL = A["mods"][2 * (Lyr - 1)]
        cur = 0
        if self.cond:
            p.py(A["dmods"].zero_)
            p.py(A["dmod_out"].zero_)
            dmo = A["dmod_out"][0]       # [B, 2D] = d scale | d shift (AdaLayerNormContinuous chunk order)
            self._mod_grad(p, dy=A["dxn"], x=A["X"]["img"][Lyr], rows=rows["img"], rpb=rpb["img"], dshift=dmo[:, D:2 * D],
                           dscale=dmo[:, 0:D], out_bs=2 * D, row_mask=self.rmask["img"])
        p.c(lib.qfx_ln_modulate_bwd, _ptr(A["dxn"]), _ptr(A["X"]["img"][Lyr]), _ptr(mo[:, 0:D]), 2 * D, None,
            _ptr(modL[:, 5 * D:6 * D]), 6 * D, _ptr(A["dX"]["img"][cur]), _ptr(A["dyg2"]["img"]), rows["img"], D, rpb["img"], eps, None)
        for i in range(Lyr - 1, -1, -1):
            nxt = cur ^ 1
            mods = {"img": A["mods"][2 * i], "txt": A["mods"][2 * i + 1]}
            gate_prev = None if i == 0 else {"img": A["mods"][2 * (i - 1)][:, 5 * D:6 * D], "txt": A["mods"][2 * (i - 1) + 1][:, 5 * D:6 * D]}
            self._emit_double_bwd(p, P["blocks"][i], A["blk"][i], self.attn_args[i], mods, {s: A["X"][s][i] for s in ("img", "txt")},
                                  dx2={s: A["dX"][s][cur] for s in ("img", "txt")}, out_dx={s: A["dX"][s][nxt] for s in ("img", "txt")},
                                  gate_prev=gate_prev, last=(i == Lyr - 1), first=(i == 0 and not self.full_bwd), norm_flags=0,
                                  prefix=f"transformer_blocks.{i}.", par=i % self._ncopy,
                                  dmods=({"img": A["dmods"][2 * i], "txt": A["dmods"][2 * i + 1]} if self.cond else None))
            if not self.side_grads:
                p.mark(f"transformer_blocks.{i}.")
            cur = nxt
        self._emit_pending_side(p)
        self._side_join(p)
        # head: the embedders' adapters (their inputs carry no gradient: rank-r launches only); d(block-0 input) = A["dX"][s][cur]
        if self.in_grad:
            self._site_bwd(p, P["img_in"], A["site"]["img_in"], A["dX"]["img"][cur], D, rows["img"], A["in_img"], cfg.in_channels)
            self._site_bwd(p, P["txt_in"], A["site"]["txt_in"], A["dX"]["txt"][cur], D, rows["txt"], A["txt_n"], cfg.joint_attention_dim)
        if self.cond:
            self.cond_head.emit_backward(p)

    def _emit_double_bwd(self, p, w, bb, a, mods, x_in, dx2, out_dx, gate_prev, last, first, norm_flags, prefix=None, par=0, dmods=None):
        """Backward of one double-stream block.  In: dx2[s] = d(block output), A["dyg2"][s] = gate2*dx2 (emitted by whoever
        produced dx2).  Out: out_dx[s] = d(block input) and A["dyg2"][s] = gate_prev*out_dx (for the previous block)."""
        A, B, D, S, H, dh, T = self.A, self.B, self.D, self.S, self.H, self.dh, self.T
        S_pad = self.S_pad
        rows, rpb, off = self.rows, self.rpb, self.off
        eps = 1e-6
        dao2 = A["dao"].view(B * S, D)
        dqkv, dyg1, VtO, VtQ = self._sb("dqkv", par), self._sb("dyg1", par), self._sb("VtO", par), self._sb("Vt", par)
        ff_side = self.side_grads and self._ff_side
        dh_ = self._sb("dh", par) if ff_side else A["dh"]
        vtf = {"VtF1": self._sb("VtF1", par) if ff_side else A.get("VtF1"), "VtF2": self._sb("VtF2", par) if ff_side else A.get("VtF2")}
        if ff_side and self._ncopy == 2:
            self._side_join(p, keep=0 if self._side_late else 1)      # the launch of block i+2 read this parity's dh: overwritten by this block's first GEMM
        dq2 = dqkv.view(B * S, 3 * D)
        STREAMS = (("img", 0), ("txt", 1))
        i = 0 if first else 1
        # LoRA weight gradients are leaves: every qfx_lora_grad of the block is deferred to ONE batched launch per rank at the
        # end of the block (their X operands -- dyg1, ao, dqkv, xm1, dh, the kept feed-forward inputs -- stay intact until the next block's backward
        # starts; the one exception, dyg2, is flushed early)
        gl = []
        if True:
            ao2 = bb["ao"].view(B * S, D)
            live = [(s, sidx) for s, sidx in STREAMS if not (last and s == "txt")]
            if last:
                # no gradient reaches the last block's text tail: d(attn out) of the text rows is zero
                p.py(A["dao"][:, :T].zero_)
            # ---- MLP backward: dh = (gate2*dx2) W2 * gelu'(h) ; dxm2 = dh W1   (both streams per launch)
            ge = []   # gradients whose X operand (dyg2) is overwritten before the end of the block: flushed right after the MLP

            def lora_bwd(s, lw, dY, ldy, Xin, ldxin, ukey, vkey, early):
                """dY -> v = dY B (K-extension of the dX GEMM) + the two deferred weight-gradient problems of an adapted linear."""
                if lw.lora is None:
                    return {}
                lo, e1 = lw.lora, A["ext1"][s]
                Vt = (vtf[vkey][s][0][:lo.Rp], vtf[vkey][s][1][:lo.Rp])
                self._down(p, X=dY, ldx=ldy, M=rows[s], K=lw.N, W_hi=lo.Bt_hi, W_lo=lo.Bt_lo, ldw=lo.Bt_hi.stride(0), R=lo.Rp,
                           Ut=Vt, ext=e1, ld_ext=e1.stride(0), defer=dbw)
                self._grad(p, Vt=bb[ukey + s], R=lo.Rp, r_valid=lo.r, X=dY, ldx=ldy, M=rows[s], K=lw.N, G=lo.gB, g_sr=1, g_sc=lo.r,
                           out_scale=lo.scale, defer=ge if early else gl)
                self._grad(p, Vt=Vt, R=lo.Rp, r_valid=lo.r, X=Xin, ldx=ldxin, M=rows[s], K=lw.K, G=lo.gA, g_sr=lw.K, g_sc=1, defer=gl)
                return dict(A2=e1, lda2=e1.stride(0), B2=lo.WeT, ldb2=lo.WeT.stride(0), K2=lo.Kext)

            groups = []
            dbw = []      # both streams' v = dY B of a site in ONE batched launch
            for s, _ in live:
                f2 = w[s + ".fc2"]
                kw = lora_bwd(s, f2, A["dyg2"][s], D, bb.get("g." + s), 4 * D, "Uf2.", "VtF2", early=True)
                groups.append(self._gargs(A1=A["dyg2"][s], lda1=D, B1=f2.WT, K1=D, M=rows[s], N=4 * D, C_=dh_[s],
                                          ldc=4 * D, epi=L.EPI_DGELU, aux=bb["h"][s], ldaux=4 * D, **kw))
                groups[-1]._next = (dh_[s], 4 * D, w[s + ".fc1"].lora is not None)   # dh feeds fc1's dX GEMM (and its adapter's v / dB)
            self._flush_batch(p, dbw, L.LoraDownArgs, lib.qfx_lora_down_batch)
            self._gemm_group(p, groups)
            groups = []
            for s, _ in live:
                f1 = w[s + ".fc1"]
                kw = lora_bwd(s, f1, dh_[s], 4 * D, bb.get("xm2." + s), D, "Uf1.", "VtF1", early=False)
                groups.append(self._gargs(A1=dh_[s], lda1=4 * D, B1=f1.WT, K1=4 * D, M=rows[s], N=D, C_=A["dxm"][s], ldc=D, **kw))
            self._flush_batch(p, dbw, L.LoraDownArgs, lib.qfx_lora_down_batch)
            self._gemm_group(p, groups)
            if ge:
                self._flush_batch(p, ge, L.LoraGradArgs, lib.qfx_lora_grad_batch)
            if dmods is not None:   # d(shift2, scale2, gate2): dy = d(xm2) (fc1 dX output), LN input x1, gate side dx2 * y2
                mg = [] if os.environ.get("QFX_MOD_GRAD_BATCH", "1") != "0" else None     # None: one launch per stream (A/B switch)
                for s, _ in live:
                    dm = dmods[s]
                    self._mod_grad(p, dy=A["dxm"][s], x=bb["x1"][s], rows=rows[s], rpb=rpb[s], dshift=dm[:, 3 * D:4 * D],
                                   dscale=dm[:, 4 * D:5 * D], dgate=dm[:, 5 * D:6 * D], dxo=dx2[s], y=bb["y2"][s], out_bs=6 * D,
                                   row_mask=self.rmask[s], defer=mg)
                if mg:
                    self._flush_mod_grad(p, mg)
            if self._ncopy == 2:
                self._side_join(p, keep=0 if self._side_late else 1)   # the launch of block i+2 read this parity's dyg1 / dqkv / v^T scratch: overwritten from here on
            groups = []
            lnl = [self._ln_bwd_args(A["dxm"][s], bb["x1"][s], mods[s][:, 4 * D:5 * D], 6 * D, dx2[s], mods[s][:, 2 * D:3 * D], 6 * D,
                                     A["dx1"][s], dyg1[s], rows[s], D, rpb[s], eps, None) for s, sidx in live]
            for ln, (s, sidx) in zip(lnl, live):
                pq_ = self._preq_out(p, dyg1[s], D, rows[s], D, s)   # "mxfp8-fb": the out-projection dX GEMM takes gate1*dx quantised
                if pq_ is not None:
                    ln.dygq, ln.dygs, ln.lddygq, ln.dygs_rows = _ptr(pq_[0]), _ptr(pq_[1]), D, rows[s]
            self._flush_ln(p, lnl, L.LnBwdArgs, lib.qfx_ln_modulate_bwd_batch)
            dbo = []
            for s, sidx in live:
                mod = mods[s]
                # attention out-projection backward (+ LoRA)
                lw = w[s + ".o"]
                kw = {}
                if lw.lora is not None:
                    lo = lw.lora
                    Vt = (VtO[s][0][:lo.Rp], VtO[s][1][:lo.Rp])
                    self._down(p, X=dyg1[s], ldx=D, M=rows[s], K=lw.N, W_hi=lo.Bt_hi, W_lo=lo.Bt_lo, ldw=lo.Bt_hi.stride(0),
                               R=lo.Rp, Ut=Vt, ext=A["ext1"][s], ld_ext=A["ext1"][s].stride(0), defer=dbo)
                    self._grad(p, Vt=bb["Uo." + s], R=lo.Rp, r_valid=lo.r, X=dyg1[s], ldx=D, M=rows[s], K=lw.N,
                               G=lo.gB, g_sr=1, g_sc=lo.r, out_scale=lo.scale, defer=gl)
                    self._grad(p, Vt=Vt, R=lo.Rp, r_valid=lo.r, X=ao2, ldx=D, M=rows[s], K=lw.K, G=lo.gA,
                               g_sr=lw.K, g_sc=1, rpb=rpb[s], x_map=(S, off[s]), defer=gl)
                    kw = dict(A2=A["ext1"][s], lda2=A["ext1"][s].stride(0), B2=lo.WeT, ldb2=lo.WeT.stride(0), K2=lo.Kext)
                groups.append(self._gargs(A1=dyg1[s], lda1=D, B1=lw.WT, K1=lw.N, M=rows[s], N=lw.K, C_=dao2, ldc=D, rpb=rpb[s],
                                          c_map=(S, off[s]), **kw))
            self._flush_batch(p, dbo, L.LoraDownArgs, lib.qfx_lora_down_batch)
            self._gemm_group(p, groups)
            # ---- attention backward
            q2 = bb["qkv"].view(B * S, 3 * D)
            self._emit_pending_side(p)          # (QFX_SIDE_AT_ATTN: the previous block's gradient launches start here)
            ops.emit_attn_backward(p, a, A)      # two-pass pair, or the one-pass kernel (QFX_ATTN_BWD)
            if not a.qk_saved:      # (else: the backward of the QK norm + RoPE runs in the epilogues of the two kernels above)
                nq_t, nk_t, nq_i, nk_i = w["norms"]
                p.c(lib.qfx_qk_norm_rope_bwd, _ptr(dqkv), _ptr(bb["sqk"]), _ptr(self.rope), _ptr(nq_t), _ptr(nk_t), _ptr(nq_i),
                    _ptr(nk_i), B, S, T, H, dh, eps, norm_flags, self.rope_bs)
            # ---- q/k/v projection backward (+ LoRA), both streams in one launch
            groups = []
            dl = []   # the q/k/v down projections of both streams: one batched launch
            dhq = []  # ... or, fused into the attention epilogues, their reduce + pack halves
            for s, sidx in STREAMS:
                grp = w[s + ".qkv_lora"]
                kw = {}
                if grp is not None:
                    Rp, Kext = grp["Rp"], grp["Kext"]
                    Vth, Vtl = VtQ[s]
                    Uth, Utl = bb["Uqkv." + s]
                    e3 = A["ext3"][s]
                    # "mxfp8-fb": the three down projections together read every element of this stream's dqkv rows and leave
                    # its MX-FP8 image (q, k, v column sections of one operand) for the qkv dX GEMM
                    pq_ = None
                    if i > 0 and all(w[s + ".qkv"][sec].lora is not None for sec in range(3)):
                        pq_ = self._preq_out(p, dq2, 3 * D, rows[s], 3 * D, "dqkv." + s, a_map=(S, off[s]))
                    hl_fused = s in getattr(a, "_hl_qkv", {})
                    if hl_fused:     # v = d(pre-norm q | k), dV times (sB)^T left the attention epilogues as per-head partial sums
                        dhq.append(self._head_reduce_args(A["hl_qkv"], H, 3 * Rp, rows[s], rpb[s], off[s], e3, (Vth[:3 * Rp], Vtl[:3 * Rp]),
                                                          Rp, Kext))
                    for sec in range(3):
                        lw = w[s + ".qkv"][sec]
                        if lw.lora is None:
                            continue
                        lo = lw.lora
                        sl = slice(sec * Rp, (sec + 1) * Rp)
                        if not hl_fused:
                            self._down(p, X=dq2[:, sec * D:], ldx=3 * D, M=rows[s], K=D, W_hi=lo.Bt_hi, W_lo=lo.Bt_lo,
                                       ldw=lo.Bt_hi.stride(0), R=Rp, Ut=(Vth[sl], Vtl[sl]), ext=e3[:, sec * Kext:],
                                       ld_ext=e3.stride(0), rpb=rpb[s], x_map=(S, off[s]), defer=dl,
                                       xq=None if pq_ is None else (_ptr(pq_[0]) + sec * D, _ptr(pq_[1]), 3 * D, rows[s], sec * D // 32))
                        self._grad(p, Vt=(Uth[sl], Utl[sl]), R=Rp, r_valid=lo.r, X=dq2[:, sec * D:], ldx=3 * D,
                                   M=rows[s], K=D, G=lo.gB, g_sr=1, g_sc=lo.r, rpb=rpb[s], x_map=(S, off[s]), out_scale=lo.scale,
                                   defer=gl)
                    los = [w[s + ".qkv"][sec].lora for sec in range(3)]
                    if all(l is not None for l in los):   # one pass over xm1 for dA of q, k and v
                        self._grad(p, Vt=(Vth[:3 * Rp], Vtl[:3 * Rp]), R=3 * Rp, r_valid=los[0].r, group_R=Rp, X=bb["xm1." + s],
                                   ldx=D, M=rows[s], K=D, G=[l.gA for l in los], g_sr=D, g_sc=1, defer=gl)
                    else:
                        for sec, lo in enumerate(los):
                            if lo is not None:
                                sl = slice(sec * Rp, (sec + 1) * Rp)
                                self._grad(p, Vt=(Vth[sl], Vtl[sl]), R=Rp, r_valid=lo.r, X=bb["xm1." + s], ldx=D, M=rows[s],
                                           K=D, G=lo.gA, g_sr=D, g_sc=1, defer=gl)
                    kw = dict(A2=e3, lda2=e3.stride(0), B2=grp["WeT"], ldb2=grp["WeT"].stride(0), K2=3 * Kext)
                if i > 0:   # nothing upstream of block 0 needs a gradient (frozen embedders, inputs without grad)
                    groups.append(self._gargs(A1=dq2, lda1=3 * D, B1=w[s + ".qkvT"], K1=3 * D, M=rows[s], N=D, C_=A["dxm"][s], ldc=D,
                                              rpb=rpb[s], a_map=(S, off[s]), **kw))
            self._flush_batch(p, dl, L.LoraDownArgs, lib.qfx_lora_down_batch)
            self._flush_head_reduce(p, dhq)
            if self.side_grads and gl and not self._side_late and os.environ.get("QFX_SIDE_FORK", "late") == "early":
                # round-6 lever: every operand of the block's gradient launches exists from here on -- fork in front of the q/k/v dX GEMM
                # instead of behind the block's last LayerNorm backward
                self._emit_side(p, gl, prefix)
            if i > 0:
                self._gemm_group(p, groups)
                if dmods is not None:   # d(shift1, scale1, gate1): dy = d(xm1) (q/k/v dX output), LN input x_in, gate side dx1 * y1
                    mg = [] if os.environ.get("QFX_MOD_GRAD_BATCH", "1") != "0" else None
                    for s, sidx in STREAMS:
                        dm = dmods[s]
                        dead = last and s == "txt"     # no out-projection / residual gradient on the last block's text tail
                        self._mod_grad(p, dy=A["dxm"][s], x=x_in[s], rows=rows[s], rpb=rpb[s], dshift=dm[:, 0:D], dscale=dm[:, D:2 * D],
                                       dgate=None if dead else dm[:, 2 * D:3 * D], dxo=None if dead else A["dx1"][s],
                                       y=None if dead else bb["y1"][s], out_bs=6 * D, row_mask=self.rmask[s], defer=mg)
                    if mg:
                        self._flush_mod_grad(p, mg)
                lnl = []
                for s, sidx in STREAMS:
                    dres = None if (last and s == "txt") else A["dx1"][s]
                    gp = gate_prev[s] if gate_prev is not None else None
                    lnl.append(self._ln_bwd_args(A["dxm"][s], x_in[s], mods[s][:, D:2 * D], 6 * D, dres, gp,
                                                 (gp.stride(0) if gp is not None else 0), out_dx[s],
                                                 A["dyg2"][s] if gp is not None else None, rows[s], D, rpb[s], eps, self.rmask[s]))
                    pq_ = self._preq_out(p, A["dyg2"][s] if gp is not None else None, D, rows[s], D, s)   # the previous block's fc2-dX operand
                    if pq_ is not None:
                        lnl[-1].dygq, lnl[-1].dygs, lnl[-1].lddygq, lnl[-1].dygs_rows = _ptr(pq_[0]), _ptr(pq_[1]), D, rows[s]
                self._flush_ln(p, lnl, L.LnBwdArgs, lib.qfx_ln_modulate_bwd_batch)
        if self.side_grads and gl:
            if self._side_late:
                self._emit_pending_side(p)      # (a block without an attention section: nothing may be dropped)
                self._pending_side = (gl, prefix)
            else:
                self._emit_side(p, gl, prefix)
        else:
            self._flush_batch(p, gl, L.LoraGradArgs, lib.qfx_lora_grad_batch)

    def _emit_side(self, p, gl, prefix):
        """Fork, the block's batched gradient launches on the side stream, the event its join will wait for."""
        if self._ncopy == 3:
            self._side_join(p, keep=1)   # block i-1 overwrites the copy the launch of block i+2 read
        p.py(self._side_fork)
        self._flush_batch(p, gl, L.LoraGradArgs, lib.qfx_lora_grad_batch, side=True)
        ev = ops.Event()
        p.py(lambda ev=ev: ev.record(self.bwd.side))
        self._side_q.append((ev, prefix))

    def _emit_pending_side(self, p):
        if self._pending_side is not None:
            gl, prefix = self._pending_side
            self._pending_side = None
            self._emit_side(p, gl, prefix)

    def set_multires(self, img_shapes, txt_seq_lens, attention_mask, S_in=None):
        """Per-batch tables of the multi-resolution path (host-side plumbing of transformer_qwen_custom.py:72-150,175-228,
        444-512).  Per sample the joint table [text rows | image rows] starts at joint row 0, i.e. the image rows follow the
        sample's OWN text length (the reference's placement); every other row keeps the identity rotation."""
        cfg = self.model.config
        B, S, T, S_i = self.B, self.S, self.T, self.S_i
        dev = self.model.device
        batched = isinstance(img_shapes, list) and len(img_shapes) > 0 and isinstance(img_shapes[0], list)
        per = [normalize_img_shapes(sh) for sh in img_shapes] if batched else [normalize_img_shapes(img_shapes)] * B
        lens = list(txt_seq_lens) if isinstance(txt_seq_lens, (list, tuple)) else [int(txt_seq_lens)] * B
        rope = torch.zeros(B, S, self.dh // 2, 2)
        rope[..., 0] = 1.0
        if all(sh == per[0] for sh in per):      # shared RoPE: first sample's shapes, max text length (custom forward :462-470)
            tbl = qwen_joint_rope(per[0], max(lens), cfg.axes_dims_rope)
            if max(lens) != T:
                raise ValueError("max(txt_seq_lens) must equal the text sequence length (reference RoPE broadcast)")
            rope[:, : tbl.shape[0]] = tbl
        else:
            for b in range(B):
                tbl = qwen_joint_rope(per[b], int(lens[b]), cfg.axes_dims_rope)
                if tbl.shape[0] > S:
                    raise ValueError(f"sample {b}: text + image tokens ({tbl.shape[0]}) exceed the padded joint length {S}")
                rope[b, : tbl.shape[0]] = tbl
        km = torch.zeros(B, S)
        rm_i = torch.ones(B, S_i)
        rm_t = torch.ones(B, T)
        S_in = S_i if S_in is None else S_in          # image rows the caller hands over; rows S_in..S_i are ladder padding
        if attention_mask is not None or S_in < S_i:
            m = torch.zeros(B, S, dtype=torch.bool)
            if attention_mask is not None:
                am = attention_mask if attention_mask.dtype == torch.bool else attention_mask > 0
                m[:, : T + S_in] = am[:, : T + S_in].cpu()
            else:
                m[:, : T + S_in] = True
            km.masked_fill_(~m, float("-inf"))
            rm_i = m[:, T:].float()
            rm_t = m[:, :T].float()
        A = self.A
        A["rope_b"].copy_(rope.to(dev, non_blocking=True))
        A["kmask"].copy_(km.to(dev, non_blocking=True))
        A["rm_img"].copy_(rm_i.reshape(-1).to(dev, non_blocking=True))
        A["rm_txt0"].copy_(rm_t.reshape(-1).to(dev, non_blocking=True))

    # ------------------------------------------------------------------ execution
    def run_forward(self, hidden_states, encoder_hidden_states, timestep):
        A = self.A
        self._copy_rows(A["in_img"].view(self.B, self.S_i, -1), hidden_states)
        A["in_txt"].view(self.B, self.T, -1).copy_(encoder_hidden_states)
        A["t"].copy_(timestep.reshape(self.B).to(F32))
        self.model.refresh_lora_operands()
        self.fwd.run()
        return A["out"].view(self.B, self.S_i, -1)

    @staticmethod
    def _copy_rows(dst, src):
        """dst [B, S_plan, C] <- src [B, S_in <= S_plan, C]; the ladder-padding rows (masked everywhere) are zeroed."""
        n = src.shape[1]
        if n == dst.shape[1]:
            dst.copy_(src)
        else:
            dst[:, :n].copy_(src)
            dst[:, n:].zero_()

    def run_backward(self, dpred, on_segment=None):
        """on_segment(prefixes): called after each marked segment of the backward program with the parameter-name prefixes
        whose LoRA gradients just became final (data-parallel bucketed all-reduce hooks in here)."""
        if getattr(self.model, "_merged", False):
            raise RuntimeError("adapters are merged into the base weights (merge_adapter): call unmerge_adapter() before training")
        self._copy_rows(self.A["dpred"].view(self.B, self.S_i, -1), dpred)
        self.model._lora.ensure_grads()
        if on_segment is None:
            self.bwd.run()
            return
        pos = 0
        for idx, prefix in self.bwd.marks:
            self.bwd.run(pos, idx)
            pos = idx
            on_segment(prefix)
        self.bwd.run(pos, None)


class _QwenDiTFn(torch.autograd.Function):
    """Whole-DiT autograd node.  LoRA parameters are inputs only so that autograd schedules the node;
    their gradients are accumulated by the kernels directly into the flat .grad buffer (returns None)."""

    @staticmethod
    def forward(ctx, model, plan, hidden_states, encoder_hidden_states, timestep, *lora_params):
        ctx.plan = plan
        out = plan.run_forward(hidden_states, encoder_hidden_states, timestep)
        n = (hidden_states[0] if isinstance(hidden_states, tuple) else hidden_states).shape[1]
        return out[:, :n].clone()      # a ladder plan carries more rows than the caller handed over

    @staticmethod
    def backward(ctx, grad_out):
        ctx.plan.model._dp_backward(ctx.plan, grad_out.contiguous())     # + the data-parallel exchange when enabled (dp.py)
        return (None,) * len(ctx.needs_input_grad)
