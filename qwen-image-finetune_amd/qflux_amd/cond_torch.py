"""Conditioning head with adapters on its linears (timestep / guidance / pooled-text embedders, AdaLN modulation linears).

These linears see M = batch rows only (`temb` is [B, D]); their cost is the 13.6 GB weight stream of the modulation matrices.
Without adapters the head is ONE batched HIP GEMV launch (qfx_mod_gemv).  With adapters on any of them -- target_modules
"all-linear" (configs/example_with_sampling.yaml:9) or the `(norm|norm1|norm1_context).linear` alternatives of
configs/face_seg_flux_kontext_fp16.yaml:11 -- the head is evaluated here as plain library GEMVs (torch.nn.functional.linear on
the bf16 weights = rocBLAS, allowed for plain small-M linears) under autograd, with peft's formula for the adapted ones:

    y = base(x);  y = (y + lora_B(lora_A(x.float())) * scaling).to(bf16)            (peft lora.Linear.forward)

The HIP backward produces d(modulation vectors) with qfx_mod_grad (column sums over all tokens) and `backward()` below pushes
them through this small graph: adapter gradients accumulate straight into the flat LoRA gradient buffer (the parameters'
`.grad` are views of it).  Rounding points follow the reference's bf16 eager graph (transformer_qwenimage.py:143-156,430-436,
565,664; transformer_flux.py:634-639,729-741)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .modules import QfxLoraLinear

BF = torch.bfloat16


def lin(mod, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear / peft lora.Linear forward on holder modules (bf16 base, fp32 adapter)."""
    if isinstance(mod, QfxLoraLinear):
        base = mod.base_layer
        y = F.linear(x, base.weight, base.bias)
        if mod.merged:
            return y
        s = mod.scaling[mod.active_adapter]
        return (y + (x.float() @ mod.A.t()) @ mod.B.t() * s).to(y.dtype)
    return F.linear(x, mod.weight, mod.bias)


class CondHead:
    """Forward/backward of the conditioning head of one plan.  `build(tproj, ...)` returns the list of output tensors in the
    order of `outs`; their values are copied into the plan's HIP buffers, the graph is kept for `backward`."""

    def __init__(self):
        self.outs = None

    def run(self, fn, dst: list):
        """fn() -> list of tensors (same order as dst); copies values into the HIP-side buffers."""
        if torch.is_inference_mode_enabled():
            outs = fn()
            self.outs = None
        else:
            with torch.enable_grad():
                outs = fn()
            self.outs = outs
        for o, d in zip(outs, dst):
            if o is not None:
                d.copy_(o.detach().reshape(d.shape))

    def backward(self, grads: list):
        """grads: fp32 HIP-side gradients of the outputs (column sums); cast to the outputs' dtype like autograd would."""
        if self.outs is None:
            return
        outs, gs = [], []
        for o, g in zip(self.outs, grads):
            if o is not None and o.requires_grad:
                outs.append(o)
                gs.append(g.reshape(o.shape).to(o.dtype))
        if outs:
            torch.autograd.backward(outs, gs)
        self.outs = None


def qwen_head(model, tproj: torch.Tensor):
    """QwenTimestepProjEmbeddings + every block's img_mod / txt_mod + norm_out.linear.  tproj: bf16 sinusoid [B, 256]."""
    te = model.time_text_embed.timestep_embedder
    temb = lin(te.linear_2, F.silu(lin(te.linear_1, tproj)))
    s = F.silu(temb)
    mods = []
    for blk in model.transformer_blocks:
        mods.append(lin(blk.img_mod[1], s))
        mods.append(lin(blk.txt_mod[1], s))
    return [torch.stack(mods, dim=0), lin(model.norm_out.linear, s)]


def flux_head(model, tproj, gproj, pooled):
    """CombinedTimestep(Guidance)TextProjEmbeddings + norm1 / norm1_context / single-block norm linears + norm_out.linear."""
    te = model.time_text_embed
    temb = lin(te.timestep_embedder.linear_2, F.silu(lin(te.timestep_embedder.linear_1, tproj)))
    if gproj is not None:
        temb = temb + lin(te.guidance_embedder.linear_2, F.silu(lin(te.guidance_embedder.linear_1, gproj)))
    temb = temb + lin(te.text_embedder.linear_2, F.silu(lin(te.text_embedder.linear_1, pooled)))
    s = F.silu(temb)
    outs = []
    mods = []
    for blk in model.transformer_blocks:
        mods.append(lin(blk.norm1.linear, s))
        mods.append(lin(blk.norm1_context.linear, s))
    smods = [lin(blk.norm.linear, s) for blk in model.single_transformer_blocks]
    outs.append(torch.stack(mods, dim=0) if mods else None)
    outs.append(torch.stack(smods, dim=0) if smods else None)
    outs.append(lin(model.norm_out.linear, s))
    return outs
