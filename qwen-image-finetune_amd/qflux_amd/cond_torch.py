"""Conditioning head with adapters on its linears (timestep / guidance / pooled-text embedders, AdaLN modulation linears).

These linears see M = batch rows only (`temb` is [B, D]); their cost is the 13.6 GB weight stream of the modulation matrices.
Without adapters the head is ONE batched HIP GEMV launch (qfx_mod_gemv).  With adapters on any of them -- target_modules
"all-linear" (configs/example_with_sampling.yaml:9) or the `(norm|norm1|norm1_context).linear` alternatives of
configs/face_seg_flux_kontext_fp16.yaml:11 -- the head is evaluated here under autograd: the small embedder linears as library
GEMVs (torch.nn.functional.linear), the AdaLN modulation linears as banks (_BankFn: frozen base weights on the HIP GEMVs in both
directions, the adapters' rank-r terms batched), with peft's formula for the adapted ones:

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


def _base(mod):
    return mod.base_layer if isinstance(mod, QfxLoraLinear) else mod


class _BankFn(torch.autograd.Function):
    """A bank of same-shape AdaLN modulation linears applied to silu(temb): out[m] = lin_m(silu(temb)).

    The frozen base part -- the 13.6 GB weight stream at Qwen size -- runs on the HIP GEMVs in both directions (qfx_mod_gemv /
    qfx_mod_gemv_t: one pass over the weights each, vs one library GEMV launch per linear and direction), the rank-r side terms
    of the adapted members as two batched fp32 contractions; rounding points of the bf16 eager graph: base output rounded to bf16,
    adapter term added in fp32, sum rounded to bf16.  d(silu(temb)) through the base weights is accumulated in fp32 over the bank
    (autograd would sum one bf16 tensor per linear)."""

    @staticmethod
    def forward(ctx, temb, bank, *ab):
        from . import ops
        base = ops.mod_gemv_tables(temb.detach().contiguous(), bank.wt, bank.bt, len(bank.mods), bank.N, apply_silu=True)   # [nmat, B, N] bf16
        ctx.bank = bank
        if not bank.idx:
            ctx.save_for_backward(temb)
            return base
        s = F.silu(temb.detach())                                                                   # bf16, as the eager graph
        A = torch.stack(ab[0::2])                                                                   # [na, r, K] fp32
        Bm = torch.stack(ab[1::2])                                                                  # [na, N, r] fp32
        u = torch.einsum("bk,ark->abr", s.float(), A)
        add = torch.einsum("abr,anr->abn", u, Bm) * bank.scale.view(-1, 1, 1)
        out = base.clone()
        out[bank.idx_t] = (base[bank.idx_t].float() + add).to(base.dtype)
        ctx.save_for_backward(temb, s, u, A, Bm)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import ops
        bank = ctx.bank
        g = g.contiguous()
        temb = ctx.saved_tensors[0]
        ds = ops.mod_gemv_t(g, table=(bank.wt, bank.K))                                             # fp32 [B, K]
        grads = []
        if bank.idx:
            _, s, u, A, Bm = ctx.saved_tensors
            ga = g[bank.idx_t].float() * bank.scale.view(-1, 1, 1)                                  # [na, B, N]
            dB = torch.einsum("abn,abr->anr", ga, u)
            du = torch.einsum("abn,anr->abr", ga, Bm)
            dA = torch.einsum("abr,bk->ark", du, s.float())
            ds = ds + torch.einsum("abr,ark->bk", du, A)
            for i in range(len(bank.idx)):
                grads += [dA[i], dB[i]]
        t = temb.detach().float()
        sg = torch.sigmoid(t)
        dtemb = (ds.to(temb.dtype).float() * (sg * (1.0 + t * (1.0 - sg)))).to(temb.dtype)          # silu backward on the bf16 gradient
        return (dtemb, None, *grads)


class ModBank:
    """Pointer tables / adapter bookkeeping of a list of same-shape modulation linears (built once per plan)."""

    def __init__(self, mods):
        from . import ops
        self.mods = list(mods)
        bases = [_base(m) for m in self.mods]
        dev = bases[0].weight.device
        self.N, self.K = bases[0].weight.shape
        self.wt = ops.ptr_table([b.weight for b in bases], dev)
        self.bt = ops.ptr_table([b.bias for b in bases], dev)
        self._keep = [(b.weight, b.bias) for b in bases]
        self.idx = [i for i, m in enumerate(self.mods) if isinstance(m, QfxLoraLinear) and not m.merged]
        self.idx_t = torch.tensor(self.idx, dtype=torch.long, device=dev)
        self.scale = torch.tensor([self.mods[i].scaling[self.mods[i].active_adapter] for i in self.idx], dtype=torch.float32, device=dev)

    def __call__(self, temb):
        ab = []
        for i in self.idx:
            ab += [self.mods[i].A, self.mods[i].B]
        return _BankFn.apply(temb, self, *ab)


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


def _banks(model, key, build):
    """Banks are cached on the model per adapter state (add / merge / load of adapters bumps model._version)."""
    c = model.__dict__.setdefault("_cond_banks", {})
    k = (key, getattr(model, "_version", 0))
    if k not in c:
        for old in [kk for kk in c if kk[0] == key]:
            del c[old]
        c[k] = build()
    return c[k]


def qwen_head(model, tproj: torch.Tensor):
    """QwenTimestepProjEmbeddings + every block's img_mod / txt_mod + norm_out.linear.  tproj: bf16 sinusoid [B, 256]."""
    te = model.time_text_embed.timestep_embedder
    temb = lin(te.linear_2, F.silu(lin(te.linear_1, tproj)))
    blocks, out = _banks(model, "qwen", lambda: (
        ModBank([m for blk in model.transformer_blocks for m in (blk.img_mod[1], blk.txt_mod[1])]), ModBank([model.norm_out.linear])))
    return [blocks(temb), out(temb)[0]]


def flux_head(model, tproj, gproj, pooled):
    """CombinedTimestep(Guidance)TextProjEmbeddings + norm1 / norm1_context / single-block norm linears + norm_out.linear."""
    te = model.time_text_embed
    temb = lin(te.timestep_embedder.linear_2, F.silu(lin(te.timestep_embedder.linear_1, tproj)))
    if gproj is not None:
        temb = temb + lin(te.guidance_embedder.linear_2, F.silu(lin(te.guidance_embedder.linear_1, gproj)))
    temb = temb + lin(te.text_embedder.linear_2, F.silu(lin(te.text_embedder.linear_1, pooled)))
    def build():
        dbl = [m for blk in model.transformer_blocks for m in (blk.norm1.linear, blk.norm1_context.linear)]
        sgl = [blk.norm.linear for blk in model.single_transformer_blocks]
        return (ModBank(dbl) if dbl else None, ModBank(sgl) if sgl else None, ModBank([model.norm_out.linear]))

    dbl, sgl, out = _banks(model, "flux", build)
    return [dbl(temb) if dbl is not None else None, sgl(temb) if sgl is not None else None, out(temb)[0]]
