"""Conditioning head with adapters on its linears -- emitted as libqfx launches into the plans' launch programs.

The head (timestep / guidance / pooled-text embedders, every AdaLN modulation linear, norm_out.linear) sees M = batch rows;
without adapters it is a handful of qfx_mod_gemv launches.  With adapters on any of its linears -- target_modules "all-linear"
(configs/example_with_sampling.yaml:9) or the `(norm|norm1|norm1_context).linear` alternatives of
configs/face_seg_flux_kontext_fp16.yaml:11 -- every linear becomes peft's

    y = base(x);  y = (y + lora_B(lora_A(x.float())) * scaling).to(bf16)            (peft lora.Linear.forward)

and the backward of the head (fed by the d(modulation) column sums of the HIP backward, qfx_mod_grad) runs through the same
graph: frozen base weights on the GEMV streams in both directions (qfx_mod_gemv / qfx_mod_gemv_t, one pass over the 13.6 GB of
modulation weights each -- the second only when an embedder adapter needs d temb), the rank-r terms of a BANK of same-shape
linears in one qfx_cond_lora_fwd / qfx_cond_lora_bwd launch pair, silu backward in qfx_silu_bwd.  Adapter gradients accumulate
straight into the flat LoRA gradient buffer.  Rounding points follow the reference's bf16 eager graph
(transformer_qwenimage.py:143-156,430-436,565,664; transformer_flux.py:634-639,729-741): base output rounded to bf16, adapter
term added in fp32, sum rounded to bf16; gradients of the head's activations are bf16 tensors (the fp32 column sums are rounded
once on entry, d(silu(temb)) is accumulated in fp32 over the bank where autograd would sum one bf16 tensor per linear).

Round 2 evaluated this head with torch (F.linear / einsum / autograd); nothing here computes with torch -- it allocates buffers
and zero-fills accumulators (memsets), like the rest of the launch programs, which also makes the head capturable in a hipGraph.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .modules import QfxLoraLinear

lib = L.lib
BF = torch.bfloat16
F32 = torch.float32


def _p(t):
    return None if t is None else t.data_ptr()


def _base(mod):
    return mod.base_layer if isinstance(mod, QfxLoraLinear) else mod


class CondBank:
    """A list of same-shape linears applied to ONE input (a single embedder linear, or every modulation linear of the model):
    device pointer tables of the frozen weights and of the adapted members' A / B / gradient slices."""

    def __init__(self, model, mods, B):
        self.mods = list(mods)
        bases = [_base(m) for m in self.mods]
        dev = bases[0].weight.device
        self.N, self.K = bases[0].weight.shape
        self.nmat, self.B = len(bases), B
        tbl = lambda ptrs: torch.tensor(ptrs, dtype=torch.int64, device=dev)      # noqa: E731
        self.wt = tbl([b.weight.data_ptr() for b in bases])
        self.bt = tbl([b.bias.data_ptr() for b in bases])
        self._keep = [(b.weight, b.bias) for b in bases]
        self.idx = [i for i, m in enumerate(self.mods) if isinstance(m, QfxLoraLinear) and not m.merged]
        self.na = len(self.idx)
        if not self.na:
            return
        ad = [self.mods[i] for i in self.idx]
        rs = {m.r[m.active_adapter] for m in ad}
        if len(rs) != 1:
            raise NotImplementedError("adapters of different rank in one bank of conditioning-head linears")
        self.r = rs.pop()
        if self.r > 64 or B > 8:
            raise NotImplementedError("conditioning-head adapters: rank <= 64 and per-GPU batch <= 8")
        st = model.lora_store
        gbase = st.gflat.data_ptr()
        self.A = tbl([m.A.data_ptr() for m in ad])
        self.Bm = tbl([m.B.data_ptr() for m in ad])
        self.dA = tbl([gbase + 4 * st.offset_of(m.A) for m in ad])
        self.dB = tbl([gbase + 4 * st.offset_of(m.B) for m in ad])
        self.scale = torch.tensor([m.scaling[m.active_adapter] for m in ad], dtype=F32, device=dev)
        self.u = torch.zeros(self.na, B, self.r, dtype=F32, device=dev)
        self.du = torch.zeros(self.na, B, self.r, dtype=F32, device=dev)

    def _args(self, x, silu):
        a = L.CondLoraArgs()
        a.x, a.B, a.K, a.apply_silu, a.na, a.r, a.N = _p(x), self.B, self.K, int(silu), self.na, self.r, self.N
        a.A, a.Bm, a.scale, a.u, a.du = _p(self.A), _p(self.Bm), _p(self.scale), _p(self.u), _p(self.du)
        a.dA, a.dB = _p(self.dA), _p(self.dB)
        return a

    def emit_fwd(self, p, x, silu, out):
        """out [nmat, B, N] bf16 = lin_m(act(x)) for every member."""
        assert out.shape[-1] == self.N and out.is_contiguous()
        p.c(lib.qfx_mod_gemv, _p(x), self.B, self.K, _p(self.wt), _p(self.bt), self.nmat, self.N, int(silu), _p(out))
        if self.na:
            a = self._args(x, silu)
            rows = out.view(-1, self.B, self.N)
            ytab = torch.tensor([rows[i].data_ptr() for i in self.idx], dtype=torch.int64, device=out.device)
            a.y, a.ldy = _p(ytab), self.N
            p.keep.append((a, ytab))
            p.c(lib.qfx_cond_lora_fwd, C.byref(a))

    def emit_bwd(self, p, x, silu, g, ds):
        """g [nmat, B, N] bf16 = gradients of the outputs; ds (fp32 [B, K], accumulated) = gradient w.r.t. act(x), or None when
        nothing upstream needs it (then only the adapters' own gradients are produced)."""
        if ds is not None:
            p.c(lib.qfx_mod_gemv_t, _p(g), self.B, self.N, self.K, _p(self.wt), self.nmat, _p(ds))
        if self.na:
            a = self._args(x, silu)
            rows = g.view(-1, self.B, self.N)
            gtab = torch.tensor([rows[i].data_ptr() for i in self.idx], dtype=torch.int64, device=g.device)
            a.g, a.ldg, a.dx = _p(gtab), self.N, _p(ds)
            p.keep.append((a, gtab))
            p.py(self.du.zero_)
            p.c(lib.qfx_cond_lora_bwd, C.byref(a))


class CondHeadHip:
    """The conditioning head of one plan.  `chains`: [(first linear, second linear, input buffer, hidden buffer, output buffer)]
    = the embedders (x -> lin1 -> silu -> lin2), summed into temb when there are several (FLUX); `banks`: [(modules, output
    buffer [nmat,B,N], fp32 gradient buffer)] = the modulation linears applied to silu(temb)."""

    def __init__(self, model, B, D, chains, temb, banks, buf):
        self.B, self.D, self.temb = B, D, temb
        self.chains = [(CondBank(model, [l1], B), CondBank(model, [l2], B), x, h, y) for l1, l2, x, h, y in chains]
        self.banks = [(CondBank(model, mods, B), out, grad) for mods, out, grad in banks]
        # d temb is needed only when an embedder carries an adapter (autograd would prune the rest the same way)
        self.need_dtemb = any(c[0].na or c[1].na for c in self.chains)
        self.gb = [buf(*grad.shape) for _, _, grad in self.banks]          # bf16 images of the fp32 column sums
        if self.need_dtemb:
            self.ds = buf(B, D, dtype=F32, zero=True)
            self.ds2 = buf(B, D, dtype=F32, zero=True)
            self.dtemb = buf(1, B, D)
            self.dh = buf(1, B, D)

    def emit_forward(self, p):
        B, D = self.B, self.D
        for l1, l2, x, h, y in self.chains:
            l1.emit_fwd(p, x, 0, h.view(1, B, D))
            l2.emit_fwd(p, h, 1, y.view(1, B, D))
        ys = [c[4] for c in self.chains]
        if len(ys) > 1:       # bf16(bf16(a + b) + c): CombinedTimestep(Guidance)TextProjEmbeddings (transformer_flux.py:731-735)
            p.c(lib.qfx_add3_bf16, _p(ys[0]), _p(ys[1]), _p(ys[2]) if len(ys) > 2 else None, _p(self.temb), B * D)
        for bank, out, _ in self.banks:
            bank.emit_fwd(p, self.temb, 1, out)

    def emit_backward(self, p):
        """Reads the fp32 gradient buffers of the banks (filled by the blocks' qfx_mod_grad launches); LAST entries of the
        backward program (data-parallel: these adapters' gradients are final only here, trainer._bucket_hook)."""
        B, D = self.B, self.D
        if self.need_dtemb:
            p.py(self.ds.zero_)
        for (bank, _, grad), gb in zip(self.banks, self.gb):
            if not (bank.na or self.need_dtemb):
                continue
            p.c(lib.qfx_cast_f32_bf16, _p(grad), _p(gb), grad.numel())
            bank.emit_bwd(p, self.temb, 1, gb, self.ds if self.need_dtemb else None)
        if not self.need_dtemb:
            return
        p.c(lib.qfx_silu_bwd, _p(self.ds), _p(self.temb), _p(self.dtemb), B * D)       # d temb (= d of every summand of temb)
        for l1, l2, x, h, y in self.chains:
            if not (l1.na or l2.na):
                continue
            need_dh = bool(l1.na)
            if need_dh:
                p.py(self.ds2.zero_)
            l2.emit_bwd(p, h, 1, self.dtemb, self.ds2 if need_dh else None)
            if need_dh:
                p.c(lib.qfx_silu_bwd, _p(self.ds2), _p(h), _p(self.dh), B * D)
                l1.emit_bwd(p, x, 0, self.dh, None)      # the embedders' inputs (sinusoid / pooled text) carry no gradient
