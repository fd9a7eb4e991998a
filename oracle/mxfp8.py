"""TEST INFRASTRUCTURE (CPU oracle; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

fp8-emulating restatement of the low-precision trunk (reference: src/qflux/models/quantize.py replaces nn.Linear by
TE fp8 / bnb int8 / NF4 linears -- third-party engines, absent offline).  What is emulated here is what the MI355X path
implements: OCP MX-FP8 operands (Microscaling Formats specification v1.0: element type e4m3, one E8M0 scale per 32 consecutive
elements along K, scale exponent = floor(log2(max|v|)) - emax(e4m3) with emax = 8, elements = RNE(v / 2^e) saturated to +-448) on
BOTH operands of the base linear in the forward pass; bias, bf16 output rounding and the LoRA branch unchanged; the backward uses
the un-quantised bf16 operands (dX = dY W, adapters on the bf16 activations) unless backward=True ("mxfp8-fb": dY and W^T quantised
along out_features).  parity unpinned against the reference's engines.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def mx_qdq(x: torch.Tensor) -> torch.Tensor:
    """quantise -> dequantise along the last dim in blocks of 32 (fp32 result)."""
    shp = x.shape
    v = x.float().reshape(-1, shp[-1] // 32, 32)
    amax = v.abs().amax(-1, keepdim=True)
    e = torch.where(amax > 0, torch.floor(torch.log2(amax.clamp_min(1e-38))) - 8, torch.full_like(amax, -127.0)).clamp(-127, 127)
    sc = torch.pow(2.0, e)
    q = (v / sc).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    return (q * sc).reshape(shp)


class _QLinearFn(torch.autograd.Function):
    """qb: "mxfp8-fb" -- the dX GEMM contracts MX-FP8 operands too (dY and W^T quantised along out_features)."""

    @staticmethod
    def forward(ctx, x, w, b, qb=False):
        ctx.save_for_backward(x, w)
        ctx.qb = qb
        y = F.linear(mx_qdq(x), mx_qdq(w)).to(torch.float32)
        if b is not None:
            y = y + b.float()
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        if ctx.qb and w.shape[0] % 128 == 0 and w.shape[0] >= 1024 and w.shape[1] >= 1024:
            dx = (mx_qdq(dy) @ mx_qdq(w.t().contiguous()).t()).to(dy.dtype)
        else:
            dx = dy.to(w.dtype) @ w
        return dx, None, None, None


_COND_HEAD = ("_mod.", "time_text_embed.", "norm_out", "norm1.linear", "norm1_context.linear", "norm.linear")


def eligible(name: str, lin: nn.Linear) -> bool:
    """What the MI355X path quantises: the many-row GEMM sites (K % 128 == 0, K and N >= 1024); the conditioning head (M = batch
    rows: modulation / embedder linears, evaluated as GEMVs) stays bf16."""
    if any(t in name for t in _COND_HEAD):
        return False
    return lin.in_features % 128 == 0 and lin.in_features >= 1024 and lin.out_features >= 1024


def quantize_oracle(model: nn.Module, predicate=eligible, backward: bool = False):
    """Patch the forward of every eligible frozen nn.Linear (the base layers of adapted linears included).  backward=True: the dX
    GEMMs contract MX-FP8 operands as well -- double-stream blocks and FLUX single blocks alike (there the MI355X path sums the
    q/k/v/proj_mlp products in one contraction over the concatenated K: same operand bytes, fp32 instead of bf16 partial sums)."""
    n = 0
    for name, m in model.named_modules():
        if isinstance(m, nn.Linear) and "lora_" not in name and predicate(name, m):
            qb = bool(backward)
            m.forward = (lambda x, m=m, qb=qb: _QLinearFn.apply(x, m.weight, m.bias, qb))
            n += 1
    return n
