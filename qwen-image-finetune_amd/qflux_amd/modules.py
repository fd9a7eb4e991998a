"""Parameter-holder modules with the reference's state-dict names and the
LoRA parameter store.  These modules own tensors only; all arithmetic is issued by
qflux_amd.plan as calls into libqfx (no nn.Linear / autograd compute anywhere on the hot path).

LoRA (peft semantics, src/qflux/trainer/base_trainer.py:929-941): wrapping a Linear `X` yields
`X.base_layer.{weight,bias}`, `X.lora_A.<adapter>.weight` [r,in] fp32, `X.lora_B.<adapter>.weight`
[out,r] fp32.  All adapter parameters are views into ONE flat fp32 buffer (and their .grad into one
flat gradient buffer), so the optimizer step is one fused kernel and the data-parallel exchange is
one RCCL all-reduce of exactly the LoRA gradients.
"""
from __future__ import annotations

import math
import re

import torch
import torch.nn as nn


class QfxLinear(nn.Module):
    """Holder for an nn.Linear's parameters (weight [out,in], bias [out])."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, dtype=torch.bfloat16):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, dtype=dtype), requires_grad=False) if bias else None

    def forward(self, *a, **k):
        raise RuntimeError("QfxLinear is a parameter holder; compute is issued through libqfx by the owning model")


class _W(nn.Module):
    """Holder of a single `.weight` (lora_A.<adapter> / lora_B.<adapter> / RMSNorm)."""

    def __init__(self, weight: torch.Tensor, requires_grad: bool):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=requires_grad)


class QfxRMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6, dtype=torch.bfloat16):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype), requires_grad=False)


class QfxLoraLinear(nn.Module):
    """peft lora.Linear look-alike (names only): base_layer + lora_A/lora_B ModuleDicts."""

    def __init__(self, base: QfxLinear, r: int, lora_alpha: float, adapter_name: str):
        super().__init__()
        self.base_layer = base
        self.in_features, self.out_features = base.in_features, base.out_features
        self.r = {adapter_name: r}
        self.lora_alpha = {adapter_name: lora_alpha}
        self.scaling = {adapter_name: float(lora_alpha) / float(r)}
        self.active_adapter = adapter_name
        self.merged = False
        dev = base.weight.device
        self.lora_A = nn.ModuleDict({adapter_name: _W(torch.zeros(r, base.in_features, dtype=torch.float32, device=dev), True)})
        self.lora_B = nn.ModuleDict({adapter_name: _W(torch.zeros(base.out_features, r, dtype=torch.float32, device=dev), True)})

    @property
    def A(self) -> nn.Parameter:
        return self.lora_A[self.active_adapter].weight

    @property
    def B(self) -> nn.Parameter:
        return self.lora_B[self.active_adapter].weight

    def get_delta_weight(self) -> torch.Tensor:
        """peft lora.Linear.get_delta_weight: (B @ A) * scaling in the adapter dtype (fp32)."""
        return (self.B.detach() @ self.A.detach()) * self.scaling[self.active_adapter]

    def merge(self):
        """peft LoraLayer.merge (safe_merge=False): `base_layer.weight.data += delta` -- an fp32 delta added in place to the bf16
        weight (computed in fp32, rounded once); afterwards the forward is the base layer alone."""
        if not self.merged:
            self.base_layer.weight.data += self.get_delta_weight().to(self.base_layer.weight.device)
            self.merged = True

    def unmerge(self):
        if self.merged:
            self.base_layer.weight.data -= self.get_delta_weight().to(self.base_layer.weight.device)
            self.merged = False

    def forward(self, *a, **k):
        raise RuntimeError("QfxLoraLinear is a parameter holder")


class LoraConfig:
    """Minimal stand-in for peft.LoraConfig (fields the reference sets, base_trainer.py:932-937)."""

    def __init__(self, r=16, lora_alpha=16, init_lora_weights="gaussian", target_modules=("to_k", "to_q", "to_v", "to_out.0"), **_):
        self.r, self.lora_alpha = int(r), float(lora_alpha)
        self.init_lora_weights = init_lora_weights
        self.target_modules = target_modules
        # the fields peft.utils.get_peft_model_state_dict reads off `model.peft_config[adapter_name]` (peft is duck-typed on them)
        self.peft_type, self.task_type = "LORA", None
        self.bias, self.use_dora, self.is_prompt_learning = "none", False, False
        self.modules_to_save = None

    def __repr__(self):
        return (f"LoraConfig(r={self.r}, lora_alpha={self.lora_alpha}, init_lora_weights={self.init_lora_weights!r}, "
                f"target_modules={self.target_modules!r})")


def match_target(name: str, target_modules) -> bool:
    """peft matching: a str is a full-match regex (or 'all-linear'), a list is suffix-matched."""
    if isinstance(target_modules, str):
        if target_modules == "all-linear":
            return True
        return re.fullmatch(target_modules, name) is not None
    return any(name == t or name.endswith("." + t) for t in target_modules)


class LoraStore:
    """Flat fp32 parameter / gradient buffers for every adapter weight of a model."""

    def __init__(self, model: nn.Module):
        self.model = model
        self.pflat: torch.Tensor | None = None
        self.gflat: torch.Tensor | None = None
        self.entries: list[tuple[str, nn.Parameter, int, int]] = []  # (name, param, offset, numel)

    def params(self):
        return [(n, p) for n, p in self.model.named_parameters() if "lora_" in n]

    def rebuild(self, device=None) -> None:
        """(Re)pack all adapter params into one flat buffer on `device`, keeping values and Parameter identity."""
        ps = self.params()
        if not ps:
            self.pflat = self.gflat = None
            self.entries = []
            return
        device = device or ps[0][1].device
        total = sum((p.numel() + 63) // 64 * 64 for _, p in ps)
        pflat = torch.zeros(total, dtype=torch.float32, device=device)
        gflat = torch.zeros(total, dtype=torch.float32, device=device)
        entries, off = [], 0
        for n, p in ps:
            k = p.numel()
            view = pflat[off:off + k].view(p.shape)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            old_grad = p.grad
            p.data = view
            p.grad = gflat[off:off + k].view(p.shape)
            if old_grad is not None:
                p.grad.copy_(old_grad.to(device))
            entries.append((n, p, off, k))
            off += (k + 63) // 64 * 64
        self.pflat, self.gflat, self.entries = pflat, gflat, entries

    def is_consistent(self, device) -> bool:
        if self.pflat is None:
            return not self.params()
        if self.pflat.device != torch.device(device) or len(self.entries) != len(self.params()):
            return False
        base = self.pflat.data_ptr()
        return all(p.data_ptr() == base + 4 * off for _, p, off, _ in self.entries)

    def ensure_grads(self) -> None:
        """optimizer.zero_grad(set_to_none=True) drops our aliased .grad views: re-attach them (zeroed)."""
        if self.gflat is None:
            return
        gbase = self.gflat.data_ptr()
        dropped = False
        for _, p, off, k in self.entries:
            if p.grad is None or p.grad.data_ptr() != gbase + 4 * off:
                dropped = True
                break
        if dropped:
            keep = [(p.grad.clone() if p.grad is not None else None) for _, p, _, _ in self.entries]
            self.gflat.zero_()
            for (_, p, off, k), g in zip(self.entries, keep):
                p.grad = self.gflat[off:off + k].view(p.shape)
                if g is not None and g.data_ptr() != p.grad.data_ptr():
                    p.grad.copy_(g)

    def offset_of(self, param: nn.Parameter) -> int:
        for _, p, off, _ in self.entries:
            if p is param:
                return off
        raise KeyError("parameter not in LoRA store")


def init_lora_(mod: QfxLoraLinear, init: str, generator: torch.Generator | None = None) -> None:
    """peft init: lora_B = 0; lora_A ~ N(0, (1/r)^2) for 'gaussian', kaiming-uniform(a=sqrt(5)) otherwise."""
    name = mod.active_adapter
    r = mod.r[name]
    with torch.no_grad():
        a = mod.A
        if init == "gaussian":
            a.copy_((torch.randn(a.shape, generator=generator) / r).to(a.device))
        else:
            bound = 1.0 / math.sqrt(a.shape[1])  # kaiming_uniform(a=sqrt(5)) on [r, in] == U(-1/sqrt(in), 1/sqrt(in))
            a.copy_(((torch.rand(a.shape, generator=generator) * 2 - 1) * bound).to(a.device))
        mod.B.zero_()
