"""CPU oracle of the Prodigy optimizer step (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

The reference selects ``prodigyopt.Prodigy`` through its optimizer config (``configs/face_seg_flux_kontext_fp16_prodigy.yaml:41-47``,
every ``tests/test_configs/test_example_*_fp16.yaml`` optimizer section, ``docs/guide/configuration.md:190-197``:
``lr: 1.0, use_bias_correction: true, safeguard_warmup: true, weight_decay: 0.01``) and instantiates it generically in
``src/qflux/trainer/base_trainer.py:884-898``.  ``prodigyopt`` is a third-party dependency (``requirements.txt:33``, unpinned) that
is NOT installable in the build container, so this file restates its published algorithm -- Mishchenko & Defazio, "Prodigy: An
Expeditiously Adaptive Parameter-Free Learner" (Adam variant), as implemented by ``prodigyopt`` 1.x ``Prodigy.step`` -- and
**parity is unpinned** for it (no golden vectors exist offline; the hand-computed first-step identities in
``tests/test_prodigy_cpu.py`` pin the formulas, not the package).

Semantics restated (one param group, ``slice_p = 1``, ``fsdp_in_use = False``):
  * host scalars (``d``, ``d_max``, ``d_numerator``, ``d_denom``) are Python floats (float64); per-tensor dot products and abs-sums
    are fp32 reductions whose ``.item()`` values are accumulated in float64;
  * ``dlr = d * lr * bias_correction`` uses the d of the PREVIOUS step for the whole step; the Adam denominator uses the NEW d;
  * ``lr == 0`` (first step of a warm-up schedule): states are created (``p0`` captured) and the call returns before touching
    anything else -- ``k`` does not advance;
  * an all-zero parameter keeps ``p0 = 0`` (LoRA B at init) -- identical arithmetic to a zero tensor.
"""
from __future__ import annotations

import math

import torch


class Prodigy:
    def __init__(self, params, lr=1.0, betas=(0.9, 0.999), beta3=None, eps=1e-8, weight_decay=0.0, decouple=True,
                 use_bias_correction=False, safeguard_warmup=False, d0=1e-6, d_coef=1.0, growth_rate=float("inf")):
        self.params = list(params)
        self.group = dict(lr=lr, betas=betas, beta3=beta3, eps=eps, weight_decay=weight_decay, decouple=decouple,
                          use_bias_correction=use_bias_correction, safeguard_warmup=safeguard_warmup, d=d0, d0=d0, d_max=d0,
                          d_numerator=0.0, d_denom=0.0, d_hat=d0, d_coef=d_coef, growth_rate=growth_rate, k=0)
        self.state = [dict() for _ in self.params]

    @torch.no_grad()
    def step(self, grads):
        """grads: list of fp32 tensors (already clipped), one per parameter."""
        g_ = self.group
        beta1, beta2 = g_["betas"]
        beta3 = g_["beta3"] if g_["beta3"] is not None else math.sqrt(beta2)
        k, d, d_max, d_coef, lr, d0 = g_["k"], g_["d"], g_["d_max"], g_["d_coef"], g_["lr"], g_["d0"]
        bias_correction = ((1 - beta2 ** (k + 1)) ** 0.5) / (1 - beta1 ** (k + 1)) if g_["use_bias_correction"] else 1.0
        dlr = d * lr * bias_correction
        decay, eps = g_["weight_decay"], g_["eps"]
        d_numerator = g_["d_numerator"] * beta3
        d_denom = 0.0
        for p, grad, st in zip(self.params, grads, self.state):
            if decay != 0 and not g_["decouple"]:
                grad = grad + decay * p
            if "step" not in st:
                st["step"] = 0
                st["s"] = torch.zeros_like(p).flatten()
                st["p0"] = p.flatten().clone()
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            if lr > 0.0:
                sg = grad.flatten()
                d_numerator += (d / d0) * dlr * torch.dot(sg, st["p0"] - p.flatten()).item()
                st["exp_avg"].mul_(beta1).add_(grad, alpha=d * (1 - beta1))
                st["exp_avg_sq"].mul_(beta2).addcmul_(grad, grad, value=d * d * (1 - beta2))
                st["s"].mul_(beta3).add_(sg, alpha=((d / d0) * d) if g_["safeguard_warmup"] else ((d / d0) * dlr))
                d_denom += st["s"].abs().sum().item()
        d_hat = d
        if d_denom == 0:
            return
        if lr > 0.0:
            d_hat = d_coef * d_numerator / d_denom
            if d == d0:
                d = max(d, d_hat)
            d_max = max(d_max, d_hat)
            d = min(d_max, d * g_["growth_rate"])
        g_.update(d_numerator=d_numerator, d_denom=d_denom, d=d, d_max=d_max, d_hat=d_hat)
        for p, st in zip(self.params, self.state):
            st["step"] += 1
            denom = st["exp_avg_sq"].sqrt().add_(d * eps)
            if decay != 0 and g_["decouple"]:
                p.add_(p, alpha=-decay * dlr)
            p.addcdiv_(st["exp_avg"], denom, value=-dlr)
        g_["k"] = k + 1


def clip_grad_norm_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ on a list of gradients (base_trainer.py:449-455); returns the clipped copies."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads]
