"""Prodigy oracle (oracle/prodigy.py) against hand-derived identities of the published algorithm.  prodigyopt itself is not installable
offline, so these pin the formulas the restatement claims (first-step closed forms, the d-estimate recursion, the lr = 0 early return,
decoupled decay), not the package -- the oracle header says "parity unpinned" for it."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.prodigy import Prodigy, clip_grad_norm_  # noqa: E402


def test_first_two_steps_closed_form():
    d0, b1, b2 = 1e-6, 0.9, 0.999
    b3 = math.sqrt(b2)
    p = torch.tensor([0.5, -1.0, 2.0]); p_init = p.clone()
    g1 = torch.tensor([1.0, -2.0, 0.5]); g2 = torch.tensor([0.5, -1.0, 1.0])
    opt = Prodigy([p], lr=1.0, betas=(b1, b2), eps=1e-8)
    opt.step([g1])
    # step 1: p0 == p -> numerator 0, d_hat 0, d stays d0;  s = (d/d0) dlr g = d0 g;  m = d0 (1-b1) g;  v = d0^2 (1-b2) g^2
    G = opt.group
    assert G["d"] == d0 and G["d_max"] == d0 and G["d_hat"] == 0.0 and G["k"] == 1
    assert abs(G["d_denom"] - d0 * g1.abs().sum().item()) < 1e-12
    st = opt.state[0]
    assert torch.allclose(st["s"], d0 * g1, rtol=1e-6, atol=0)
    assert torch.allclose(st["exp_avg"], d0 * (1 - b1) * g1, rtol=1e-6, atol=0)
    want = p_init - d0 * (d0 * (1 - b1) * g1) / ((d0 * d0 * (1 - b2) * g1 * g1).sqrt() + d0 * 1e-8)
    assert torch.allclose(p, want, rtol=0, atol=1e-12)
    # step 2: numerator = (d/d0) dlr <g2, p0 - p1>; denominator = |b3 s1 + d0 g2|_1; d_hat = num / den; d0 == d -> d = max(d, d_hat)
    delta = (p_init - p).double()
    num = d0 * float((g2.double() * delta).sum())
    den = float((b3 * d0 * g1.double() + d0 * g2.double()).abs().sum())
    opt.step([g2])
    assert abs(G["d_hat"] - num / den) / (num / den) < 1e-5
    assert G["d"] == max(d0, G["d_hat"]) and G["d_max"] == G["d"] and G["k"] == 2


def test_lr_zero_is_an_early_return_and_p0_is_captured():
    p = torch.tensor([1.0, 2.0]); opt = Prodigy([p], lr=0.0)
    opt.step([torch.tensor([1.0, 1.0])])
    assert opt.group["k"] == 0 and torch.equal(p, torch.tensor([1.0, 2.0])) and torch.equal(opt.state[0]["p0"], p)
    assert torch.count_nonzero(opt.state[0]["exp_avg"]) == 0 and torch.count_nonzero(opt.state[0]["s"]) == 0


def test_distance_estimate_grows_and_quadratic_converges():
    torch.manual_seed(0)
    tgt = torch.randn(64)
    p = torch.zeros(64)
    opt = Prodigy([p], lr=1.0, use_bias_correction=True, safeguard_warmup=True, weight_decay=0.0)
    ds = []
    for _ in range(400):
        opt.step([p - tgt])
        ds.append(opt.group["d"])
    assert all(b >= a for a, b in zip(ds, ds[1:]))          # growth_rate = inf: d is the running max of d_hat
    assert ds[-1] > 1e3 * ds[0]
    assert (p - tgt).norm() < 0.05 * tgt.norm()


def test_decoupled_decay_and_clip():
    p = torch.tensor([1.0, -1.0]); g = torch.tensor([3.0, 4.0])
    (gc,) = clip_grad_norm_([g], 1.0)
    assert torch.allclose(gc, g / (5.0 + 1e-6))
    a, b = p.clone(), p.clone()
    oa, ob = Prodigy([a], weight_decay=0.0), Prodigy([b], weight_decay=0.5)
    oa.step([gc]); ob.step([gc])
    dlr = 1e-6
    assert torch.allclose(b, a - 0.5 * dlr * p, rtol=0, atol=1e-12)
