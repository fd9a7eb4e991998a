"""Fused Prodigy step (qfx_prodigy_step) vs the CPU oracle (oracle/prodigy.py): same gradient sequence, several parameter tensors
packed in one flat buffer, global-norm clip, decoupled decay, bias correction, safeguard warm-up, an lr = 0 first call.
Tolerance: 2e-5 relative on the elementwise states per step (both sides re-synchronised after every step), on the d estimate 1e-5 relative plus
32 ulp of the ABSOLUTE size of its numerator's cancelling fp32 sum <g, p0 - p> (flat-buffer order here, per-tensor torch.dot there)."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("cfg", [dict(use_bias_correction=True, safeguard_warmup=True, weight_decay=0.01),
                                 dict(use_bias_correction=False, safeguard_warmup=False, weight_decay=0.0),
                                 dict(use_bias_correction=True, safeguard_warmup=False, weight_decay=0.1, beta3=0.9, d_coef=2.0,
                                      growth_rate=1.5)])
def test_prodigy_step_matches_oracle(cfg):
    from qflux_amd import ops
    from qflux_amd._lib import PRODIGY_STATE
    from oracle.prodigy import Prodigy, clip_grad_norm_
    torch.manual_seed(3)
    shapes = [(16, 96), (96, 16), (8, 40), (1000,)]
    params = [torch.randn(s) * 0.1 for s in shapes]
    params[1].zero_()                                     # LoRA B at init: the package keeps a zero p0 for it
    tgt = [torch.randn(s) for s in shapes]
    n = sum(p.numel() for p in params)
    flat = torch.cat([p.flatten() for p in params]).to(DEV)
    m, v, s = (torch.zeros(n, device=DEV) for _ in range(3))
    p0 = flat.clone()
    state = torch.zeros(PRODIGY_STATE, dtype=torch.float64, device=DEV)
    ops.prodigy_init_state(state, 1e-6)
    gn = torch.zeros((), device=DEV)
    okw = {k: cfg[k] for k in cfg if k != "weight_decay"}
    opt = Prodigy(params, lr=1.0, weight_decay=cfg["weight_decay"], **okw)
    lrs = [0.0, 0.5] + [1.0] * 88                          # warm-up style: the first call is the package's early return
    for it, lr in enumerate(lrs):
        # gradient of a noisy quadratic, evaluated at the ORACLE's iterate so both sides see the same sequence
        grads = [(p - t) * 3.0 + 0.3 * torch.randn_like(p) for p, t in zip(params, tgt)]
        gflat = torch.cat([g.flatten() for g in grads]).to(DEV)
        opt.group["lr"] = lr
        gc = clip_grad_norm_(grads, 1.0)
        G = opt.group
        # size of the cancelling sum behind this step's numerator increment: (d/d0) dlr sum_i |g_i (p0 - p)_i|
        b1_, b2_ = G["betas"]
        bc = math.sqrt(1 - b2_ ** (G["k"] + 1)) / (1 - b1_ ** (G["k"] + 1)) if G["use_bias_correction"] else 1.0
        amp = (G["d"] / G["d0"]) * G["d"] * lr * bc
        mag = sum(float((g_.flatten() * (st["p0"] - p_.flatten())).abs().sum()) for g_, p_, st in zip(gc, params, opt.state)
                  if "p0" in st)
        opt.step(gc)
        gn.zero_(); ops.sumsq(gflat, gn)
        ops.prodigy_step(flat, gflat, m, v, s, p0, state, lr=lr, weight_decay=cfg["weight_decay"], gnorm_sq=gn, max_norm=1.0, **okw)
        hs = state.cpu().tolist()
        assert int(hs[5]) == G["k"], (it, hs[5], G["k"])
        if G["k"] == 0:
            assert torch.equal(flat.cpu(), torch.cat([p.flatten() for p in params]))
            continue
        # <g, p0 - p> is a cancelling fp32 sum (flat buffer here, per-tensor torch.dot there): 32 ulp of its absolute size
        tol_num = 1e-5 * abs(G["d_numerator"]) + 4e-6 * amp * mag
        tol_d = G["d_coef"] * tol_num / G["d_denom"] + 1e-5 * abs(G["d_hat"])
        for i, (key, tol) in enumerate((("d", tol_d), ("d_max", tol_d), ("d_numerator", tol_num), ("d_denom", 1e-5 * G["d_denom"]),
                                        ("d_hat", tol_d))):
            assert abs(hs[i] - G[key]) <= tol, (it, key, hs[i], G[key], tol)
        want = torch.cat([p.flatten() for p in params])
        assert _rel(flat.cpu(), want) < 2e-5, (it, _rel(flat.cpu(), want))
        assert _rel(m.cpu(), torch.cat([st["exp_avg"].flatten() for st in opt.state])) < 2e-5
        assert _rel(v.cpu(), torch.cat([st["exp_avg_sq"].flatten() for st in opt.state])) < 2e-5
        assert _rel(s.cpu(), torch.cat([st["s"] for st in opt.state])) < 2e-5
        # keep the two trajectories glued (the comparison is per step, not of accumulated drift)
        flat.copy_(want.to(DEV))
        m.copy_(torch.cat([st["exp_avg"].flatten() for st in opt.state]).to(DEV))
        v.copy_(torch.cat([st["exp_avg_sq"].flatten() for st in opt.state]).to(DEV))
        s.copy_(torch.cat([st["s"] for st in opt.state]).to(DEV))
        state[:5] = torch.tensor([G[k_] for k_ in ("d", "d_max", "d_numerator", "d_denom", "d_hat")], dtype=torch.float64).to(DEV)
    assert opt.group["d"] > 1e-2                          # the estimate left d0 = 1e-6 far behind (both regimes were compared)


def test_prodigy_zero_gradient_is_a_no_op_and_bad_args():
    from qflux_amd import ops
    from qflux_amd._lib import PRODIGY_STATE
    n = 4096
    flat = torch.randn(n, device=DEV); before = flat.clone()
    m, v, s, g = (torch.zeros(n, device=DEV) for _ in range(4))
    state = torch.zeros(PRODIGY_STATE, dtype=torch.float64, device=DEV)
    ops.prodigy_init_state(state, 1e-6)
    ops.prodigy_step(flat, g, m, v, s, before, state, weight_decay=0.5)
    assert torch.equal(flat, before) and state[5].item() == 0.0      # d_denom == 0 -> the package returns before any update
    with pytest.raises(RuntimeError):
        ops.prodigy_step(flat, g, m, v, s, before, state, weight_decay=0.5, decouple=False)
    with pytest.raises(RuntimeError):
        ops.prodigy_step(flat, g, m, v, s, before, state, betas=(0.0, 0.999))


def test_trainer_prodigy_learns_and_resumes(tmp_path):
    """QwenLoraTrainStep(optimizer="prodigy") with the reference's init_args on the tiny DiT: the loss on a fixed batch falls, d has
    grown from d0, and a checkpoint (prodigyopt state layout) resumes bit-identically."""
    from common import TINY
    from parity_util import build_pair, tiny_embeddings
    from qflux_amd.trainer import QwenLoraTrainStep
    args = dict(use_bias_correction=True, safeguard_warmup=True)
    _, a = build_pair(dict(TINY), device=DEV)
    sa = QwenLoraTrainStep(a, lr=1.0, weight_decay=0.01, optimizer="prodigy", optimizer_args=args)
    e, nz, u = tiny_embeddings(seed=5)
    losses = [sa.train_step(e, noise=nz, u=u).item() for _ in range(120)]
    sd = sa.state_dict()
    g = sd["param_groups"][0]
    assert g["k"] == 120 and g["d"] > 100 * g["d0"] and g["use_bias_correction"] and g["safeguard_warmup"]
    assert set(sd["state"][0]) == {"step", "s", "p0", "exp_avg", "exp_avg_sq"}
    assert losses[-1] < losses[0], (losses[0], losses[-1])
    print('prodigy tiny: d', g['d'], 'loss', losses[0], '->', losses[-1])
    sa.save_checkpoint(str(tmp_path / "ck"))
    sa.train_step(e, noise=nz, u=u)
    want = a.lora_store.pflat.detach().cpu().clone()
    _, b = build_pair(dict(TINY), device=DEV)
    sb = QwenLoraTrainStep(b, lr=0.3, optimizer="prodigy")
    sb.load_checkpoint(str(tmp_path / "ck"), adapter_name="lora_edit")
    assert sb.lr == 1.0 and sb.optimizer_args["safeguard_warmup"] is True
    sb.train_step(e, noise=nz, u=u)
    assert torch.equal(b.lora_store.pflat.detach().cpu(), want)
    with pytest.raises(ValueError):
        QwenLoraTrainStep(b, optimizer="prodigy", optimizer_args={"slice_p": 11})
