"""FlowMatchEulerSchedule (restated diffusers FlowMatchEulerDiscreteScheduler, dynamic exponential shift) on CPU: the schedule must
come from the checkpoint's scheduler_config.json, not from fall-back constants (base_trainer.py:1009-1043)."""
import json
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))


def _import_sampling():
    # the package dlopens libqfx.so on import; the schedule itself is host-side arithmetic
    from qflux_amd import sampling
    return sampling


def test_schedule_from_config_closed_forms(tmp_path):
    S = _import_sampling()
    cfg = dict(_class_name="FlowMatchEulerDiscreteScheduler", num_train_timesteps=1000, base_image_seq_len=256, max_image_seq_len=8192,
               base_shift=0.5, max_shift=0.9, shift_terminal=0.02, use_dynamic_shifting=True, time_shift_type="exponential", shift=1.0)
    os.makedirs(tmp_path / "scheduler")
    with open(tmp_path / "scheduler" / "scheduler_config.json", "w") as f:
        json.dump(cfg, f)
    for src in (cfg, str(tmp_path), str(tmp_path / "scheduler"), str(tmp_path / "scheduler" / "scheduler_config.json")):
        sch = S.FlowMatchEulerSchedule.from_config(src)
        assert sch.cfg["max_image_seq_len"] == 8192 and sch.cfg["max_shift"] == 0.9 and sch.shift_terminal == 0.02
    n, seq = 8, 4096
    ts = sch.set_timesteps(n, seq)
    # calculate_shift is the line through (256, 0.5) and (8192, 0.9)  (custom_flowmatch_scheduler.py:20-30)
    mu = 0.5 + (0.9 - 0.5) * (seq - 256) / (8192 - 256)
    assert abs(S.calculate_shift(seq, 256, 8192, 0.5, 0.9) - mu) < 1e-12
    raw = [math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0)) for s in torch.linspace(1.0, 1.0 / n, n, dtype=torch.float64).tolist()]
    scale = (1.0 - raw[-1]) / (1.0 - 0.02)
    want = [1.0 - (1.0 - r) / scale for r in raw]
    assert torch.allclose(sch.sigmas[:-1].double(), torch.tensor(want, dtype=torch.float64), atol=1e-6)
    assert abs(sch.sigmas[0].item() - 1.0) < 1e-6 and abs(sch.sigmas[n - 1].item() - 0.02) < 1e-6 and sch.sigmas[n].item() == 0.0
    assert torch.equal(ts, sch.sigmas[:-1] * 1000)
    # the fall-back constants give a DIFFERENT table: a sampler built without the config does not follow this checkpoint
    dflt = S.FlowMatchEulerSchedule()
    dflt.set_timesteps(n, seq)
    assert (dflt.sigmas - sch.sigmas).abs().max() > 1e-2
    # Euler step in fp32, result in the model dtype
    x = torch.randn(2, 4, 8).bfloat16()
    v = torch.randn(2, 4, 8).bfloat16()
    out = sch.step(v, 0.7, 0.5, x)
    assert out.dtype == torch.bfloat16 and torch.equal(out, (x.float() + (0.5 - 0.7) * v.float()).bfloat16())


def test_schedule_rejects_unrestated_variants():
    S = _import_sampling()
    for bad in (dict(use_dynamic_shifting=False), dict(time_shift_type="linear"), dict(use_karras_sigmas=True)):
        with pytest.raises(NotImplementedError):
            S.FlowMatchEulerSchedule.from_config(bad)
