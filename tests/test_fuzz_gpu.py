"""Seeded shape fuzz of the whole training step against the CPU oracle (tiny width, 2 blocks): ragged and degenerate token grids
(1 x k images, odd counts, a single text token), 2-4 images per sample, batch 1-3, ranks that are not a multiple of 16, random
target subsets, both entry points (fused step / autograd drop-in).  The reference's own tests cover ragged / padded inputs per
module (tests/src/models/test_*_per_sample_rope.py, test_flux_transformer_padding.py); here the same classes of input go through
the complete step.  Bars: those of parity_util (loss 2e-2, prediction 2e-2, LoRA gradients 4e-2 of the tensor maximum)."""
import random

import pytest
import torch

from parity_util import run_flux_step_parity, run_tiny_step_parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

_QWEN_TARGETS = ["to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out", "img_mlp.net.0.proj",
                 "img_mlp.net.2", "txt_mlp.net.0.proj", "txt_mlp.net.2"]
_FLUX_TARGETS = ["to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out", "ff.net.0.proj", "ff.net.2",
                 "ff_context.net.2", "proj_mlp", "proj_out"]


def _qwen_case(seed):
    rnd = random.Random(1000 + seed)
    n_img = rnd.choice([2, 2, 3, 4])
    dims = [1, 2, 3, 4, 5, 6, 7, 9, 10]
    shapes = tuple((1, rnd.choice(dims), rnd.choice(dims)) for _ in range(n_img))
    return dict(shapes=shapes, T=rnd.choice([1, 3, 7, 16, 33]), B=rnd.choice([1, 2, 3]), r=rnd.choice([1, 4, 8, 12, 16, 24]),
                targets=tuple(sorted(rnd.sample(_QWEN_TARGETS, rnd.choice([1, 3, 4, 6, 12])))), fused=rnd.random() < 0.7)


@pytest.mark.parametrize("seed", range(10))
def test_qwen_step_shape_fuzz_vs_oracle(seed):
    case = _qwen_case(seed)
    res = run_tiny_step_parity(DEV, verbose=False, **case)
    print(case, {k: res[k] for k in ("loss_rel", "pred_rel", "grad_rel_worst", "grad_worst_name") if k in res})
    assert res["ok"], (case, res)


def _flux_case(seed):
    rnd = random.Random(2000 + seed)
    dims = [1, 2, 3, 4, 5, 6, 8, 9]
    return dict(hw=(rnd.choice(dims), rnd.choice(dims)), T=rnd.choice([1, 2, 7, 16, 31]), B=rnd.choice([1, 2, 3]), r=rnd.choice([1, 4, 8, 12, 16]),
                guidance=rnd.random() < 0.5, fused=rnd.random() < 0.7,
                targets=tuple(sorted(rnd.sample(_FLUX_TARGETS, rnd.choice([1, 3, 4, 7, 13])))))


@pytest.mark.parametrize("seed", range(8))
def test_flux_step_shape_fuzz_vs_oracle(seed):
    case = _flux_case(seed)
    res = run_flux_step_parity(DEV, verbose=False, **case)
    print(case, {k: res[k] for k in ("loss_rel", "pred_rel", "grad_rel_worst", "grad_worst_name") if k in res})
    assert res["ok"], (case, res)
