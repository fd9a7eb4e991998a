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


def _flux_multires_case(seed):
    rnd = random.Random(3000 + seed)
    g = torch.Generator().manual_seed(3000 + seed)
    B = rnd.choice([1, 2, 3, 4])
    T = rnd.choice([1, 5, 7, 16])
    dims = [1, 2, 3, 4, 5, 6]
    samples, lens = [], []
    for _ in range(B):
        h, w = rnd.choice(dims), rnd.choice(dims)
        ctl = [(rnd.choice(dims), rnd.choice(dims)) for _ in range(rnd.choice([1, 1, 2]))]
        n_t, n_c = h * w, sum(a * b for a, b in ctl)
        lens.append(n_t + n_c)
        samples.append(dict(image_latents=torch.randn(n_t, 64, generator=g).half(), control_latents=torch.randn(n_c, 64, generator=g).half(),
                            hw=(h, w), control_hw=ctl, noise=torch.randn(n_t, 64, generator=g).to(torch.bfloat16),
                            t=torch.rand((), generator=g).to(torch.bfloat16)))
    txt = dict(text_ids=torch.zeros(T, 3), pooled_prompt_embeds=torch.randn(B, 16, generator=g).half(),
               prompt_embeds=torch.randn(B, T, 64, generator=g).half())
    return samples, txt, lens, dict(r=rnd.choice([2, 4, 8]), fused=rnd.random() < 0.6,
                                    targets=tuple(sorted(rnd.sample(_FLUX_TARGETS, rnd.choice([2, 4, 13])))))


@pytest.mark.parametrize("seed", range(8))
def test_flux_multires_step_shape_fuzz_vs_oracle(seed):
    """cfg #5 class of inputs: ragged batches (every sample its own target grid and 1-2 control grids), right-padded, additive key
    mask, per-sample RoPE; padded output rows exactly zero; loss / prediction / every adapter gradient vs the oracle restatement of
    the reference's custom model (transformer_flux_custom.py)."""
    from common import FLUX_TINY, fill_weights
    from oracle import flux_dit as FO
    from oracle import qwen_dit as O
    from parity_util import relmax
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import FluxKontextTrainStep
    BF = torch.bfloat16
    samples, txt, lens, opt = _flux_multires_case(seed)
    cfg = dict(FLUX_TINY, guidance_embeds=True, joint_attention_dim=64)
    oracle = FO.OracleFluxDiT(**cfg)
    O.add_lora(oracle, r=opt["r"], lora_alpha=2 * opt["r"], adapter_name="a", target_modules=opt["targets"])
    fill_weights(oracle, seed=6 + seed)
    for n, p in oracle.named_parameters():
        if "lora" not in n:
            p.data = p.data.to(BF)
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**cfg)
    hip.add_adapter(LoraConfig(r=opt["r"], lora_alpha=2 * opt["r"], target_modules=list(opt["targets"])), "a")
    hip.load_state_dict(oracle.state_dict(), strict=True)
    so = [dict(s, control_latents=s["control_latents"].to(BF)) for s in samples]
    loss_o, pred_o = FO.flux_compute_loss_multires(oracle, so, txt, BF, return_pred=True)
    loss_o.float().backward()
    step = FluxKontextTrainStep(hip)
    if opt["fused"]:
        loss_h = step.forward_backward_multires(samples, txt)
    else:
        loss_h = step.compute_loss_multires(samples, txt)
        loss_h.backward()
    plans = [p for k, p in hip._plans.items() if "multires" in k]
    og = {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}
    worst = max([relmax(p.grad, og[n]) for n, p in hip.named_parameters() if "lora" in n and og[n] is not None] + [0.0])
    e_pred = None
    if plans:      # a batch whose samples all share one shape takes the shared-RoPE program (transformer_flux_custom.py:262-273)
        out = plans[0].A["out"].view(len(samples), -1, 64)
        e_pred = relmax(out[:, :pred_o.shape[1]], pred_o)
        for b, Ln in enumerate(lens):
            assert out[b, Ln:].numel() == 0 or out[b, Ln:].abs().max().item() == 0.0
    print(dict(B=len(samples), T=txt["prompt_embeds"].shape[1], lens=lens, **opt), "loss", loss_o.item(), loss_h.item(), "pred", e_pred, "grad", worst)
    assert abs(loss_h.item() - loss_o.item()) / abs(loss_o.item()) < 2e-2
    assert (e_pred is None or e_pred < 2e-2) and worst < 8e-2


@pytest.mark.parametrize("seed", range(6))
def test_qwen_multires_forward_backward_shape_fuzz_vs_oracle(seed):
    """Qwen multi-resolution model (transformer_qwen_custom.py:384-573) on random ragged batches: per-sample shape lists of 2-3
    images, ragged text lengths, right-padded to the batch maximum with the padding mask.  Valid rows vs the oracle, padded rows
    exactly zero, every adapter gradient."""
    from common import TINY
    from parity_util import build_pair
    BF = torch.bfloat16
    rnd = random.Random(4000 + seed)
    g = torch.Generator().manual_seed(4000 + seed)
    B = rnd.choice([2, 3, 4])
    dims = [1, 2, 3, 4, 5, 6, 8]
    shapes = [[(1, rnd.choice(dims), rnd.choice(dims)) for _ in range(rnd.choice([2, 2, 3]))] for _ in range(B)]
    n_img = [sum(f * h * w for f, h, w in sh) for sh in shapes]
    S_max, T = max(n_img), rnd.choice([4, 9, 17])
    lens = [rnd.randint(1, T) for _ in range(B)]
    lens[rnd.randrange(B)] = T
    full = torch.zeros(B, T + S_max, dtype=torch.bool)
    x = torch.zeros(B, S_max, 64)
    pe = torch.zeros(B, T, TINY["joint_attention_dim"])
    for b in range(B):
        full[b, :lens[b]] = True
        full[b, T:T + n_img[b]] = True
        x[b, :n_img[b]] = torch.randn(n_img[b], 64, generator=g)
        pe[b, :lens[b]] = torch.randn(lens[b], TINY["joint_attention_dim"], generator=g) * 4
    x, pe = x.to(BF), pe.to(BF)
    tt = torch.rand(B, generator=g)
    tgt = torch.randn(B, S_max, TINY["out_channels"] * 4, generator=g)
    valid = full[:, T:]
    r = rnd.choice([2, 4, 8])
    targets = tuple(sorted(rnd.sample(_QWEN_TARGETS, rnd.choice([2, 4, 12]))))
    oracle, hip = build_pair(dict(TINY), r=r, device=DEV, targets=targets, seed=2 + seed)
    out_o = oracle(hidden_states=x, encoder_hidden_states=pe, timestep=tt, img_shapes=shapes, txt_seq_lens=lens, attention_mask=full)[0]
    (((out_o.float() - tgt) ** 2) * valid.unsqueeze(-1)).sum().div(valid.sum() * tgt.shape[-1]).backward()
    out_h = hip(hidden_states=x.to(DEV), encoder_hidden_states=pe.to(DEV), timestep=tt.to(DEV), img_shapes=shapes, txt_seq_lens=lens,
                attention_mask=full, return_dict=False)[0]
    (((out_h.float() - tgt.to(DEV)) ** 2) * valid.to(DEV).unsqueeze(-1)).sum().div(valid.sum().item() * tgt.shape[-1]).backward()
    oh = out_h.float().cpu()
    assert (~valid).sum() == 0 or oh[~valid].abs().max().item() == 0.0
    e = ((oh - out_o.float())[valid].abs().max() / out_o.float()[valid].abs().max()).item()
    og = {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}
    worst = max([((p.grad.cpu() - og[n]).abs().max() / og[n].abs().max()).item() for n, p in hip.named_parameters()
                 if "lora" in n and og[n] is not None and og[n].abs().max() > 0] + [0.0])
    print(dict(B=B, T=T, lens=lens, n_img=n_img, r=r, targets=targets), "pred rel", e, "grad worst", worst)
    assert e < 2e-2 and worst < 8e-2
