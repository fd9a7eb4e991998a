"""Parity at BASELINE.json's full WIDTH and sequence length (D=3072 = 24x128 heads, S_i=2048, T=384, r=16; 1-2 blocks):
the oracle itself on one block (fp32 on the host cores, the only place an oracle finishes in seconds) and
size-independent properties of the whole step that need no oracle."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "qwen-image-finetune_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
FULL = dict(patch_size=2, in_channels=64, out_channels=16, attention_head_dim=128, num_attention_heads=24, joint_attention_dim=3584,
            axes_dims_rope=(16, 56, 56))


def _emb(B=1, side=32, T=384, seed=5):
    g = torch.Generator().manual_seed(seed)
    S_t = side * side
    emb = dict(image_latents=torch.randn(B, S_t, 64, generator=g).half().float(), control_latents=torch.randn(B, S_t, 64, generator=g).half().float(),
               prompt_embeds=(torch.randn(B, T, 3584, generator=g) * 4).half().float(), prompt_embeds_mask=torch.ones(B, T, dtype=torch.int64),
               img_shapes=[[(1, side, side), (1, side, side)]] * B)
    return emb, torch.randn(B, S_t, 64, generator=g), torch.tensor([0.7109, 0.1611][:B])


def _hip(layers, seed=3, lora=True):
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    with torch.device(DEV):
        m = QwenImageTransformer2DModel(num_layers=layers, **FULL)
    g = torch.Generator(device=DEV).manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 2:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.02).to(p.dtype))
            elif "norm" in n:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.02).to(p.dtype))
    if lora:
        m.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
    return m


def test_one_full_width_block_vs_oracle():
    """HIP vs the oracle on ONE full-width block at the full sequence length, both in the reference's training dtype layout
    (bf16 trunk and activations, fp32 adapters); the oracle's own fp32 run shows how much of the difference is bf16 rounding."""
    import time
    from oracle import qwen_dit as O
    from qflux_amd.trainer import QwenLoraTrainStep
    hip = _hip(1)
    with torch.no_grad():   # LoRA B is zero-initialised: give it values so that dA is non-zero
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(7)).to(p.device) * 1e-2)
    hip.refresh_lora_operands()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    emb, noise, u = _emb()
    sd = {k: v.cpu() for k, v in hip.state_dict().items()}
    res = {}
    for tag, dt in (("bf16", BF), ("fp32", torch.float32)):
        oracle = O.OracleQwenDiT(num_layers=1, **FULL)
        O.add_lora(oracle, r=16, lora_alpha=16, adapter_name="default")
        oracle.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
        if dt == BF:
            for n, p in oracle.named_parameters():
                if "lora" not in n:
                    p.data = p.data.to(BF)
        t0 = time.time()
        loss_o, pred_o = O.qwen_compute_loss(oracle, emb, noise, u, dt, return_pred=True)
        loss_o.float().backward()
        res[tag] = (loss_o.item(), pred_o.detach().float(), {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}, time.time() - t0)
    step = QwenLoraTrainStep(hip)
    loss_h = step.forward_backward(emb, noise=noise, u=u).item()
    plan = list(hip._plans.values())[0]
    pred_h = plan.A["out"].view(1, -1, 64)[:, : res["bf16"][1].shape[1]].float().cpu()

    def cmp(pred, grads, ref):
        e = ((pred - ref[1]).abs().max() / ref[1].abs().max()).item()
        cs = []
        for n, g in grads.items():
            if ref[2][n] is not None and ref[2][n].abs().max() > 0:
                a, b = g.float().cpu().flatten(), ref[2][n].float().flatten()
                cs.append((torch.dot(a, b) / (a.norm() * b.norm())).item())
        return e, min(cs), len(cs)
    hg = {n: p.grad for n, p in hip.named_parameters() if "lora" in n}
    e_hb, c_hb, n_hb = cmp(pred_h, hg, res["bf16"])
    e_hf, c_hf, _ = cmp(pred_h, hg, res["fp32"])
    e_bf, c_bf, _ = cmp(res["bf16"][1], res["bf16"][2], res["fp32"])
    print(f"full-width block: loss hip {loss_h:.5f} oracle-bf16 {res['bf16'][0]:.5f} oracle-fp32 {res['fp32'][0]:.5f}; "
          f"pred rel hip~bf16 {e_hb:.4f} hip~fp32 {e_hf:.4f} bf16~fp32 {e_bf:.4f}; grad cos hip~bf16 {c_hb:.4f} hip~fp32 {c_hf:.4f} "
          f"bf16~fp32 {c_bf:.4f}; oracle seconds {res['bf16'][3]:.1f}/{res['fp32'][3]:.1f}")
    assert n_hb == 8
    assert abs(loss_h - res["bf16"][0]) / abs(res["bf16"][0]) < 1e-2     # |dloss| bar of the reference's e2e tests
    # the HIP path must sit at least as close to the fp32 truth as the eager-bf16 restatement does (same rounding points)
    assert e_hf < 1.25 * e_bf + 1e-3 and c_hf > c_bf - 0.02
    assert e_hb < 1.5 * e_bf + 1e-3


def test_zero_lora_b_is_the_base_model_exactly():
    """With lora_B = 0 (peft's init) the adapted model must reproduce the frozen model BIT-exactly: the K-extension adds
    exact zeros after the bf16 mid-rounding of the base output."""
    base = _hip(2, lora=False)
    emb, noise, u = _emb()
    x = torch.cat([emb["image_latents"], emb["control_latents"]], 1).to(BF).to(DEV)
    pe = emb["prompt_embeds"].to(BF).to(DEV)
    t = torch.tensor([0.5], device=DEV)
    kw = dict(hidden_states=x, timestep=t, encoder_hidden_states=pe, encoder_hidden_states_mask=None, img_shapes=emb["img_shapes"],
              txt_seq_lens=[384], return_dict=False)
    with torch.no_grad():
        out0 = base(**kw)[0].clone()
        from qflux_amd.modules import LoraConfig
        base.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
        out1 = base(**kw)[0].clone()
    assert torch.equal(out0, out1)


def test_backward_is_linear_in_the_loss_scale_and_samples_are_independent():
    from qflux_amd.trainer import QwenLoraTrainStep
    hip = _hip(2)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(9)).to(p.device) * 1e-2)
    step = QwenLoraTrainStep(hip)
    emb, noise, u = _emb(B=2)
    st = hip.lora_store
    l1 = step.forward_backward(emb, noise=noise, u=u, grad_scale=1.0).item()
    g1 = st.gflat.clone(); step.zero_grad()
    l2 = step.forward_backward(emb, noise=noise, u=u, grad_scale=0.5).item()
    g2 = st.gflat.clone(); step.zero_grad()
    # a power-of-two loss scale is exact in bf16: every launch of the backward must scale with it
    lin = ((g2 * 2 - g1).abs().max() / g1.abs().max()).item()
    # swapping the two samples swaps the predictions bit-exactly and leaves loss / summed gradients unchanged (atomics order aside)
    plan = list(hip._plans.values())[0]
    outA = plan.A["out"].view(2, -1, 64).clone()
    sw = {k: (v.flip(0) if isinstance(v, torch.Tensor) else v) for k, v in emb.items()}
    l3 = step.forward_backward(sw, noise=noise.flip(0), u=u.flip(0)).item()
    g3 = st.gflat.clone(); step.zero_grad()
    outB = plan.A["out"].view(2, -1, 64)
    print("linearity", lin, "swap grad", ((g3 - g1).abs().max() / g1.abs().max()).item(), "loss", l1, l2, l3)
    assert abs(l1 - l2) < 1e-5 * abs(l1) and lin < 1e-5    # the loss reduction uses fp32 atomics: order varies
    assert torch.equal(outA.flip(0), outB)
    assert abs(l3 - l1) < 1e-6 * abs(l1) + 1e-7 and ((g3 - g1).abs().max() / g1.abs().max()).item() < 1e-4


def test_one_full_width_block_all_linear_vs_oracle():
    """target_modules="all-linear" at the model's full WIDTH (one block, short sequences so that the CPU oracle finishes in
    seconds): every Linear adapted -- the D = 3072 instantiations of the modulation-gradient kernel, the conditioning-head banks
    (N = 18432 / 6144 modulation linears on qfx_mod_gemv / qfx_mod_gemv_t), feed-forward and embedder adapters -- against the bf16
    oracle, tensor by tensor."""
    from oracle import qwen_dit as O
    from parity_util import _grad_tol_factor, relmax
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    with torch.device(DEV):
        hip = QwenImageTransformer2DModel(num_layers=1, **FULL)
    g = torch.Generator(device=DEV).manual_seed(3)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if p.ndim == 1 and "norm" in n:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.02).to(p.dtype))
    hip.add_adapter(LoraConfig(r=16, lora_alpha=16, target_modules="all-linear"), "default", generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(7)).to(p.device) * 1e-2)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    emb, noise, u = _emb(B=2, side=16, T=64)
    oracle = O.OracleQwenDiT(num_layers=1, **FULL)
    O.add_lora(oracle, r=16, lora_alpha=16, adapter_name="default", target_modules="all-linear")
    oracle.load_state_dict({k: v.float().cpu() for k, v in hip.state_dict().items()}, strict=True)
    for n, p in oracle.named_parameters():
        if "lora" not in n:
            p.data = p.data.to(BF)
    loss_o, pred_o = O.qwen_compute_loss(oracle, emb, noise, u, BF, return_pred=True)
    loss_o.float().backward()
    step = QwenLoraTrainStep(hip)
    loss_h = step.forward_backward(emb, noise=noise, u=u).item()
    plan = list(hip._plans.values())[0]
    pred_h = plan.A["out"].view(2, -1, 64)[:, : pred_o.shape[1]]
    e = relmax(pred_h, pred_o)
    og = {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}
    worst, wname, n_cmp = 0.0, None, 0
    for n, p in hip.named_parameters():
        if "lora" not in n:
            continue
        if og[n] is None:
            assert p.grad.abs().max().item() == 0.0, n
            continue
        r = relmax(p.grad, og[n]) / _grad_tol_factor(n)
        n_cmp += 1
        if r > worst:
            worst, wname = r, n
    print(f"full-width all-linear block: loss {loss_h:.5f} / {loss_o.item():.5f}, pred rel {e:.4f}, worst grad rel {worst:.4f} ({wname}), {n_cmp} tensors")
    assert abs(loss_h - loss_o.item()) / abs(loss_o.item()) < 1e-2 and e < 2e-2 and worst < 2e-2 and n_cmp >= 30      # (the single block is also the LAST one: its text tail is dead compute)
