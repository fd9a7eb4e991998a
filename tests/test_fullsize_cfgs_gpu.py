"""Driver-visible parity at the full sizes of EVERY BASELINE.json config (the round-1 suite only reached cfg #2's size):

  cfg #4  attention forward + backward at S = 8576 (1024^2 target + control, T = 384), all 24 heads x 128, vs an fp32
          restatement of F.scaled_dot_product_attention (reference op: transformer_qwenimage.py:329-337) -- grid sizes, lse /
          dsum indexing and the XCD remap with 24 heads x 67 query blocks;
  cfg #3  ONE full-width Qwen block at S_i = 3072 (three 512^2 images, frame index 0/1/2 = the 2509 cumulative-offset RoPE,
          transformer_qwenimage.py:243-248), T = 512, LoRA r = 32, vs the bf16 oracle (and the oracle's own fp32 run as yardstick);
  cfg #1/#5  ONE full-width FLUX double block + ONE single block (D = 3072, joint dim 4096, pooled 768), LoRA r = 16: the shared
          step and a ragged two-bucket batch {320^2, 512^2} through the multi-resolution step (per-sample RoPE, additive key mask,
          padded rows exactly zero; transformer_flux_custom.py:537-616), vs oracle/flux_dit.py.
"""
import math
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "qwen-image-finetune_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()


# ---------------------------------------------------------------------------------------------- cfg #4
def test_attention_s8576_all_heads_vs_fp32_sdpa():
    from qflux_amd import ops
    Bn, S, H, dh = 1, 8576, 24, 128
    D = H * dh
    S_pad = (S + 63) // 64 * 64
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(Bn, S, 3 * D, generator=g).to(BF).to(DEV)
    do = (torch.randn(Bn, S, D, generator=g) * 0.5).to(BF).to(DEV)
    scale = 1.0 / math.sqrt(dh)
    ld = 3 * D
    Q, K, V = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    O = torch.empty(Bn, S, D, dtype=BF, device=DEV)
    lse2 = torch.zeros(Bn, H, S_pad, dtype=torch.float32, device=DEV)
    dsum = torch.zeros(Bn, H, S_pad, dtype=torch.float32, device=DEV)
    dqkv = torch.zeros(Bn, S, 3 * D, dtype=BF, device=DEV)
    a = ops.attn_args(Bn, S, S_pad, H, dh, scale, Q=Q, K=K, V=V, ldq=ld, ldk=ld, ldv=ld, O=O, ldo=D, lse2=lse2)
    a.dO, a.lddo, a.dsum = do.data_ptr(), D, dsum.data_ptr()
    a.dQ, a.dK, a.dV = dqkv[:, :, :D].data_ptr(), dqkv[:, :, D:2 * D].data_ptr(), dqkv[:, :, 2 * D:].data_ptr()
    a.lddq = a.lddk = a.lddv = ld
    ops.attn_call("qfx_attn_fwd", a)
    ops.attn_call("qfx_attn_bwd_dq", a)      # also publishes dsum = rowsum(dO * O)
    ops.attn_call("qfx_attn_bwd_dkv", a)
    torch.cuda.synchronize()
    # fp32 reference (plain softmax attention + its analytic backward), four heads at a time: S x S scores are 294 MB per head
    worst = {}
    for h0 in range(0, H, 4):
        hs = slice(h0 * dh, (h0 + 4) * dh)
        q, k, v = (t[0, :, hs].float().reshape(S, 4, dh).transpose(0, 1) for t in (Q, K, V))       # [4, S, dh]
        dO = do[0, :, hs].float().reshape(S, 4, dh).transpose(0, 1)
        s = torch.matmul(q, k.transpose(1, 2)) * scale
        lse = torch.logsumexp(s, dim=-1)
        p = torch.exp(s - lse[..., None])
        del s
        o = torch.matmul(p, v)
        dv = torch.matmul(p.transpose(1, 2), dO)
        dp = torch.matmul(dO, v.transpose(1, 2))
        delta = (dO * o).sum(-1, keepdim=True)
        ds = p * (dp - delta) * scale
        del p, dp
        dq = torch.matmul(ds, k)
        dk = torch.matmul(ds.transpose(1, 2), q)
        del ds

        def put(name, got, ref):
            worst[name] = max(worst.get(name, 0.0), _rel(got, ref))
        back = lambda t: t.transpose(0, 1).reshape(S, 4 * dh)   # noqa: E731
        put("o", O[0, :, hs], back(o))
        put("lse", lse2[0, h0:h0 + 4, :S], lse / math.log(2.0))
        put("dsum", dsum[0, h0:h0 + 4, :S], delta[..., 0])
        put("dq", dqkv[0, :, hs], back(dq))
        put("dk", dqkv[0, :, D + h0 * dh: D + (h0 + 4) * dh], back(dk))
        put("dv", dqkv[0, :, 2 * D + h0 * dh: 2 * D + (h0 + 4) * dh], back(dv))
    print("attention S=8576 H=24 dh=128 rel-to-max errors:", {k: round(v, 5) for k, v in worst.items()})
    assert worst["o"] < 1e-2 and worst["lse"] < 1e-3 and worst["dsum"] < 2e-2
    assert worst["dq"] < 2e-2 and worst["dk"] < 2e-2 and worst["dv"] < 2e-2
    assert S_pad == S or lse2[0, :, S:].abs().max().item() == 0.0      # pad columns of the statistics stay untouched


# ---------------------------------------------------------------------------------------------- cfg #3
QWEN_FULL = dict(patch_size=2, in_channels=64, out_channels=16, attention_head_dim=128, num_attention_heads=24, joint_attention_dim=3584,
                 axes_dims_rope=(16, 56, 56))


def test_cfg3_full_width_block_three_images_r32_vs_oracle():
    from oracle import qwen_dit as O
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    r, T, side = 32, 512, 32
    shapes = [(1, side, side)] * 3
    with torch.device(DEV):
        hip = QwenImageTransformer2DModel(num_layers=1, **QWEN_FULL)
    g = torch.Generator(device=DEV).manual_seed(13)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if p.ndim == 1 and "norm" in n:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.02).to(p.dtype))
    hip.add_adapter(LoraConfig(r=r, lora_alpha=r), "default", generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(7)).to(p.device) * 1e-2)
    gg = torch.Generator().manual_seed(21)
    S_t = side * side
    emb = dict(image_latents=torch.randn(1, S_t, 64, generator=gg).half().float(), control_latents=torch.randn(1, 2 * S_t, 64, generator=gg).half().float(),
               prompt_embeds=(torch.randn(1, T, 3584, generator=gg) * 4).half().float(), prompt_embeds_mask=torch.ones(1, T, dtype=torch.int64),
               img_shapes=[shapes])
    noise, u = torch.randn(1, S_t, 64, generator=gg), torch.tensor([0.4])
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    sd = {k: v.cpu() for k, v in hip.state_dict().items()}
    res = {}
    for tag, dt in (("bf16", BF), ("fp32", torch.float32)):
        oracle = O.OracleQwenDiT(num_layers=1, **QWEN_FULL)
        O.add_lora(oracle, r=r, lora_alpha=r, adapter_name="default")
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
    assert plan.S_i == 3 * S_t and plan.T == T
    pred_h = plan.A["out"].view(1, -1, 64)[:, :S_t].float().cpu()
    hg = {n: p.grad.float().cpu() for n, p in hip.named_parameters() if "lora" in n}

    def cmp(pred, grads, ref):
        cs = [_cos(gr, ref[2][n]) for n, gr in grads.items() if ref[2][n] is not None and ref[2][n].abs().max() > 0]
        return _rel(pred, ref[1]), min(cs), len(cs)
    e_hb, c_hb, n_hb = cmp(pred_h, hg, res["bf16"])
    e_hf, c_hf, _ = cmp(pred_h, hg, res["fp32"])
    e_bf, c_bf, _ = cmp(res["bf16"][1], {n: v for n, v in res["bf16"][2].items() if v is not None}, res["fp32"])
    print(f"cfg#3 block (S_i=3072, T=512, r=32): loss hip {loss_h:.5f} oracle-bf16 {res['bf16'][0]:.5f} fp32 {res['fp32'][0]:.5f}; pred rel "
          f"hip~bf16 {e_hb:.4f} hip~fp32 {e_hf:.4f} bf16~fp32 {e_bf:.4f}; grad cos hip~bf16 {c_hb:.4f} hip~fp32 {c_hf:.4f} bf16~fp32 {c_bf:.4f}; "
          f"oracle s {res['bf16'][3]:.1f}/{res['fp32'][3]:.1f}")
    assert n_hb == 8
    assert abs(loss_h - res["bf16"][0]) / abs(res["bf16"][0]) < 1e-2
    assert e_hf < 1.25 * e_bf + 1e-3 and c_hf > c_bf - 0.02 and e_hb < 1.5 * e_bf + 1e-3


# ---------------------------------------------------------------------------------------------- cfg #4 (whole step at 1024^2)
def _qwen_full(layers, r, seed=13, b_seed=7):
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    with torch.device(DEV):
        hip = QwenImageTransformer2DModel(num_layers=layers, **QWEN_FULL)
    g = torch.Generator(device=DEV).manual_seed(seed)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if p.ndim == 1 and "norm" in n:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.02).to(p.dtype))
    if r is None:
        return hip
    hip.add_adapter(LoraConfig(r=r, lora_alpha=r), "default", generator=torch.Generator().manual_seed(0))
    if b_seed is not None:
        with torch.no_grad():
            for n, p in hip.named_parameters():
                if "lora_B" in n:
                    p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(b_seed)).to(p.device) * 1e-2)
    return hip


def _emb_1024(B=1, T=384, seed=23):
    g = torch.Generator().manual_seed(seed)
    side = 64                                   # 1024^2 px -> 64 x 64 packed-latent tokens per image
    S_t = side * side
    emb = dict(image_latents=torch.randn(B, S_t, 64, generator=g).half().float(), control_latents=torch.randn(B, S_t, 64, generator=g).half().float(),
               prompt_embeds=(torch.randn(B, T, 3584, generator=g) * 4).half().float(), prompt_embeds_mask=torch.ones(B, T, dtype=torch.int64),
               img_shapes=[[(1, side, side), (1, side, side)]] * B)
    return emb, torch.randn(B, S_t, 64, generator=g), torch.tensor([0.4, 0.8][:B])


def test_cfg4_full_width_block_1024sq_step_vs_oracle():
    """cfg #4 as a STEP (not only its attention kernels): one full-width Qwen block at S_i = 8192 (1024^2 target + control), T = 384,
    r = 16 through QwenLoraTrainStep.forward_backward -- the GEMM / LayerNorm / rank-r kernels at M = 8192 / 8576 rows and the plan
    arena at that size -- vs the bf16 oracle of the same step (reference: qwen_image_edit_trainer.py:777-849 on
    transformer_qwenimage.py:570-672)."""
    from oracle import qwen_dit as O
    from qflux_amd.trainer import QwenLoraTrainStep
    r = 16
    hip = _qwen_full(1, r)
    emb, noise, u = _emb_1024()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    oracle = O.OracleQwenDiT(num_layers=1, **QWEN_FULL)
    O.add_lora(oracle, r=r, lora_alpha=r, adapter_name="default")
    oracle.load_state_dict({k: v.float().cpu() for k, v in hip.state_dict().items()}, strict=True)
    for n, p in oracle.named_parameters():
        if "lora" not in n:
            p.data = p.data.to(BF)
    t0 = time.time()
    loss_o, pred_o = O.qwen_compute_loss(oracle, emb, noise, u, BF, return_pred=True)
    loss_o.float().backward()
    t_or = time.time() - t0
    step = QwenLoraTrainStep(hip)
    loss_h = step.forward_backward(emb, noise=noise, u=u).item()
    plan = list(hip._plans.values())[0]
    assert plan.S_i == 8192 and plan.S == 8576
    S_t = 4096
    e = _rel(plan.A["out"].view(1, -1, 64)[:, :S_t].cpu(), pred_o)
    og = {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}
    cs, rels = [], []
    for n, p in hip.named_parameters():
        if "lora" in n and og[n] is not None and og[n].abs().max() > 0:
            cs.append(_cos(p.grad.cpu(), og[n]))
            rels.append(_rel(p.grad.cpu(), og[n]))
    print(f"cfg#4 block (S_i=8192, T=384, r=16): loss hip {loss_h:.5f} oracle-bf16 {loss_o.item():.5f}; pred rel {e:.4f}; LoRA grads n={len(cs)} "
          f"min cos {min(cs):.4f} worst rel {max(rels):.4f}; oracle {t_or:.1f} s")
    assert len(cs) == 8
    assert abs(loss_h - loss_o.item()) / abs(loss_o.item()) < 1e-2 and e < 2e-2 and min(cs) > 0.995 and max(rels) < 2e-2


def test_cfg4_properties_at_1024sq_two_blocks():
    """Size-independent properties at the cfg #4 shape, two blocks (so that a complete block backward incl. the q/k/v dX GEMMs and
    the LayerNorm backward runs at M = 8192): zero-B exactness (adapted model == frozen model, bit for bit) and linearity of the
    whole backward in a power-of-two loss scale."""
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    hip = _qwen_full(2, None)                     # frozen model, no adapters yet
    emb, noise, u = _emb_1024()
    x = torch.cat([emb["image_latents"], emb["control_latents"]], 1).to(BF).to(DEV)
    kw = dict(hidden_states=x, timestep=torch.tensor([0.5], device=DEV), encoder_hidden_states=emb["prompt_embeds"].to(BF).to(DEV),
              encoder_hidden_states_mask=None, img_shapes=emb["img_shapes"], txt_seq_lens=[384], return_dict=False)
    with torch.no_grad():
        out_base = hip(**kw)[0].clone()
        hip.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))   # peft init: lora_B = 0
        out_lora = hip(**kw)[0].clone()
    assert torch.equal(out_lora, out_base)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(9)).to(p.device) * 1e-2)
    step = QwenLoraTrainStep(hip)
    st = hip.lora_store
    l1 = step.forward_backward(emb, noise=noise, u=u, grad_scale=1.0).item()
    g1 = st.gflat.clone(); step.zero_grad()
    l2 = step.forward_backward(emb, noise=noise, u=u, grad_scale=0.5).item()
    g2 = st.gflat.clone(); step.zero_grad()
    lin = ((g2 * 2 - g1).abs().max() / g1.abs().max()).item()
    print(f"cfg#4 two blocks: loss {l1:.5f} / {l2:.5f}, backward linearity {lin:.2e}, |g| max {g1.abs().max().item():.3e}")
    assert abs(l1 - l2) < 1e-5 * abs(l1) and lin < 1e-5 and g1.abs().max().item() > 0 and torch.isfinite(g1).all()


def test_cfg4_step_is_bit_reproducible():
    """Every arena tensor of two identical fused steps at the cfg #4 shape (two blocks, S = 8576: four-wave attention blocks, every
    fused epilogue) comes out bit-identical when both start from a zeroed arena.  Only the flat LoRA gradient is exempt from the
    bit test: the weight-gradient launches (and the scalar loss) add with fp32 atomics (order-dependent in the last bit): compared to 1e-5 / 1e-6.
    Round 4: a codegen change made ~0.1 % of the 16-row fragments of the fused QK-norm backward irreproducible (tools/find_nondet.py
    locates the launch); nothing else in the suite looks at run-to-run equality."""
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    hip = _qwen_full(2, None)
    emb, noise, u = _emb_1024()
    hip.add_adapter(LoraConfig(r=16, lora_alpha=16), "default", generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(9)).to(p.device) * 1e-2)
    step = QwenLoraTrainStep(hip)
    st = hip.lora_store
    step.forward_backward(emb, noise=noise, u=u); step.zero_grad()
    plan = list(hip._plans.values())[0]
    from parity_util import assert_step_bit_reproducible
    assert_step_bit_reproducible(plan, lambda: step.forward_backward(emb, noise=noise, u=u), st.gflat, step.zero_grad, "cfg #4, two blocks")


def test_cfg3_step_is_bit_reproducible():
    """VERDICT r4 #3c: the same run-to-run bit test at the cfg #3 shape (three 512^2 images: S_i = 3072, T = 512, LoRA r = 32, two blocks)."""
    from parity_util import assert_step_bit_reproducible
    from qflux_amd.trainer import QwenLoraTrainStep
    hip = _qwen_full(2, 32)
    side, T = 32, 512
    gg = torch.Generator().manual_seed(21)
    S_t = side * side
    emb = dict(image_latents=torch.randn(1, S_t, 64, generator=gg).half().float(), control_latents=torch.randn(1, 2 * S_t, 64, generator=gg).half().float(),
               prompt_embeds=(torch.randn(1, T, 3584, generator=gg) * 4).half().float(), prompt_embeds_mask=torch.ones(1, T, dtype=torch.int64),
               img_shapes=[[(1, side, side)] * 3])
    noise, u = torch.randn(1, S_t, 64, generator=gg), torch.tensor([0.4])
    step = QwenLoraTrainStep(hip)
    step.forward_backward(emb, noise=noise, u=u); step.zero_grad()
    plan = list(hip._plans.values())[0]
    assert plan.S_i == 3 * S_t and plan.T == T
    assert_step_bit_reproducible(plan, lambda: step.forward_backward(emb, noise=noise, u=u), hip.lora_store.gflat, step.zero_grad, "cfg #3, two blocks")


# ---------------------------------------------------------------------------------------------- cfg #1 / #5
FLUX_FULL = dict(patch_size=1, in_channels=64, out_channels=64, num_layers=1, num_single_layers=1, attention_head_dim=128,
                 num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True,
                 axes_dims_rope=(16, 56, 56))


def _flux_pair(r=16):
    from oracle import flux_dit as FO
    from oracle import qwen_dit as O
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**FLUX_FULL)
    g = torch.Generator(device=DEV).manual_seed(17)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if p.ndim == 1 and "norm" in n:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.02).to(p.dtype))
    hip.add_adapter(LoraConfig(r=r, lora_alpha=r), "default", generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(8)).to(p.device) * 1e-2)
    oracle = FO.OracleFluxDiT(**FLUX_FULL)
    O.add_lora(oracle, r=r, lora_alpha=r, adapter_name="default")
    oracle.load_state_dict({k: v.float().cpu() for k, v in hip.state_dict().items()}, strict=True)
    for n, p in oracle.named_parameters():
        if "lora" not in n:
            p.data = p.data.to(BF)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    return oracle, hip, FO


def _grad_report(oracle, hip):
    og = {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}
    cs, rels = [], []
    for n, p in hip.named_parameters():
        if "lora" not in n:
            continue
        if og[n] is None or og[n].abs().max() == 0:
            continue
        cs.append(_cos(p.grad.cpu(), og[n]))
        rels.append(_rel(p.grad.cpu(), og[n]))
    return min(cs), max(rels), len(cs)


def test_flux_full_width_double_and_single_block_shared_step_vs_oracle():
    from qflux_amd.trainer import FluxKontextTrainStep
    oracle, hip, FO = _flux_pair()
    g = torch.Generator().manual_seed(31)
    B, h, w, T = 1, 32, 32, 512           # 512^2 target + one 512^2 control: S_i = 2048
    S_t = h * w
    ctl = FO.prepare_latent_image_ids(h, w)
    ctl[:, 0] = 1
    emb = dict(image_latents=torch.randn(B, S_t, 64, generator=g).half(), control_latents=torch.randn(B, S_t, 64, generator=g).half(),
               control_ids=ctl, text_ids=torch.zeros(T, 3), latent_hw=(h, w), pooled_prompt_embeds=torch.randn(B, 768, generator=g).half(),
               prompt_embeds=torch.randn(B, T, 4096, generator=g).half())
    noise = torch.randn(B, S_t, 64, generator=g).to(BF)
    t = torch.tensor([0.7109]).to(BF)
    loss_o, pred_o = FO.flux_compute_loss(oracle, dict(emb, control_latents=emb["control_latents"].to(BF)), noise, t, BF, return_pred=True)
    loss_o.float().backward()
    step = FluxKontextTrainStep(hip)
    loss_h = step.forward_backward(emb, noise=noise, t=t).item()
    plan = list(hip._plans.values())[0]
    e = _rel(plan.A["out"].view(B, -1, 64)[:, :S_t].cpu(), pred_o)
    c, rg, n = _grad_report(oracle, hip)
    print(f"flux full width shared: loss {loss_h:.5f} / {loss_o.item():.5f}, pred rel {e:.4f}, LoRA grads n={n} min cos {c:.4f} worst rel {rg:.4f}")
    assert abs(loss_h - loss_o.item()) / abs(loss_o.item()) < 1e-2 and e < 2e-2 and c > 0.995 and rg < 2e-2 and n >= 6


@pytest.mark.parametrize("specs", [
    [((20, 20), [(20, 20)]), ((32, 32), [(32, 32)])],                     # 320^2 and 512^2 buckets: 800 / 2048 image tokens
    [((20, 20), [(20, 20)]), ((40, 40), [(40, 40)])],                     # 320^2 and 640^2 buckets: 800 / 3200 image tokens
    [((40, 26), [(40, 26)]), ((40, 40), [(40, 40)])],                     # non-square 640 x 416 px (40 x 26 tokens = 1040,
], ids=["320+512", "320+640", "640x416+640"])                             # flux_kontext_trainer.py:654-661) next to 640^2
def test_flux_full_width_ragged_two_bucket_batch_vs_oracle(specs):
    from qflux_amd.trainer import FluxKontextTrainStep
    oracle, hip, FO = _flux_pair()
    g = torch.Generator().manual_seed(53)
    T = 512
    samples = []
    for (h, w), ctl in specs:
        n_t, n_c = h * w, sum(a * b for a, b in ctl)
        samples.append(dict(image_latents=torch.randn(n_t, 64, generator=g).half(), control_latents=torch.randn(n_c, 64, generator=g).half(),
                            hw=(h, w), control_hw=ctl, noise=torch.randn(n_t, 64, generator=g).to(BF), t=torch.rand((), generator=g).to(BF)))
    txt = dict(text_ids=torch.zeros(T, 3), pooled_prompt_embeds=torch.randn(2, 768, generator=g).half(),
               prompt_embeds=torch.randn(2, T, 4096, generator=g).half())
    so = [dict(s, control_latents=s["control_latents"].to(BF)) for s in samples]
    loss_o, pred_o = FO.flux_compute_loss_multires(oracle, so, txt, BF, return_pred=True)
    loss_o.float().backward()
    step = FluxKontextTrainStep(hip)
    loss_h = step.forward_backward_multires(samples, txt).item()
    plan = [p for k, p in hip._plans.items() if "multires" in k][0]
    out = plan.A["out"].view(2, -1, 64)
    e = _rel(out[:, : pred_o.shape[1]].cpu(), pred_o)
    n0 = sum(a * b for a, b in [specs[0][0]] + specs[0][1])
    assert out[0, n0:].abs().max().item() == 0.0             # padded rows of the small sample are exactly zero
    c, rg, n = _grad_report(oracle, hip)
    print(f"flux full width ragged: loss {loss_h:.5f} / {loss_o.item():.5f}, pred rel {e:.4f}, LoRA grads n={n} min cos {c:.4f} worst rel {rg:.4f}")
    assert abs(loss_h - loss_o.item()) / abs(loss_o.item()) < 1e-2 and e < 2e-2 and c > 0.995 and rg < 2e-2 and n >= 6


def test_flux_ragged_multires_step_is_bit_reproducible():
    """VERDICT r4 #3c: run-to-run bit test of a ragged two-bucket multi-resolution FLUX step (320^2 + 640^2: masked attention, per-sample
    RoPE, row masks) at full width."""
    from parity_util import assert_step_bit_reproducible
    from qflux_amd.trainer import FluxKontextTrainStep
    _, hip, FO = _flux_pair()
    g = torch.Generator().manual_seed(53)
    T = 512
    specs = [((20, 20), [(20, 20)]), ((40, 40), [(40, 40)])]
    samples = []
    for (h, w), ctl in specs:
        n_t, n_c = h * w, sum(a * b for a, b in ctl)
        samples.append(dict(image_latents=torch.randn(n_t, 64, generator=g).half(), control_latents=torch.randn(n_c, 64, generator=g).half(),
                            hw=(h, w), control_hw=ctl, noise=torch.randn(n_t, 64, generator=g).to(BF), t=torch.rand((), generator=g).to(BF)))
    txt = dict(text_ids=torch.zeros(T, 3), pooled_prompt_embeds=torch.randn(2, 768, generator=g).half(),
               prompt_embeds=torch.randn(2, T, 4096, generator=g).half())
    step = FluxKontextTrainStep(hip)
    step.forward_backward_multires(samples, txt); step.zero_grad()
    plan = [p for k, p in hip._plans.items() if "multires" in k][0]
    assert_step_bit_reproducible(plan, lambda: step.forward_backward_multires(samples, txt), hip.lora_store.gflat, step.zero_grad, "FLUX ragged 320^2 + 640^2")


def test_cfg1_literal_flux_r4_two_double_blocks_256sq_step_vs_oracle():
    """VERDICT r4 #3d: BASELINE.json configs[0] at its literal shape -- FLUX-Kontext LoRA r = 4, 2 double blocks (+ 1 single block, the
    smallest trunk the single-block program exists for), 256 x 256 target + one control (S_i = 2 x 256 tokens), T = 512, batch 1, full
    width (D = 3072, joint dim 4096) -- one shared-mode step against the bf16 oracle (the reference runs this config on CPU in fp32 as
    plumbing: configs/example_fluxkontext_fp16.yaml)."""
    from oracle import flux_dit as FO
    from oracle import qwen_dit as O
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import FluxKontextTrainStep
    cfg = dict(FLUX_FULL, num_layers=2, num_single_layers=1)
    r = 4
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**cfg)
    g = torch.Generator(device=DEV).manual_seed(19)
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if p.ndim == 1 and "norm" in n:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=DEV) * 0.02).to(p.dtype))
    hip.add_adapter(LoraConfig(r=r, lora_alpha=r), "default", generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for n, p in hip.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(8)).to(p.device) * 1e-2)
    oracle = FO.OracleFluxDiT(**cfg)
    O.add_lora(oracle, r=r, lora_alpha=r, adapter_name="default")
    oracle.load_state_dict({k: v.float().cpu() for k, v in hip.state_dict().items()}, strict=True)
    for n, p in oracle.named_parameters():
        if "lora" not in n:
            p.data = p.data.to(BF)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    gg = torch.Generator().manual_seed(31)
    B, h, w, T = 1, 16, 16, 512           # 256^2 px -> 16 x 16 packed-latent tokens
    S_t = h * w
    ctl = FO.prepare_latent_image_ids(h, w)
    ctl[:, 0] = 1
    emb = dict(image_latents=torch.randn(B, S_t, 64, generator=gg).half(), control_latents=torch.randn(B, S_t, 64, generator=gg).half(),
               control_ids=ctl, text_ids=torch.zeros(T, 3), latent_hw=(h, w), pooled_prompt_embeds=torch.randn(B, 768, generator=gg).half(),
               prompt_embeds=torch.randn(B, T, 4096, generator=gg).half())
    noise = torch.randn(B, S_t, 64, generator=gg).to(BF)
    t = torch.tensor([0.7109]).to(BF)
    loss_o, pred_o = FO.flux_compute_loss(oracle, dict(emb, control_latents=emb["control_latents"].to(BF)), noise, t, BF, return_pred=True)
    loss_o.float().backward()
    step = FluxKontextTrainStep(hip)
    loss_h = step.forward_backward(emb, noise=noise, t=t).item()
    plan = list(hip._plans.values())[0]
    assert plan.S_i == 2 * S_t and plan.T == T
    e = _rel(plan.A["out"].view(B, -1, 64)[:, :S_t].cpu(), pred_o)
    c, rg, n = _grad_report(oracle, hip)
    print(f"cfg #1 literal (FLUX r=4, 2 double + 1 single, 256^2, T=512): loss {loss_h:.5f} / {loss_o.item():.5f}, pred rel {e:.4f}, LoRA grads n={n} min cos {c:.4f} worst rel {rg:.4f}")
    assert abs(loss_h - loss_o.item()) / abs(loss_o.item()) < 1e-2 and e < 2e-2 and c > 0.995 and rg < 2.5e-2 and n >= 12
