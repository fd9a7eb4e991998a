"""GPU parity of the FLUX-Kontext hot path (double + single blocks) against the CPU oracle and the reference's vectors."""
import pytest
import torch

from parity_util import BF, relmax, run_flux_step_parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_flux_tiny_step_fused_matches_oracle():
    res = run_flux_step_parity(DEV, verbose=True)
    assert res["ok"], res


def test_flux_tiny_step_autograd_path_no_guidance():
    res = run_flux_step_parity(DEV, verbose=True, guidance=False, fused=False)
    assert res["ok"], res


def test_flux_both_stream_lora_b1_odd_sizes():
    res = run_flux_step_parity(DEV, verbose=True, hw=(5, 7), T=11, B=1, r=8,
                               targets=("to_k", "to_q", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out"))
    assert res["ok"], res


def test_flux_feed_forward_lora_targets():
    """FLUX double blocks with adapters on ff / ff_context (part of the reference's broad regex config,
    configs/face_seg_flux_kontext_fp16.yaml:11) next to attention adapters on double and single blocks."""
    res = run_flux_step_parity(DEV, verbose=True, hw=(6, 4), T=9, B=2, r=8,
                               targets=("to_q", "to_v", "to_out.0", "to_add_out", "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj",
                                        "ff_context.net.2"))
    assert res["ok"], res


def test_flux_head_dim_128():
    cfg = dict(patch_size=1, in_channels=64, out_channels=64, num_layers=1, num_single_layers=2, attention_head_dim=128,
               num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
    res = run_flux_step_parity(DEV, verbose=True, cfg=cfg, hw=(8, 8), T=16, B=2, r=16)
    assert res["ok"], res


def test_flux_optimizer_steps_reduce_loss():
    from common import FLUX_TINY
    from oracle import flux_dit as FO
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import FluxKontextTrainStep
    cfg = dict(FLUX_TINY, joint_attention_dim=64, guidance_embeds=True)
    with torch.device(DEV):
        m = FluxTransformer2DModel(**cfg)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_((torch.randn(p.shape, generator=g) * (0.5 / p.shape[-1] ** 0.5 if p.ndim == 2 else 0.05) + (1.0 if "norm_" in n and p.ndim == 1 else 0.0)).to(p.dtype))
    m.add_adapter(LoraConfig(r=4, lora_alpha=8), "a", generator=g)
    step = FluxKontextTrainStep(m, lr=3e-3, weight_decay=0.0)
    ctl = FO.prepare_latent_image_ids(4, 6); ctl[:, 0] = 1
    emb = dict(image_latents=torch.randn(2, 24, 64, generator=g).half(), control_latents=torch.randn(2, 24, 64, generator=g).half(),
               control_ids=ctl, text_ids=torch.zeros(7, 3), latent_hw=(4, 6),
               pooled_prompt_embeds=torch.randn(2, 16, generator=g).half(), prompt_embeds=torch.randn(2, 7, 64, generator=g).half())
    noise = torch.randn(2, 24, 64, generator=g)
    t = torch.tensor([0.3, 0.8])
    losses = [step.train_step(emb, noise=noise, t=t).item() for _ in range(8)]
    print(losses)
    assert losses[-1] < losses[0]


def test_flux_forward_matches_reference_golden_vectors(golden_dir):
    """bf16 HIP forward vs the fp32 output captured from the reference's transformer_flux.py (tests/golden/flux_tiny_fwd)."""
    import os
    from safetensors.torch import load_file
    from common import FLUX_TINY, fill_weights
    from oracle import flux_dit as FO
    from qflux_amd.models import FluxTransformer2DModel
    t = load_file(os.path.join(golden_dir, "flux_tiny_fwd.safetensors"))
    cfg = dict(FLUX_TINY, guidance_embeds=True)
    oracle = FO.OracleFluxDiT(**cfg)
    fill_weights(oracle, seed=3)
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**cfg)
    hip.load_state_dict({k: v.to(BF) for k, v in oracle.state_dict().items()}, strict=True)
    with torch.no_grad():
        out = hip(hidden_states=t["in.hidden_states"].to(DEV).to(BF), encoder_hidden_states=t["in.encoder_hidden_states"].to(DEV).to(BF),
                  pooled_projections=t["in.pooled"].to(DEV).to(BF), timestep=t["in.timestep"].to(DEV), img_ids=t["in.img_ids"],
                  txt_ids=t["in.txt_ids"], guidance=torch.ones(2, device=DEV), return_dict=False)[0]
    e = relmax(out, t["out.sample"])
    print("flux hip bf16 vs reference fp32 golden: rel", e)
    assert e < 5e-2


def _multires_case(seed=51, jd=64, pooled=16):
    from oracle import flux_dit as FO  # noqa: F401
    g = torch.Generator().manual_seed(seed)
    T = 7
    shapes = [((4, 6), [(4, 6)]), ((3, 5), [(4, 4)]), ((5, 5), [(2, 3), (3, 3)])]
    samples = []
    for (h, w), ctl in shapes:
        n_t, n_c = h * w, sum(a * b for a, b in ctl)
        samples.append(dict(image_latents=torch.randn(n_t, 64, generator=g).half(), control_latents=torch.randn(n_c, 64, generator=g).half(),
                            hw=(h, w), control_hw=ctl, noise=torch.randn(n_t, 64, generator=g).to(BF),
                            t=torch.rand((), generator=g).to(BF)))
    txt = dict(text_ids=torch.zeros(T, 3), pooled_prompt_embeds=torch.randn(3, pooled, generator=g).half(),
               prompt_embeds=torch.randn(3, T, jd, generator=g).half())
    return samples, txt


@pytest.mark.parametrize("fused", [True, False])
def test_flux_multires_step_matches_oracle(fused):
    """cfg #5 flavour: ragged batch (3 different target sizes, 1-2 control images each), right-padded; per-sample RoPE,
    additive key mask, padded rows exactly zero; LoRA grads vs the oracle of the reference's custom model."""
    from common import FLUX_TINY, fill_weights
    from oracle import flux_dit as FO
    from oracle import qwen_dit as O
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import FluxKontextTrainStep
    cfg = dict(FLUX_TINY, guidance_embeds=True, joint_attention_dim=64)
    oracle = FO.OracleFluxDiT(**cfg)
    O.add_lora(oracle, r=4, lora_alpha=8, adapter_name="a")
    fill_weights(oracle, seed=6)
    for n, p in oracle.named_parameters():
        if "lora" not in n:
            p.data = p.data.to(BF)
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**cfg)
    hip.add_adapter(LoraConfig(r=4, lora_alpha=8), "a")
    hip.load_state_dict(oracle.state_dict(), strict=True)
    samples, txt = _multires_case()
    so = [dict(s, control_latents=s["control_latents"].to(BF)) for s in samples]
    loss_o, pred_o = FO.flux_compute_loss_multires(oracle, so, txt, BF, return_pred=True)
    loss_o.float().backward()
    step = FluxKontextTrainStep(hip)
    if fused:
        loss_h = step.forward_backward_multires(samples, txt)
    else:
        loss_h = step.compute_loss_multires(samples, txt)
        loss_h.backward()
    plan = [p for k, p in hip._plans.items() if "multires" in k][0]
    out = plan.A["out"].view(3, -1, 64)
    n_t_max = pred_o.shape[1]
    e_pred = relmax(out[:, :n_t_max], pred_o)
    lens = [24 + 24, 15 + 16, 25 + 6 + 9]
    for b, L in enumerate(lens):
        assert out[b, L:].numel() == 0 or out[b, L:].abs().max().item() == 0.0          # padded outputs exactly zero (test_qwen_custom.py:304-305 analogue)
    og = {n: p.grad for n, p in oracle.named_parameters() if "lora" in n}
    worst = max(relmax(p.grad, og[n]) for n, p in hip.named_parameters() if "lora" in n and og[n] is not None)
    print("multires", fused, "loss", loss_o.item(), loss_h.item(), "pred rel", e_pred, "grad worst", worst)
    assert abs(loss_h.item() - loss_o.item()) / abs(loss_o.item()) < 2e-2
    assert e_pred < 2e-2 and worst < 8e-2


def test_multires_plan_cache_is_bounded_over_20_distinct_ragged_batches():
    """cfg #5: bucketed batches arrive with a continuum of padded lengths.  The plans live on a ladder of lengths in an LRU cache
    with a byte budget (plan_cache.py): 20 distinct ragged batches build only a few plans, device memory stops growing, and a
    batch that re-uses a ladder size gives the same loss as on its first visit (no stale per-batch state in a cached plan)."""
    import gc
    import random
    from common import FLUX_TINY
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.plan_cache import PlanCache, ladder
    from qflux_amd.trainer import FluxKontextTrainStep
    cfg = dict(FLUX_TINY, guidance_embeds=True, joint_attention_dim=64)
    with torch.device(DEV):
        m = FluxTransformer2DModel(**cfg)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_((torch.randn(p.shape, generator=g) * (0.5 / p.shape[-1] ** 0.5 if p.ndim == 2 else 0.05) + (1.0 if "norm_" in n and p.ndim == 1 else 0.0)).to(p.dtype))
    m.add_adapter(LoraConfig(r=4, lora_alpha=8), "a", generator=g)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lora_B" in n:
                p.normal_(0, 0.02)
    step = FluxKontextTrainStep(m)
    rng = random.Random(5)
    T = 7

    def batch(seed):
        gg = torch.Generator().manual_seed(seed)
        r = random.Random(seed)
        samples = []
        for _ in range(2):
            h, w = r.randint(4, 22), r.randint(4, 22)
            ch, cw = r.randint(4, 22), r.randint(4, 22)
            samples.append(dict(image_latents=torch.randn(h * w, 64, generator=gg).half(), control_latents=torch.randn(ch * cw, 64, generator=gg).half(),
                                hw=(h, w), control_hw=[(ch, cw)], noise=torch.randn(h * w, 64, generator=gg).to(BF),
                                t=torch.rand((), generator=gg).to(BF)))
        txt = dict(text_ids=torch.zeros(T, 3), pooled_prompt_embeds=torch.randn(2, 16, generator=gg).half(),
                   prompt_embeds=torch.randn(2, T, 64, generator=gg).half())
        return samples, txt

    first = {}
    lens = set()
    mem = []
    for i in range(20):
        samples, txt = batch(100 + i)
        lens.add(max(s["image_latents"].shape[0] + s["control_latents"].shape[0] for s in samples))
        first[i] = step.forward_backward_multires(samples, txt).item()
        step.zero_grad()
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated())
    plans = m._plans
    assert isinstance(plans, PlanCache)
    sizes = {ladder(n) for n in lens}
    print("distinct padded lengths", len(lens), "ladder sizes", len(sizes), "plans built", plans.builds, "cached", len(plans),
          "MB", [round(x / 2**20) for x in mem])
    assert len(lens) >= 15 and plans.builds == len(sizes) <= 8
    # replay: same batches -> no new plan, identical losses (fp32 atomics in the loss reduction aside), no memory growth
    for i in range(20):
        samples, txt = batch(100 + i)
        l2 = step.forward_backward_multires(samples, txt).item()
        step.zero_grad()
        assert abs(l2 - first[i]) <= 1e-5 * abs(first[i]), (i, l2, first[i])
    torch.cuda.synchronize()
    assert plans.builds == len(sizes) and torch.cuda.memory_allocated() <= max(mem) * 1.02
    # a tight byte budget evicts least-recently-used plans and gives their arena back
    biggest = max(plans.sizes.values())
    plans._budget = int(biggest * 1.5)
    plans._evict(keep=next(reversed(plans)))
    gc.collect()
    torch.cuda.synchronize()
    assert plans.total_bytes() <= plans._budget and len(plans) < len(sizes)
    assert torch.cuda.memory_allocated() < max(mem)
    samples, txt = batch(100)
    assert abs(step.forward_backward_multires(samples, txt).item() - first[0]) <= 1e-5 * abs(first[0])     # rebuilt on demand


# the reference's broad FLUX target regex (configs/face_seg_flux_kontext_fp16.yaml:11) WITHOUT its modulation alternatives
# ((norm|norm1|norm1_context).linear -- GEMV sites, covered by test_flux_reference_regex_with_modulation_targets)
_REGEX_GEMM_SITES = (r"(.*x_embedder|.*transformer_blocks\.[0-9]+\.attn\.(to_k|to_q|to_v|to_add_out)|.*transformer_blocks\.[0-9]+\.attn\.to_out\.0|"
                     r".*single_transformer_blocks\.[0-9]+\.attn\.to_out|.*single_transformer_blocks\.[0-9]+\.(proj_mlp|proj_out)|"
                     r".*(?<!single_)transformer_blocks\.[0-9]+\.ff\.net\.2|.*(?<!single_)transformer_blocks\.[0-9]+\.ff\.net\.0\.proj|"
                     r".*(?<!single_)transformer_blocks\.[0-9]+\.ff_context\.net\.0\.proj|.*(?<!single_)transformer_blocks\.[0-9]+\.ff_context\.net\.2|"
                     r".*(?<!single_)transformer_blocks\.[0-9]+\.attn\.(to_add_out|add_k_proj|add_q_proj|add_v_proj))")


def test_flux_reference_regex_gemm_sites():
    """Embedder, double-block attention + feed-forward of both streams, single-block q/k/v + proj_mlp + proj_out adapters."""
    res = run_flux_step_parity(DEV, verbose=True, hw=(6, 4), T=9, B=2, r=8, targets=_REGEX_GEMM_SITES)
    assert res["ok"] and res["n_lora"] == 2 * (1 + 2 * 12 + 2 * 5), res


def test_flux_embedders_output_projection_and_single_block_sites():
    """List-style targets: x_embedder, context_embedder, the output projection and every single-block linear (suffix `proj_out`
    matches the model's output projection AND the single blocks' proj_out, as peft's suffix matching does)."""
    res = run_flux_step_parity(DEV, verbose=True, hw=(5, 6), T=8, B=2, r=4,
                               targets=("x_embedder", "context_embedder", "proj_out", "proj_mlp", "to_q", "to_v"))
    assert res["ok"] and res["n_lora"] == 2 * (2 + 1 + 2 * 2 + 2 * 2 + 2 * 2), res


# configs/face_seg_flux_kontext_fp16.yaml:11, verbatim
_REFERENCE_REGEX = (r"(.*x_embedder|.*transformer_blocks\.[0-9]+\.(norm|norm1)\.linear|.*transformer_blocks\.[0-9]+\.attn\.(to_k|to_q|to_v|to_add_out)|"
                    r".*transformer_blocks\.[0-9]+\.attn\.to_out\.0|.*single_transformer_blocks\.[0-9]+\.attn\.to_out|"
                    r".*single_transformer_blocks\.[0-9]+\.(proj_mlp|proj_out)|.*(?<!single_)transformer_blocks\.[0-9]+\.ff\.net\.2|"
                    r".*(?<!single_)transformer_blocks\.[0-9]+\.ff\.net\.0\.proj|.*(?<!single_)transformer_blocks\.[0-9]+\.norm1_context\.linear|"
                    r".*(?<!single_)transformer_blocks\.[0-9]+\.ff_context\.net\.0\.proj|.*(?<!single_)transformer_blocks\.[0-9]+\.ff_context\.net\.2|"
                    r".*(?<!single_)transformer_blocks\.[0-9]+\.attn\.(to_add_out|add_k_proj|add_q_proj|add_v_proj))")


def test_flux_reference_regex_with_modulation_targets():
    """The reference's shipped broad regex as is: GEMM sites + norm1 / norm1_context / single-block norm linears."""
    res = run_flux_step_parity(DEV, verbose=True, hw=(6, 4), T=9, B=2, r=8, targets=_REFERENCE_REGEX)
    assert res["ok"] and res["n_lora"] == 2 * (1 + 2 * 14 + 2 * 6), res


def test_flux_all_linear_targets():
    res = run_flux_step_parity(DEV, verbose=True, hw=(4, 6), T=7, B=2, r=4, targets="all-linear")
    assert res["ok"], res
    res = run_flux_step_parity(DEV, verbose=True, hw=(4, 6), T=7, B=1, r=4, targets="all-linear", guidance=False, fused=False)
    assert res["ok"], res


@pytest.mark.parametrize("true_cfg", [1.0, 3.0])
def test_flux_sampling_loop_matches_oracle(true_cfg):
    """f4: FLUX-Kontext validation sampler (flux_kontext_trainer.py:902-976) -- 4 guidance-distilled Euler steps, optional true CFG
    without norm rescale -- on the training launch programs (inference mode, LoRA applied) vs the oracle's restatement."""
    from common import FLUX_TINY, fill_weights
    from oracle import flux_dit as FO
    from oracle import qwen_dit as O
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.sampling import FluxSampler
    cfg = dict(FLUX_TINY, guidance_embeds=True, joint_attention_dim=64)
    oracle = FO.OracleFluxDiT(**cfg)
    O.add_lora(oracle, r=4, lora_alpha=8, adapter_name="a")
    fill_weights(oracle, seed=6)
    for n, p in oracle.named_parameters():
        if "lora" not in n:
            p.data = p.data.to(BF)
    with torch.device(DEV):
        hip = FluxTransformer2DModel(**cfg)
    hip.add_adapter(LoraConfig(r=4, lora_alpha=8), "a")
    hip.load_state_dict(oracle.state_dict(), strict=True)
    g = torch.Generator().manual_seed(9)
    B, h, w, T = 2, 4, 6, 7
    S_t = h * w
    lat_ids = FO.prepare_latent_image_ids(h, w)
    ctl_ids = FO.prepare_latent_image_ids(h, w)
    ctl_ids[:, 0] = 1
    emb = dict(latents=torch.randn(B, S_t, 64, generator=g), latent_ids=lat_ids, control_latents=torch.randn(B, S_t, 64, generator=g).half().float(),
               control_ids=ctl_ids, pooled_prompt_embeds=torch.randn(B, cfg["pooled_projection_dim"], generator=g).half().float(),
               prompt_embeds=torch.randn(B, T, 64, generator=g).half().float(), text_ids=torch.zeros(T, 3), guidance=2.5,
               num_inference_steps=4, true_cfg_scale=true_cfg)
    if true_cfg > 1.0:
        emb.update(negative_pooled_prompt_embeds=torch.randn(B, cfg["pooled_projection_dim"], generator=g).half().float(),
                   negative_prompt_embeds=torch.randn(B, T, 64, generator=g).half().float(), negative_text_ids=torch.zeros(T, 3))
    ref = FO.flux_sample(oracle, emb, BF)
    out = FluxSampler(hip).sample(emb)
    rel = ((out.float().cpu() - ref.float()).abs().max() / ref.float().abs().max()).item()
    print(f"flux sampling 4 steps, true_cfg {true_cfg}: rel", rel)
    assert out.shape == ref.shape and rel < 3e-2
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for n, p in hip.named_parameters() if "lora" in n)
