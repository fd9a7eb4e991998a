"""Parity at FULL DEPTH: the 60-block Qwen-Image-Edit DiT at BASELINE.json configs[1] (D = 3072, S_i = 2048, T = 384,
r = 16) and the 19 + 38-block FLUX-Kontext DiT, one whole LoRA step each, plus a short TRAINING RUN at the headline size.

The oracle (plain torch, oracle/*.py) is the checker and runs ON THE GPU here -- the only place where a 20 B-parameter eager graph
finishes in seconds (82 GB of fp32 weights next to 288 GB of HBM).  That is test-only use of torch compute on the device: the
product path never touches it.  Three runs per model on identical weights and draws: the oracle in fp32 (the truth), the oracle in
the reference's training layout (bf16 trunk + activations, fp32 adapters: what diffusers computes), and the HIP launch programs.
A 60-block bf16 graph is a chaotic map of its rounding errors, so the bar is relative: the HIP path must sit as close to the fp32
truth as the eager bf16 graph does (same rounding points, different summation orders) -- for the prediction, for the residual stream
every 10 blocks (read back from the plan's arena) and for the LoRA gradients of EVERY block (the arena offsets of blocks 2..59, the
side-stream join two blocks later and the block-parity scratch alternation are only exercised at this depth)."""
import json
import os
import sys
import time
import zlib

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "qwen-image-finetune_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
QWEN_FULL = dict(patch_size=2, in_channels=64, out_channels=16, attention_head_dim=128, num_attention_heads=24, joint_attention_dim=3584,
                 axes_dims_rope=(16, 56, 56))
FLUX_FULL = dict(patch_size=1, in_channels=64, out_channels=None, attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096,
                 pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56))


def _dump(name, obj):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, f"parity_{name}.json"), "w") as f:
            json.dump(obj, f, indent=1)


def _fill_named(model, lora_b_std):
    """Every parameter from a generator seeded by its NAME (the HIP modules and the oracle share the reference's state-dict names), so
    each of the three models regenerates identical weights on the device without an 80 GB host copy: trunk weights N(0, 0.02^2)
    rounded to bf16 (exactly representable in the fp32 oracle too), norm weights 1, lora_A ~ N(0, (1/r)^2) (peft "gaussian"),
    lora_B ~ N(0, lora_b_std^2) (peft initialises it to zero; the gradient check needs dA != 0)."""
    with torch.no_grad():
        for n, p in model.named_parameters():
            g = torch.Generator(device=p.device).manual_seed(zlib.crc32(n.encode()))
            if "lora_A" in n:
                v = torch.randn(p.shape, generator=g, device=p.device) / p.shape[0]
            elif "lora_B" in n:
                v = torch.randn(p.shape, generator=g, device=p.device) * lora_b_std
            elif p.ndim == 1 and "norm" in n and n.endswith("weight"):
                v = torch.ones(p.shape, device=p.device)
            else:
                v = (torch.randn(p.shape, generator=g, device=p.device) * 0.02).to(BF).float()
            p.copy_(v.to(p.dtype))


def _free():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _relmax(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()


def _block_of(name):
    parts = name.split(".")
    return (parts[0], int(parts[1])) if parts[0] in ("transformer_blocks", "single_transformer_blocks") else ("head", 0)


def _per_block_cos(ga, gb):
    """LoRA gradients grouped per DiT block (all adapter tensors of a block concatenated): cosine of a against b per block."""
    groups = {}
    for n in ga:
        if gb.get(n) is None or ga.get(n) is None:
            continue
        groups.setdefault(_block_of(n), []).append(n)
    out = {}
    for k, names in groups.items():
        a = torch.cat([ga[n].float().flatten().cpu() for n in sorted(names)])
        b = torch.cat([gb[n].float().flatten().cpu() for n in sorted(names)])
        if b.abs().max() > 0:
            out[k] = _cos(a, b)
    return out


def _to_dev(emb):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in emb.items()}


def _need_memory(gb):
    _free()
    free, total = torch.cuda.mem_get_info()
    assert free > gb * (1 << 30), f"this test needs ~{gb} GB of free HBM on {DEV} (free {free >> 30} GB of {total >> 30} GB)"


# ============================================================================================== Qwen-Image-Edit, 60 blocks
def _qwen_emb(seed, side=32, T=384):
    g = torch.Generator().manual_seed(seed)
    S_t = side * side
    emb = dict(image_latents=torch.randn(1, S_t, 64, generator=g).half().float(), control_latents=torch.randn(1, S_t, 64, generator=g).half().float(),
               prompt_embeds=(torch.randn(1, T, 3584, generator=g) * 4).half().float(), prompt_embeds_mask=torch.ones(1, T, dtype=torch.int64),
               img_shapes=[[(1, side, side), (1, side, side)]])
    return emb, torch.randn(1, S_t, 64, generator=g), torch.rand(1, generator=g)


def _oracle_run(oracle, blocks, hook_ids, loss_fn, pick):
    """One forward + backward of the oracle with the residual stream captured after the blocks in hook_ids."""
    caps, hs = {}, []
    for i in hook_ids:
        hs.append(blocks[i].register_forward_hook(lambda m, a, out, i=i: caps.__setitem__(i, pick(out).detach().float().cpu())))
    for p in oracle.parameters():
        p.grad = None
    t0 = time.time()
    loss, pred = loss_fn()
    loss.float().backward()
    torch.cuda.synchronize()
    dt = time.time() - t0
    for h in hs:
        h.remove()
    grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in oracle.named_parameters() if "lora" in n}
    return dict(loss=loss.item(), pred=pred.detach().float().cpu(), hid=caps, grads=grads, seconds=dt)


def test_qwen_60_blocks_step_and_training_run_vs_oracle_on_gpu():
    from oracle import qwen_dit as O
    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import QwenLoraTrainStep
    L, r = 60, 16
    _need_memory(200)
    emb, noise, u = _qwen_emb(5)
    u = torch.tensor([0.7109])                      # the reference's e2e constant (tests/e2e/test_flux_loss.py:119)
    emb_d, noise_d = _to_dev(emb), noise.to(DEV)
    hook_ids = [9, 19, 29, 39, 49, 59]

    # ---- the oracle on the device: fp32 (truth), then the SAME module with its trunk cast to bf16 (the reference's training layout)
    with torch.device(DEV):
        oracle = O.OracleQwenDiT(num_layers=L, **QWEN_FULL)
        O.add_lora(oracle, r=r, lora_alpha=r, adapter_name="default")
    _fill_named(oracle, 1e-2)
    ref = {}
    for tag, dt in (("fp32", torch.float32), ("bf16", BF)):
        if dt == BF:
            for n, p in oracle.named_parameters():
                if "lora" not in n:
                    p.data = p.data.to(BF)        # exact: the values are bf16-representable
            _free()
        ref[tag] = _oracle_run(oracle, oracle.transformer_blocks, hook_ids,
                               lambda: O.qwen_compute_loss(oracle, emb_d, noise_d, u, dt, return_pred=True), lambda out: out[1])
    _free()

    # ---- the HIP launch programs on the same weights
    with torch.device(DEV):
        hip = QwenImageTransformer2DModel(num_layers=L, **QWEN_FULL)
    hip.add_adapter(LoraConfig(r=r, lora_alpha=r), "default")
    _fill_named(hip, 1e-2)
    hip.refresh_lora_operands()
    step = QwenLoraTrainStep(hip, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)
    loss_h = step.forward_backward(emb, noise=noise, u=u).item()
    torch.cuda.synchronize()
    plan = list(hip._plans.values())[0]
    S_t = emb["image_latents"].shape[1]
    pred_h = plan.A["out"].view(1, -1, 64)[:, :S_t].float().cpu()
    hid_h = {i: plan.A["X"]["img"][i + 1].view(1, -1, hip.inner_dim).float().cpu() for i in hook_ids}
    grads_h = {n: p.grad.detach().float().cpu().clone() for n, p in hip.named_parameters() if "lora" in n}
    step.zero_grad()

    rep = dict(config="Qwen-Image-Edit 60 blocks, D=3072, S_i=2048, T=384, r=16, B=1", loss=dict(hip=loss_h, bf16=ref["bf16"]["loss"], fp32=ref["fp32"]["loss"]),
               oracle_seconds=dict(fp32=ref["fp32"]["seconds"], bf16=ref["bf16"]["seconds"]))
    rep["pred_rel_l2"] = dict(hip_vs_bf16=_rel(pred_h, ref["bf16"]["pred"]), hip_vs_fp32=_rel(pred_h, ref["fp32"]["pred"]),
                              bf16_vs_fp32=_rel(ref["bf16"]["pred"], ref["fp32"]["pred"]))
    rep["pred_rel_max"] = dict(hip_vs_bf16=_relmax(pred_h, ref["bf16"]["pred"]), hip_vs_fp32=_relmax(pred_h, ref["fp32"]["pred"]),
                               bf16_vs_fp32=_relmax(ref["bf16"]["pred"], ref["fp32"]["pred"]))
    rep["residual_rel_l2_after_block"] = {
        str(i + 1): dict(hip_vs_fp32=_rel(hid_h[i], ref["fp32"]["hid"][i]), bf16_vs_fp32=_rel(ref["bf16"]["hid"][i], ref["fp32"]["hid"][i]),
                         hip_vs_bf16=_rel(hid_h[i], ref["bf16"]["hid"][i])) for i in hook_ids}
    c_hf = _per_block_cos(grads_h, ref["fp32"]["grads"])
    c_bf = _per_block_cos(ref["bf16"]["grads"], ref["fp32"]["grads"])
    c_hb = _per_block_cos(grads_h, ref["bf16"]["grads"])
    rep["lora_grad_cos_per_block"] = {f"{k[0]}.{k[1]}": dict(hip_vs_fp32=c_hf[k], bf16_vs_fp32=c_bf[k], hip_vs_bf16=c_hb[k]) for k in sorted(c_hf)}
    rep["lora_grad_cos_min"] = dict(hip_vs_fp32=min(c_hf.values()), bf16_vs_fp32=min(c_bf.values()), hip_vs_bf16=min(c_hb.values()))
    gn = lambda g: float(torch.sqrt(sum((v.double() ** 2).sum() for v in g.values() if v is not None)))   # noqa: E731
    rep["lora_grad_global_norm"] = dict(hip=gn(grads_h), bf16=gn(ref["bf16"]["grads"]), fp32=gn(ref["fp32"]["grads"]))
    print("\n[full depth, Qwen 60 blocks] " + json.dumps({k: v for k, v in rep.items() if k != "lora_grad_cos_per_block"}))
    del ref["fp32"]["hid"], ref["bf16"]["hid"]

    # one step: bars relative to what the eager bf16 graph itself does against fp32
    assert len(c_hf) == L and all(v is not None for v in grads_h.values())
    assert abs(loss_h - ref["bf16"]["loss"]) / abs(ref["bf16"]["loss"]) < 1e-2          # |dloss| bar of the reference's e2e tests
    assert abs(loss_h - ref["fp32"]["loss"]) <= 1.5 * abs(ref["bf16"]["loss"] - ref["fp32"]["loss"]) + 1e-3 * abs(ref["fp32"]["loss"])
    pr = rep["pred_rel_l2"]
    assert pr["hip_vs_fp32"] < 1.25 * pr["bf16_vs_fp32"] + 1e-3, pr
    for i in hook_ids:
        d = rep["residual_rel_l2_after_block"][str(i + 1)]
        assert d["hip_vs_fp32"] < 1.25 * d["bf16_vs_fp32"] + 1e-3, (i, d)
    for k in c_hf:
        assert c_hf[k] > c_bf[k] - 0.02, (k, c_hf[k], c_bf[k])
    g_n = rep["lora_grad_global_norm"]
    assert abs(g_n["hip"] - g_n["fp32"]) <= 1.5 * abs(g_n["bf16"] - g_n["fp32"]) + 2e-2 * g_n["fp32"], g_n
    # ... and the TIGHT bars (VERDICT r4 #3a): the HIP path against the eager bf16 graph it restates -- same rounding points, different
    # summation orders -- at 2x what round 4 observed (profiles/r04_parity_fulldepth_qwen60.json: prediction 1.96e-2, residual stream
    # 1.0-1.9e-2, per-block gradient cosine 1.00000, global gradient norm 2.8440 vs 2.8448).  A regression of the HIP path to 0.15 of
    # the bf16 graph would pass every relative bar above; it fails these.
    assert pr["hip_vs_bf16"] < 4e-2, pr
    for i in hook_ids:
        d = rep["residual_rel_l2_after_block"][str(i + 1)]
        assert d["hip_vs_bf16"] < 4e-2, (i, d)
    for k in c_hb:
        assert c_hb[k] > 0.9995, (k, c_hb[k])
    assert abs(g_n["hip"] - g_n["bf16"]) / g_n["bf16"] < 2e-3, g_n

    # ---- a TRAINING RUN at the headline size (north_star: loss curve within 1e-3 MSE of the reference): the fused HIP step (forward,
    # backward, clip, AdamW) against the bf16 oracle trained by torch.optim.AdamW + clip_grad_norm_ on the same draws, from peft's init
    # (lora_B = 0), rotating over three synthetic cached samples with fresh noise / timestep draws per step
    steps, lr, wd = 12, 1e-4, 0.01
    with torch.no_grad():
        for m_ in (oracle, hip):
            for n, p in m_.named_parameters():
                if "lora_B" in n:
                    p.zero_()
    hip.refresh_lora_operands()
    params = [p for n, p in oracle.named_parameters() if "lora" in n]
    opt = torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    pool = [_qwen_emb(100 + i) for i in range(3)]
    g = torch.Generator().manual_seed(7)
    lo, lh, secs = [], [], [0.0, 0.0]
    for it in range(steps):
        e = pool[it % len(pool)][0]
        nz = torch.randn(e["image_latents"].shape, generator=g)
        uu = torch.rand(1, generator=g)
        torch.cuda.synchronize(); t0 = time.time()
        loss_o = O.qwen_compute_loss(oracle, _to_dev(e), nz.to(DEV), uu, BF)
        opt.zero_grad(set_to_none=True)
        loss_o.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        lo.append(loss_o.item()); t1 = time.time()
        lh.append(step.train_step(e, noise=nz, u=uu).item()); t2 = time.time()
        secs[0] += t1 - t0; secs[1] += t2 - t1
    d = torch.tensor(lh) - torch.tensor(lo)
    mse, worst = float((d ** 2).mean()), float(d.abs().max())
    got = {n: p.detach().float().cpu() for n, p in hip.named_parameters() if "lora" in n}
    want = {n: p.detach().float().cpu() for n, p in oracle.named_parameters() if "lora" in n}
    drift = max(_relmax(got[n], want[n]) for n in want)
    cosB = _cos(torch.cat([got[n].flatten() for n in sorted(got) if "lora_B" in n]), torch.cat([want[n].flatten() for n in sorted(want) if "lora_B" in n]))
    rep["training_run"] = dict(steps=steps, lr=lr, loss_curve_mse=mse, max_abs_loss_diff=worst, oracle=lo, hip=lh, adapter_rel_drift_max=drift,
                               lora_B_cosine=cosB, seconds_per_step=dict(oracle_eager_bf16_gpu=secs[0] / steps, hip_fused=secs[1] / steps))
    print("[full depth, Qwen 60 blocks] training run: " + json.dumps(rep["training_run"]))
    _dump("fulldepth_qwen60", rep)
    assert mse < 1e-3, (mse, worst)                       # the north-star tolerance, at the real configuration
    assert cosB > 0.98, cosB                             # the two runs learned the same update (lora_B starts at 0: it IS the update)

    # ---- run-to-run bit reproducibility of the fused step at full depth (VERDICT r4 #3c: cfg #2, 60 blocks, side-stream gradient launches,
    # block-parity scratch): every arena tensor of two identical steps from a zeroed arena
    del oracle, opt, params
    _free()
    from parity_util import assert_step_bit_reproducible
    n_t = assert_step_bit_reproducible(plan, lambda: step.forward_backward(emb, noise=noise, u=u), hip.lora_store.gflat, step.zero_grad, "cfg #2, 60 blocks")
    print(f"[full depth, Qwen 60 blocks] bit-reproducible over {n_t} arena tensors")


# ============================================================================================== FLUX-Kontext, 19 + 38 blocks
def test_flux_19_plus_38_blocks_step_vs_oracle_on_gpu():
    from oracle import flux_dit as FO
    from oracle import qwen_dit as O
    from qflux_amd.models import FluxTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd.trainer import FluxKontextTrainStep
    Ld, Ls, r, T, side = 19, 38, 16, 512, 32
    _need_memory(150)
    g = torch.Generator().manual_seed(31)
    S_t = side * side
    ctl_ids = FO.prepare_latent_image_ids(side, side)
    ctl_ids[:, 0] = 1
    emb = dict(image_latents=torch.randn(1, S_t, 64, generator=g).half(), control_latents=torch.randn(1, S_t, 64, generator=g).half(),
               control_ids=ctl_ids, text_ids=torch.zeros(T, 3), latent_hw=(side, side),
               pooled_prompt_embeds=torch.randn(1, 768, generator=g).half(), prompt_embeds=torch.randn(1, T, 4096, generator=g).half())
    noise = torch.randn(1, S_t, 64, generator=g).to(BF)
    t = torch.tensor([0.7109]).to(BF)
    emb_d = _to_dev(dict(emb, control_latents=emb["control_latents"].to(BF)))
    hook_d, hook_s = [Ld - 1], [9, 19, 29, Ls - 1]

    with torch.device(DEV):
        oracle = FO.OracleFluxDiT(num_layers=Ld, num_single_layers=Ls, **FLUX_FULL)
        O.add_lora(oracle, r=r, lora_alpha=r, adapter_name="default")
    _fill_named(oracle, 1e-2)
    ref = {}
    for tag, dt in (("fp32", torch.float32), ("bf16", BF)):
        if dt == BF:
            for n, p in oracle.named_parameters():
                if "lora" not in n:
                    p.data = p.data.to(BF)
            _free()
        caps = {}
        hs = [oracle.transformer_blocks[i].register_forward_hook(lambda m, a, out, i=i: caps.__setitem__(("d", i), out[1].detach().float().cpu()))
              for i in hook_d]
        hs += [oracle.single_transformer_blocks[i].register_forward_hook(lambda m, a, out, i=i: caps.__setitem__(("s", i), out[1].detach().float().cpu()))
               for i in hook_s]
        e_in = emb_d if dt == BF else {k: (v.float() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in emb_d.items()}
        for p in oracle.parameters():
            p.grad = None
        t0 = time.time()
        loss_o, pred_o = FO.flux_compute_loss(oracle, e_in, noise.to(DEV).to(dt), t.to(DEV).to(dt), dt, return_pred=True)
        loss_o.float().backward()
        torch.cuda.synchronize()
        for h in hs:
            h.remove()
        ref[tag] = dict(loss=loss_o.item(), pred=pred_o.detach().float().cpu(), hid=caps, seconds=time.time() - t0,
                        grads={n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in oracle.named_parameters() if "lora" in n})
    del oracle
    _free()

    with torch.device(DEV):
        hip = FluxTransformer2DModel(num_layers=Ld, num_single_layers=Ls, **FLUX_FULL)
    hip.add_adapter(LoraConfig(r=r, lora_alpha=r), "default")
    _fill_named(hip, 1e-2)
    hip.refresh_lora_operands()
    step = FluxKontextTrainStep(hip)
    loss_h = step.forward_backward(emb, noise=noise, t=t).item()
    torch.cuda.synchronize()
    plan = list(hip._plans.values())[0]
    D = hip.inner_dim
    pred_h = plan.A["out"].view(1, -1, plan.A["out"].shape[-1])[:, :S_t].float().cpu()
    hid_h = {("d", Ld - 1): plan.A["J"][0].view(1, -1, D)[:, T:].float().cpu()}
    for i in hook_s:
        hid_h[("s", i)] = plan.A["J"][i + 1].view(1, -1, D)[:, T:].float().cpu()
    grads_h = {n: p.grad.detach().float().cpu().clone() for n, p in hip.named_parameters() if "lora" in n}

    rep = dict(config="FLUX-Kontext 19 double + 38 single blocks, D=3072, S_i=2048, T=512, r=16, B=1, guidance embeds",
               loss=dict(hip=loss_h, bf16=ref["bf16"]["loss"], fp32=ref["fp32"]["loss"]), oracle_seconds=dict(fp32=ref["fp32"]["seconds"], bf16=ref["bf16"]["seconds"]))
    rep["pred_rel_l2"] = dict(hip_vs_bf16=_rel(pred_h, ref["bf16"]["pred"]), hip_vs_fp32=_rel(pred_h, ref["fp32"]["pred"]),
                              bf16_vs_fp32=_rel(ref["bf16"]["pred"], ref["fp32"]["pred"]))
    rep["residual_rel_l2"] = {f"{k[0]}{k[1] + 1}": dict(hip_vs_fp32=_rel(hid_h[k], ref["fp32"]["hid"][k]), bf16_vs_fp32=_rel(ref["bf16"]["hid"][k], ref["fp32"]["hid"][k]),
                                                       hip_vs_bf16=_rel(hid_h[k], ref["bf16"]["hid"][k])) for k in hid_h}
    c_hf = _per_block_cos(grads_h, ref["fp32"]["grads"])
    c_bf = _per_block_cos(ref["bf16"]["grads"], ref["fp32"]["grads"])
    c_hb = _per_block_cos(grads_h, ref["bf16"]["grads"])
    rep["lora_grad_cos_per_block"] = {f"{k[0]}.{k[1]}": dict(hip_vs_fp32=c_hf[k], bf16_vs_fp32=c_bf[k], hip_vs_bf16=c_hb[k]) for k in sorted(c_hf)}
    rep["lora_grad_cos_min"] = dict(hip_vs_fp32=min(c_hf.values()), bf16_vs_fp32=min(c_bf.values()), hip_vs_bf16=min(c_hb.values()))
    gn = lambda g_: float(torch.sqrt(sum((v.double() ** 2).sum() for v in g_.values() if v is not None)))   # noqa: E731
    rep["lora_grad_global_norm"] = dict(hip=gn(grads_h), bf16=gn(ref["bf16"]["grads"]), fp32=gn(ref["fp32"]["grads"]))
    print("\n[full depth, FLUX 19+38] " + json.dumps({k: v for k, v in rep.items() if k != "lora_grad_cos_per_block"}))
    _dump("fulldepth_flux19_38", rep)
    assert len(c_hf) == Ld + Ls
    assert abs(loss_h - ref["bf16"]["loss"]) / abs(ref["bf16"]["loss"]) < 1e-2
    pr = rep["pred_rel_l2"]
    assert pr["hip_vs_fp32"] < 1.25 * pr["bf16_vs_fp32"] + 1e-3, pr
    for k, d in rep["residual_rel_l2"].items():
        assert d["hip_vs_fp32"] < 1.25 * d["bf16_vs_fp32"] + 1e-3, (k, d)
    for k in c_hf:
        assert c_hf[k] > c_bf[k] - 0.02, (k, c_hf[k], c_bf[k])
    # tight bars against the eager bf16 graph, 2x the round-4 observations (prediction 1.71e-2, residual stream 1.2-1.7e-2, cosine 0.999999)
    assert pr["hip_vs_bf16"] < 4e-2, pr
    for k, d in rep["residual_rel_l2"].items():
        assert d["hip_vs_bf16"] < 4e-2, (k, d)
    for k in c_hb:
        assert c_hb[k] > 0.9995, (k, c_hb[k])
    g_n = rep["lora_grad_global_norm"]
    assert abs(g_n["hip"] - g_n["bf16"]) / g_n["bf16"] < 2e-3, g_n
    step.zero_grad()
    from parity_util import assert_step_bit_reproducible
    n_t = assert_step_bit_reproducible(plan, lambda: step.forward_backward(emb, noise=noise, t=t), hip.lora_store.gflat, step.zero_grad, "FLUX 19 + 38 blocks")
    print(f"[full depth, FLUX 19+38] bit-reproducible over {n_t} arena tensors")
