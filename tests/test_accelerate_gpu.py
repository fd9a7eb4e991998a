"""The reference's OWN loop body under a real `accelerate.Accelerator` on the drop-in DiT.

base_trainer.py:767-773 builds `Accelerator(gradient_accumulation_steps=..., mixed_precision="no")`, :384-388 wraps the container of
LoRA layers (`AttnProcsLayers(get_lora_layers(self.dit))`) together with the optimizer in `accelerator.prepare`, :508-561 drives

    with accelerator.accumulate(self.dit):
        loss = ...; accelerator.backward(loss); accelerator.clip_grad_norm_(self.dit.parameters(), max_norm)
        optimizer.step(); lr_scheduler.step(); optimizer.zero_grad()

This test executes exactly that against the HIP module (whole DiT = one autograd node, gradients written by the kernels into the flat
buffer) and compares with the oracle trained by a plain torch loop that applies accelerate's documented semantics by hand (loss / k per
micro-step, optimizer step + clip + zero_grad only every k-th micro-step)."""
import pytest
import torch
import torch.nn as nn

from parity_util import BF, QWEN_BARS, build_pair, relmax, tiny_embeddings

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
accelerate = pytest.importorskip("accelerate")


def _get_lora_layers(model):
    """The traversal of the reference's qflux.utils.lora_utils.get_lora_layers (:25-38): every submodule whose dotted name contains
    'lora' (restated: the reference package itself does not travel to the GPU box)."""
    out = {}

    def rec(name, module):
        if "lora" in name:
            out[name] = module
        for sub, child in module.named_children():
            rec(f"{name}.{sub}", child)
    for name, module in model.named_children():
        rec(name, module)
    return out


class _AttnProcsLayers(nn.Module):
    """Stand-in for diffusers.loaders.AttnProcsLayers (third party): a ModuleList over the dict's values."""

    def __init__(self, state_dict):
        super().__init__()
        self.layers = nn.ModuleList(state_dict.values())


def test_reference_loop_under_accelerate_matches_oracle_plain_loop():
    from accelerate import Accelerator
    from common import TINY
    from oracle import qwen_dit as O
    from qflux_amd.trainer import QwenLoraTrainStep

    k, micro_steps, lr, wd, max_norm = 2, 4, 3e-3, 0.01, 1.0
    oracle, hip = build_pair(dict(TINY), device=DEV)
    accelerator = Accelerator(gradient_accumulation_steps=k, mixed_precision="no")
    assert accelerator.gradient_accumulation_steps == k
    # base_trainer.py:884-909: the optimizer over the trainable (= LoRA) parameters, built BEFORE prepare
    hparams = [p for p in hip.parameters() if p.requires_grad]
    assert hparams and all("lora" in n for n, p in hip.named_parameters() if p.requires_grad)
    optimizer = torch.optim.AdamW(hparams, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: 1.0)
    lora_layers_model = _AttnProcsLayers(_get_lora_layers(hip))
    assert len(list(lora_layers_model.parameters())) == len(hparams)
    lora_layers_model, optimizer, scheduler = accelerator.prepare(lora_layers_model, optimizer, scheduler)     # :385-387
    hip = hip.to(accelerator.device)                                                                          # :388
    assert hip.lora_store.is_consistent(DEV)          # prepare() / .to() kept the flat-buffer views
    helper = QwenLoraTrainStep(hip)      # only its compute_loss: the body of _compute_loss (qwen_image_edit_trainer.py:777-849) on dit(...)

    start = {n: p.detach().float().cpu().clone() for n, p in hip.named_parameters() if "lora" in n}
    oparams = [p for n, p in oracle.named_parameters() if "lora" in n]
    oopt = torch.optim.AdamW(oparams, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    pool = [tiny_embeddings(seed=300 + i)[0] for i in range(micro_steps)]
    g = torch.Generator().manual_seed(17)
    lo, lh, syncs = [], [], []
    oopt.zero_grad(set_to_none=True)
    for it in range(micro_steps):
        emb = pool[it]
        noise = torch.randn(emb["image_latents"].shape, generator=g)
        u = torch.rand(emb["image_latents"].shape[0], generator=g)
        # ---- the reference's loop body, verbatim in structure (base_trainer.py:518-533)
        with accelerator.accumulate(hip):
            loss = helper.compute_loss(emb, noise=noise, u=u)
            accelerator.backward(loss)
            if accelerator.sync_gradients:                                   # clip_gradients(), :449-455
                accelerator.clip_grad_norm_(hip.parameters(), max_norm)
            optimizer.step()
            scheduler.step()
            optimizer.zero_grad()
        syncs.append(bool(accelerator.sync_gradients))
        if accelerator.sync_gradients:
            avg = accelerator.gather(loss.detach()).mean()                   # :535
            assert torch.isfinite(avg)
        lh.append(loss.item())
        # ---- oracle: plain torch loop with accelerate's semantics applied by hand
        loss_o = O.qwen_compute_loss(oracle, emb, noise, u, BF)
        (loss_o / k).backward()
        if (it + 1) % k == 0:
            torch.nn.utils.clip_grad_norm_(oparams, max_norm)
            oopt.step()
            oopt.zero_grad(set_to_none=True)
        lo.append(loss_o.item())
    assert syncs == [False, True, False, True], syncs
    assert hip.lora_store.is_consistent(DEV)          # optimizer.zero_grad(set_to_none) was survived (views re-attached)
    rel = [abs(a - b) / abs(b) for a, b in zip(lh, lo)]
    got = {n: p.detach().float().cpu() for n, p in hip.named_parameters() if "lora" in n}
    # Adam's first steps move every element by ~lr * sign(g): an element whose gradient is bf16 noise around zero may go the other way in
    # the two runs (2 * lr apart after one step), so the adapters are compared as UPDATE VECTORS (cosine over all elements), not by max |d|
    ref = {n: p.detach().float().cpu() for n, p in oracle.named_parameters() if "lora" in n}
    du_h = torch.cat([(got[n] - start[n]).flatten() for n in got])
    du_o = torch.cat([(ref[n] - start[n]).flatten() for n in got])
    cos = float(torch.dot(du_h, du_o) / (du_h.norm() * du_o.norm() + 1e-30))
    drift = max(relmax(got[n], ref[n]) for n in got)
    moved = float(du_h.abs().max())
    print("accelerate loop: loss rel", rel, "update cosine after", micro_steps // k, "optimizer steps:", cos, "max rel drift", drift)
    assert max(rel) < QWEN_BARS[0] * 4, rel            # later micro-steps see adapters that already took an optimizer step
    assert cos > 0.9 and moved > 0, (cos, moved)
