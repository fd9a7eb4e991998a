"""Cached-embedding LoRA training step for FLUX-Kontext on MI355X.

Mirrors FluxKontextLoraTrainer._compute_loss_shared_mode (src/qflux/trainer/flux_kontext_trainer.py:494-577):
noise ~ N(0,1) bf16, t ~ U(0,1) bf16 (both injectable, :515-521), x_t = (1-t) x0 + t noise, latent ids built per
step (:871-883), concat with the control latents / ids, guidance = 1 when the model has guidance embeddings,
model call, slice, target = noise - x0, MSE.  Same two ways to drive it as QwenLoraTrainStep.
"""
from __future__ import annotations

import torch

from .. import ops
from .qwen_step import BF, QwenLoraTrainStep


def prepare_latent_image_ids(height: int, width: int) -> torch.Tensor:
    ids = torch.zeros(height, width, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(height)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(width)[None, :]
    return ids.reshape(height * width, 3)


class FluxKontextTrainStep(QwenLoraTrainStep):
    def _prepare_flux(self, embeddings, noise=None, t=None):
        dev = self.dit.device
        x0 = embeddings["image_latents"].to(dev, non_blocking=True)
        if x0.dtype != torch.float16:   # the embedding cache stores fp16 (cache_manager.py:78); keep its rounding behaviour
            x0 = x0.to(torch.float16)
        ctrl = embeddings["control_latents"].to(dev, non_blocking=True).to(self.weight_dtype)
        pe = embeddings["prompt_embeds"].to(dev, non_blocking=True).to(self.weight_dtype)
        pooled = embeddings["pooled_prompt_embeds"].to(dev, non_blocking=True).to(self.weight_dtype)
        B = x0.shape[0]
        noise = torch.randn(x0.shape, device=dev, dtype=self.weight_dtype) if noise is None else noise.to(dev).to(self.weight_dtype)
        t = torch.rand((B,), device=dev, dtype=self.weight_dtype) if t is None else t.to(dev).to(self.weight_dtype)
        packed, target = ops.flowmatch_prepare(x0.contiguous(), noise.contiguous(), ctrl.contiguous(), t.contiguous(), mode=1)
        h, w = embeddings["latent_hw"]
        img_ids = torch.cat([prepare_latent_image_ids(h, w), embeddings["control_ids"].float().cpu()], dim=0)
        txt_ids = embeddings["text_ids"].float().cpu()
        guidance = torch.ones((B,), device=dev, dtype=self.weight_dtype) if self.dit.config.guidance_embeds else None
        return packed, target, pe, pooled, t, guidance, img_ids, txt_ids, x0.shape[1]

    def compute_loss(self, embeddings, noise=None, t=None):
        packed, target, pe, pooled, t, guidance, img_ids, txt_ids, S_t = self._prepare_flux(embeddings, noise, t)
        pred = self.dit(hidden_states=packed, timestep=t, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=pe,
                        txt_ids=txt_ids, img_ids=img_ids, joint_attention_kwargs={}, return_dict=False)[0]
        pred = pred[:, :S_t]
        el = (pred.float() - target.float()) ** 2
        if self.criterion == "mask_edit":      # forward_loss(..., edit_mask=embeddings["edit_mask"]) (flux_kontext_trainer.py:570-574)
            el = el * self._token_weights(embeddings, target.shape[0], S_t, el.device).unsqueeze(-1)
        return torch.mean(el.reshape(target.shape[0], -1), dim=1).mean()

    def forward_backward(self, embeddings, noise=None, t=None, grad_scale=1.0, sync=True):
        """Same contract as QwenLoraTrainStep.forward_backward (criterion "mse" | "mask_edit", sync=False = no_sync micro-step)."""
        packed, target, pe, pooled, t, guidance, img_ids, txt_ids, S_t = self._prepare_flux(embeddings, noise, t)
        dit = self.dit
        plan = dit.get_plan(packed.shape[0], packed.shape[1], pe.shape[1], img_ids, txt_ids)
        dit.lora_store
        self._ensure_synced()
        pred = plan.run_forward((packed, pooled, guidance), pe, t)
        if self.criterion == "mask_edit":
            B = packed.shape[0]
            tw = self._token_weights(embeddings, B, S_t, pred.device)
            loss, dpred = ops.mse_token_weighted_fwd_bwd(pred, target, tw, S_t, 1.0 / (B * S_t), gscale=grad_scale)
        else:
            loss, dpred = ops.mse_loss_fwd_bwd(pred, target, S_t, gscale=grad_scale)
        self._mark_unexchanged()
        plan.run_backward(dpred, on_segment=self._bucket_hook() if (self.world > 1 and sync) else None)
        return loss

    def train_step(self, embeddings, noise=None, t=None, micro_batches=None):
        """One optimisation step; micro_batches = further embedding dicts of the gradient-accumulation window (as the base class)."""
        extra = list(micro_batches or [])
        k = 1 + len(extra)
        loss = self.forward_backward(embeddings, noise, t, sync=not extra)
        for j, mb in enumerate(extra):
            loss = loss + self.forward_backward(mb, sync=(j == len(extra) - 1))
        scale = self.allreduce_grads() / k
        self.optimizer_step(grad_scale=scale)
        self.zero_grad()
        return loss / k if k > 1 else loss


def _build_multires_batch(step, samples, txt):
    """Host/device plumbing of _compute_loss_multi_resolution_mode (flux_kontext_trainer.py:579-760): per-sample ids,
    x_t, right-padding to the batch maximum, masks.  samples[i]: image_latents [n_t,64], control_latents [n_c,64],
    hw, control_hw [(h,w),...], optional noise / t."""
    dev, dt = step.dit.device, step.weight_dtype
    B = len(samples)
    seqs, ids, n_t = [], [], []
    noises, ts = [], []
    for smp in samples:
        x0 = smp["image_latents"].to(dev)
        ctrl = smp["control_latents"].to(dev)
        noise = smp["noise"].to(dev).to(dt) if "noise" in smp else torch.randn(x0.shape, device=dev, dtype=dt)
        t = smp["t"].to(dev).to(dt).reshape(1) if "t" in smp else torch.rand((1,), device=dev, dtype=dt)
        t_ = t.unsqueeze(1)
        x_t = (1.0 - t_) * x0 + t_ * noise                      # bf16 * cache dtype promotes like the reference
        seqs.append(torch.cat([x_t.to(dt), ctrl.to(dt)], dim=0))
        h, w = smp["hw"]
        parts = [prepare_latent_image_ids(h, w)]
        for j, (ch, cw) in enumerate(smp["control_hw"]):
            ci = prepare_latent_image_ids(ch, cw)
            ci[..., 0] = j + 1
            parts.append(ci)
        ids.append(torch.cat(parts, dim=0))
        assert ids[-1].shape[0] == seqs[-1].shape[0]
        n_t.append(x0.shape[0]); noises.append(noise); ts.append(t)
    S_max, n_t_max = max(s.shape[0] for s in seqs), max(n_t)
    T = txt["text_ids"].shape[0]
    inp = torch.zeros(B, S_max, 64, device=dev, dtype=dt)
    idb = torch.zeros(B, S_max, 3)
    full = torch.ones(B, T + S_max, dtype=torch.bool)
    tok_w = torch.zeros(B, n_t_max)
    target = torch.zeros(B, n_t_max, 64, device=dev, dtype=dt)
    for i in range(B):
        L = seqs[i].shape[0]
        inp[i, :L] = seqs[i]
        idb[i, :L] = ids[i]
        full[i, T + L:] = False
        tok_w[i, : n_t[i]] = 1.0
        target[i, : n_t[i]] = noises[i] - samples[i]["image_latents"].to(dev).to(dt)
    timestep = torch.cat(ts)
    guidance = torch.ones((B,), device=dev, dtype=dt) if step.dit.config.guidance_embeds else None
    pe = txt["prompt_embeds"].to(dev).to(dt)
    pooled = txt["pooled_prompt_embeds"].to(dev).to(dt)
    return dict(inp=inp, ids=idb, mask=full, tok_w=tok_w.to(dev), target=target, timestep=timestep, guidance=guidance, pe=pe,
                pooled=pooled, txt_ids=txt["text_ids"].float().cpu(), n_t_max=n_t_max, n_valid=float(sum(n_t)))


def _compute_loss_multires(self, samples, txt):
    """Autograd path: AttentionMaskMseLoss(reduction='mean') on the masked prediction (attention_mask_loss.py:146-226)."""
    b = _build_multires_batch(self, samples, txt)
    pred = self.dit(hidden_states=b["inp"], timestep=b["timestep"], guidance=b["guidance"], pooled_projections=b["pooled"],
                    encoder_hidden_states=b["pe"], txt_ids=b["txt_ids"], img_ids=b["ids"], attention_mask=b["mask"],
                    joint_attention_kwargs={}, return_dict=False)[0][:, : b["n_t_max"]]
    el = (pred.float() - b["target"].float()) ** 2
    tok = (el * b["tok_w"].unsqueeze(-1)).mean(dim=2)
    return tok.sum() / (b["n_valid"] + 1e-12)


def _forward_backward_multires(self, samples, txt, grad_scale=1.0, sync=True):
    b = _build_multires_batch(self, samples, txt)
    dit = self.dit
    B, S_i, T = b["inp"].shape[0], b["inp"].shape[1], b["pe"].shape[1]
    valid = b["mask"][:, T:].sum(dim=1).tolist()
    plan = dit.get_plan_multires(B, S_i, T, b["ids"], valid)
    dit.lora_store
    self._ensure_synced()
    pred = plan.run_forward((b["inp"], b["pooled"], b["guidance"]), b["pe"], b["timestep"])
    loss, dpred = ops.mse_token_weighted_fwd_bwd(pred, b["target"], b["tok_w"].contiguous(), b["n_t_max"], 1.0 / (b["n_valid"] + 1e-12),
                                                 gscale=grad_scale)
    self._mark_unexchanged()
    plan.run_backward(dpred, on_segment=self._bucket_hook() if (self.world > 1 and sync) else None)
    return loss


FluxKontextTrainStep.compute_loss_multires = _compute_loss_multires
FluxKontextTrainStep.forward_backward_multires = _forward_backward_multires
