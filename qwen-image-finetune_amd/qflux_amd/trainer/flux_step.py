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
        return torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="mean")

    def forward_backward(self, embeddings, noise=None, t=None, grad_scale=1.0):
        packed, target, pe, pooled, t, guidance, img_ids, txt_ids, S_t = self._prepare_flux(embeddings, noise, t)
        dit = self.dit
        plan = dit.get_plan(packed.shape[0], packed.shape[1], pe.shape[1], img_ids, txt_ids)
        dit.lora_store
        pred = plan.run_forward((packed, pooled, guidance), pe, t)
        loss, dpred = ops.mse_loss_fwd_bwd(pred, target, S_t, gscale=grad_scale)
        plan.run_backward(dpred)
        return loss

    def train_step(self, embeddings, noise=None, t=None):
        loss = self.forward_backward(embeddings, noise, t)
        scale = self.allreduce_grads()
        self.optimizer_step(grad_scale=scale)
        self.zero_grad()
        return loss
