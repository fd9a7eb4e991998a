"""Validation / sampling forward: the denoising loop of the reference on the SAME launch programs as training
(inference-mode forward, LoRA applied, no gradients).

Mirrors  QwenImageEditTrainer.sampling_from_embeddings   src/qflux/trainer/qwen_image_edit_trainer.py:1116-1289
         BaseTrainer.prepare_predict_timesteps           src/qflux/trainer/base_trainer.py:1009-1043
         calculate_shift                                 src/qflux/scheduler/custom_flowmatch_scheduler.py:20-30
The scheduler itself (diffusers FlowMatchEulerDiscreteScheduler: dynamic exponential time shift, optional terminal
stretch, Euler step) is third-party and restated from its published algorithm -- parity unpinned for that part.
"""
from __future__ import annotations

import contextlib
import math

import torch

BF = torch.bfloat16


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


class FlowMatchEulerSchedule:
    """set_timesteps(sigmas=linspace(1, 1/N, N), mu) + step() of FlowMatchEulerDiscreteScheduler with
    use_dynamic_shifting=True, time_shift_type="exponential" (the Qwen-Image / FLUX scheduler_config.json)."""

    def __init__(self, num_train_timesteps=1000, base_image_seq_len=256, max_image_seq_len=4096, base_shift=0.5, max_shift=1.15,
                 shift_terminal=None):
        self.num_train_timesteps = num_train_timesteps
        self.cfg = dict(base_image_seq_len=base_image_seq_len, max_image_seq_len=max_image_seq_len, base_shift=base_shift,
                        max_shift=max_shift)
        self.shift_terminal = shift_terminal

    _KEYS = ("num_train_timesteps", "base_image_seq_len", "max_image_seq_len", "base_shift", "max_shift", "shift_terminal")

    @classmethod
    def from_config(cls, config):
        """Build the schedule from the model repository's scheduler/scheduler_config.json (a path to the json / the folder that
        holds it, or the parsed dict).  The reference reads base/max_image_seq_len, base/max_shift from `scheduler.config`
        (base_trainer.py:1027-1034) and the scheduler itself applies `shift_terminal`; the constructor defaults above are only
        the reference's `.get(..., default)` fall-backs, NOT the values of a given checkpoint -- validation sampling of a real
        model should always come through here.  Only use_dynamic_shifting=True / time_shift_type="exponential" schedulers are
        restated (what Qwen-Image and FLUX ship); anything else raises."""
        import json
        import os
        if not isinstance(config, dict):
            path = str(config)
            if os.path.isdir(path):
                for cand in ("scheduler_config.json", os.path.join("scheduler", "scheduler_config.json")):
                    if os.path.exists(os.path.join(path, cand)):
                        path = os.path.join(path, cand)
                        break
            with open(path) as f:
                config = json.load(f)
        if not config.get("use_dynamic_shifting", True):
            raise NotImplementedError("FlowMatchEulerSchedule restates the dynamic-shifting scheduler only")
        if config.get("time_shift_type", "exponential") != "exponential":
            raise NotImplementedError(f"time_shift_type {config.get('time_shift_type')!r}")
        for k in ("use_karras_sigmas", "use_exponential_sigmas", "use_beta_sigmas", "invert_sigmas", "stochastic_sampling"):
            if config.get(k):
                raise NotImplementedError(f"scheduler option {k} is not part of the restated sampling path")
        return cls(**{k: config[k] for k in cls._KEYS if k in config})

    def set_timesteps(self, num_inference_steps: int, image_seq_len: int):
        sig = torch.linspace(1.0, 1.0 / num_inference_steps, num_inference_steps, dtype=torch.float64)
        mu = calculate_shift(image_seq_len, self.cfg["base_image_seq_len"], self.cfg["max_image_seq_len"], self.cfg["base_shift"],
                             self.cfg["max_shift"])
        sig = math.exp(mu) / (math.exp(mu) + (1.0 / sig - 1.0))          # time_shift, sigma exponent 1
        if self.shift_terminal:
            one_minus = 1.0 - sig
            sig = 1.0 - one_minus / (one_minus[-1] / (1.0 - self.shift_terminal))
        sig = sig.to(torch.float32)
        self.timesteps = sig * self.num_train_timesteps
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        return self.timesteps

    @staticmethod
    def step(model_output, sigma, sigma_next, sample):
        prev = sample.to(torch.float32) + (sigma_next - sigma) * model_output.to(torch.float32)
        return prev.to(model_output.dtype)


class QwenSampler:
    def __init__(self, dit, weight_dtype=BF, schedule: FlowMatchEulerSchedule | None = None, scheduler_config=None):
        """scheduler_config: the checkpoint's scheduler_config.json (path, folder or dict) -> FlowMatchEulerSchedule.from_config.
        With neither argument the reference's fall-back constants are used (base_trainer.py:1027-1034 defaults)."""
        self.dit = dit
        self.weight_dtype = weight_dtype
        if schedule is None:
            schedule = FlowMatchEulerSchedule.from_config(scheduler_config) if scheduler_config is not None else FlowMatchEulerSchedule()
        self.schedule = schedule

    @torch.inference_mode()
    def sample(self, embeddings: dict) -> torch.Tensor:
        """embeddings: control_latents [B,S_c,64], prompt_embeds [B,T,J], prompt_embeds_mask, img_shapes, num_inference_steps,
        true_cfg_scale, latents [B,S_t,64] (initial noise, packed) and, for true CFG, negative_prompt_embeds(+_mask).
        Returns the final packed latents [B,S_t,64]."""
        dit, dev, dt = self.dit, self.dit.device, self.weight_dtype
        steps = int(embeddings["num_inference_steps"])
        cfg = float(embeddings.get("true_cfg_scale", 1.0))
        do_cfg = cfg > 1 and embeddings.get("negative_prompt_embeds") is not None
        ctrl = embeddings["control_latents"].to(dev, dtype=dt)
        pe = embeddings["prompt_embeds"].to(dev, dtype=dt)
        mask = embeddings.get("prompt_embeds_mask")
        latents = embeddings["latents"].to(dev, dtype=dt)
        txt_seq_lens = [pe.shape[1]] * pe.shape[0] if mask is None else mask.sum(dim=1).tolist()
        if do_cfg:
            npe = embeddings["negative_prompt_embeds"].to(dev, dtype=dt)
            nmask = embeddings.get("negative_prompt_embeds_mask")
            n_lens = [npe.shape[1]] * npe.shape[0] if nmask is None else nmask.sum(dim=1).tolist()
        guidance = None
        if dit.config.guidance_embeds:
            guidance = torch.full([latents.shape[0]], float(embeddings.get("guidance", 1.0)), device=dev, dtype=torch.float32)
        timesteps = self.schedule.set_timesteps(steps, latents.shape[1])
        sig = self.schedule.sigmas
        ctx = getattr(dit, "cache_context", None)
        for i, t in enumerate(timesteps):
            x = torch.cat([latents, ctrl], dim=1)
            ts = (t.expand(latents.shape[0]).to(dt) / 1000).to(dev)
            with (ctx("cond") if ctx else contextlib.nullcontext()):
                pred = dit(hidden_states=x, timestep=ts, guidance=guidance, encoder_hidden_states_mask=mask, encoder_hidden_states=pe,
                           img_shapes=embeddings["img_shapes"], txt_seq_lens=txt_seq_lens, attention_kwargs={}, return_dict=False)[0]
            pred = pred[:, : latents.size(1)]
            if do_cfg:
                with (ctx("uncond") if ctx else contextlib.nullcontext()):
                    neg = dit(hidden_states=x, timestep=ts, guidance=guidance, encoder_hidden_states_mask=nmask,
                              encoder_hidden_states=npe, img_shapes=embeddings["img_shapes"], txt_seq_lens=n_lens, attention_kwargs={},
                              return_dict=False)[0][:, : latents.size(1)]
                comb = neg + cfg * (pred - neg)
                cond_norm = torch.norm(pred, dim=-1, keepdim=True)
                noise_norm = torch.norm(comb, dim=-1, keepdim=True)
                pred = comb * (cond_norm / noise_norm)
            latents = self.schedule.step(pred, float(sig[i]), float(sig[i + 1]), latents)
        return latents


class FluxSampler:
    """FLUX-Kontext validation / sampling loop on the training launch programs (inference mode, LoRA applied).

    Mirrors FluxKontextLoraTrainer.sampling_from_embeddings (src/qflux/trainer/flux_kontext_trainer.py:902-976): guidance-
    distilled Euler flow-match loop -- `guidance` embedded every step, timestep = bf16(t) / 1000, control tokens concatenated
    behind the latents, ids = [latent ids | control ids]; optional true CFG WITHOUT the norm rescale of the Qwen loop
    (`neg + s * (pred - neg)`, :965); scheduler as QwenSampler (prepare_predict_timesteps, base_trainer.py:1009-1043)."""

    def __init__(self, dit, weight_dtype=BF, schedule: FlowMatchEulerSchedule | None = None, scheduler_config=None):
        self.dit = dit
        self.weight_dtype = weight_dtype
        if schedule is None:
            schedule = FlowMatchEulerSchedule.from_config(scheduler_config) if scheduler_config is not None else FlowMatchEulerSchedule()
        self.schedule = schedule

    @torch.inference_mode()
    def sample(self, embeddings: dict) -> torch.Tensor:
        """embeddings: latents [B,S_t,64] + latent_ids [S_t,3] (initial noise, packed), control_latents [B,S_c,64], control_ids
        [S_c,3], pooled_prompt_embeds [B,P], prompt_embeds [B,T,J], text_ids [T,3], guidance, num_inference_steps,
        true_cfg_scale and, for true CFG, negative_pooled_prompt_embeds / negative_prompt_embeds / negative_text_ids.
        Returns the final packed latents [B,S_t,64]."""
        dit, dev, dt = self.dit, self.dit.device, self.weight_dtype
        steps = int(embeddings["num_inference_steps"])
        cfg = float(embeddings.get("true_cfg_scale", 1.0))
        do_cfg = cfg > 1.0 and "negative_pooled_prompt_embeds" in embeddings
        ctrl = embeddings["control_latents"].to(dev, dtype=dt)
        latents = embeddings["latents"].to(dev, dtype=dt)
        ids = torch.cat([embeddings["latent_ids"].float().cpu(), embeddings["control_ids"].float().cpu()], dim=0)
        B, n = latents.shape[0], latents.shape[1]
        pooled = embeddings["pooled_prompt_embeds"].to(dev, dtype=dt)
        pe = embeddings["prompt_embeds"].to(dev, dtype=dt)
        txt_ids = embeddings["text_ids"].float().cpu()
        guidance = None
        if dit.config.guidance_embeds:
            guidance = torch.full([B], float(embeddings.get("guidance", 1.0)), device=dev, dtype=torch.float32)
        if do_cfg:
            npooled = embeddings["negative_pooled_prompt_embeds"].to(dev, dtype=dt)
            npe = embeddings["negative_prompt_embeds"].to(dev, dtype=dt)
            ntxt_ids = embeddings["negative_text_ids"].float().cpu()
        timesteps = self.schedule.set_timesteps(steps, n)
        sig = self.schedule.sigmas
        for i, t in enumerate(timesteps):
            x = torch.cat([latents, ctrl], dim=1)
            ts = (t.expand(B).to(dt) / 1000).to(dev)
            pred = dit(hidden_states=x, timestep=ts, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=pe,
                       txt_ids=txt_ids, img_ids=ids, joint_attention_kwargs={}, return_dict=False)[0][:, :n]
            if do_cfg:
                neg = dit(hidden_states=x, timestep=ts, guidance=guidance, pooled_projections=npooled, encoder_hidden_states=npe,
                          txt_ids=ntxt_ids, img_ids=ids, joint_attention_kwargs={}, return_dict=False)[0][:, :n]
                pred = neg + cfg * (pred - neg)
            latents = self.schedule.step(pred, float(sig[i]), float(sig[i + 1]), latents)
        return latents
