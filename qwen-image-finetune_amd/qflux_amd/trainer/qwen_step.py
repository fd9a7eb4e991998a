"""The cached-embedding LoRA training step for Qwen-Image-Edit on MI355X.

Mirrors the reference's step (names and argument meaning):
  QwenImageEditTrainer._compute_loss      src/qflux/trainer/qwen_image_edit_trainer.py:777-849
  BaseTrainer.train_epoch (one iteration) src/qflux/trainer/base_trainer.py:508-561
  BaseTrainer.clip_gradients              :449-455   (global-norm clip, here over the trainable params only)
  DDP of the LoRA container               :384-393   (here: ONE explicit RCCL all-reduce of the flat LoRA gradient)

Two ways to drive it:
  * drop-in: `loss = step.compute_loss(embeddings)` returns an autograd scalar built on `self.dit(...)`;
    `loss.backward()`, any torch optimizer over the LoRA parameters.
  * fused (bench / production): `step.train_step(embeddings)` = prepare -> DiT forward program -> criterion
    kernel (loss + dpred) -> DiT backward program -> all-reduce -> fused clip+AdamW; no autograd graph, no
    host synchronisation anywhere in the step.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .. import ops
from .._lib import PRODIGY_STATE as L_PRODIGY_STATE

BF = torch.bfloat16


def flowmatch_tables(num_train_timesteps: int = 1000, shift: float = 1.0):
    """FlowMatchEulerDiscreteScheduler.timesteps / .sigmas as built at construction (third-party lookup
    tables, qwen_image_edit_trainer.py:807-810,851-861; dynamic shifting => identity shift at init)."""
    ts = torch.linspace(1, num_train_timesteps, num_train_timesteps).flip(0)
    sig = ts / num_train_timesteps
    sig = shift * sig / (1 + (shift - 1) * sig)
    return sig * num_train_timesteps, sig


def map_mask_to_latent(image_mask: torch.Tensor) -> torch.Tensor:
    """[B,H,W] pixel-space edit mask -> [B, (H/16)(W/16)] packed-latent token mask: 8x8 average pool (VAE stride), then
    the maximum over each 2x2 packing patch (src/qflux/losses/edit_mask_loss.py:7-36).  Host-side plumbing."""
    B, H, W = image_mask.shape
    lh, lw = H // 8, W // 8
    m = torch.nn.functional.avg_pool2d(image_mask.float().unsqueeze(1), kernel_size=8, stride=8).squeeze(1)
    m = m.reshape(B, lh // 2, 2, lw // 2, 2).permute(0, 1, 3, 2, 4).reshape(B, lh // 2, lw // 2, 4)
    return m.max(dim=-1)[0].reshape(B, (lh // 2) * (lw // 2))


class QwenLoraTrainStep:
    def __init__(self, dit, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=None, max_grad_norm=1.0,
                 weight_dtype=BF, process_group=None, criterion="mse", forground_weight=2.0, background_weight=1.0,
                 bucket_mb=24.0, optimizer="adamw", optimizer_args=None):
        """optimizer: "adamw" (torch.optim.AdamW semantics, lr/betas/eps/weight_decay above) or "prodigy" (prodigyopt.Prodigy, the
        reference's parameter-free choice: configs/face_seg_flux_kontext_fp16_prodigy.yaml:41-47); optimizer_args = the extra
        init_args of that class (use_bias_correction, safeguard_warmup, beta3, decouple, d0, d_coef, growth_rate).  With Prodigy
        `lr` is the schedule multiplier the reference sets to 1.0.
        criterion: "mse" = MseLoss (losses/mse_loss.py:46-83); "mask_edit" = MaskEditLoss(forground_weight,
        background_weight) (losses/edit_mask_loss.py:39-90), fed by embeddings["edit_mask"] [B,S_t] (all-ones when absent)."""
        if criterion not in ("mse", "mask_edit"):
            raise ValueError(f"unknown criterion {criterion!r}")
        if optimizer not in ("adamw", "adam", "adam8bit", "prodigy"):
            raise ValueError(f"unknown optimizer {optimizer!r}")
        # weight_decay=None = "the optimizer class's own default": 0.01 for AdamW (torch.optim.AdamW), 0 for Adam / Adam8bit / Prodigy
        # keeps its explicit value (the reference's Prodigy configs pass 0.01).  An EXPLICIT value is never reinterpreted.
        if optimizer in ("adam", "adam8bit"):
            # bitsandbytes.optim.Adam8bit -- what most of the reference's YAMLs select (configs/face_seg_config.yaml:56-59:
            # lr + betas only) -- is Adam with blockwise 8-bit quantised moments, a device to fit 24-48 GB cards.  The LoRA state
            # here is 2 x 94 MB of fp32 next to 288 GB of HBM: the moments stay fp32 (strictly closer to exact Adam than the 8-bit
            # code book; optimizer.bin then holds fp32 exp_avg / exp_avg_sq in torch.optim.Adam's layout, not bnb's state1 / state2 /
            # absmax blocks).  Adam's weight decay is the L2 form (added to the gradient), not AdamW's decoupled one: only the
            # configs' weight_decay = 0 is mapped.
            if weight_decay is not None and float(weight_decay) != 0.0:
                raise NotImplementedError(f"{optimizer} with L2 weight decay {weight_decay} (bnb adds wd * p to the gradient; the fused "
                                          "kernel implements AdamW's decoupled form only; the reference's configs use none)")
            weight_decay = 0.0
            self.optimizer_alias, optimizer = optimizer, "adamw"
        if weight_decay is None:
            weight_decay = 0.01 if optimizer == "adamw" else 0.0
        self.optimizer = optimizer
        self.optimizer_args = dict(beta3=None, decouple=True, use_bias_correction=False, safeguard_warmup=False, d0=1e-6,
                                   d_coef=1.0, growth_rate=float("inf"))
        unknown = set(optimizer_args or {}) - set(self.optimizer_args)
        if unknown or (optimizer_args and optimizer != "prodigy"):
            raise ValueError(f"unsupported optimizer_args for {optimizer}: {sorted(unknown) or sorted(optimizer_args)}")
        self.optimizer_args.update(optimizer_args or {})
        self._ps = self._p0 = self._pstate = None
        self.criterion, self.fg, self.bg = criterion, float(forground_weight), float(background_weight)
        self.dit = dit
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.weight_dtype = weight_dtype
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.timesteps_tbl, self.sigmas_tbl = flowmatch_tables()
        self.global_step = 0
        self._m = self._v = None
        self._gnorm = None
        # data-parallel exchange overlapped with the backward: the flat gradient is all-reduced in buckets of whole DiT
        # blocks as soon as their backward segment has been enqueued (the gradient of block i is final when its segment ends)
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self._pending = []
        self._reduced = False
        self._synced = False      # rank 0's adapter / optimizer state is broadcast before the first step (broadcast_state)
        self._synced_version = None
        # QFX_DP_FORCE=1: run the bucketed exchange on a ONE-rank process group too (the collectives are then identities) -- lets a
        # one-GPU box execute the real RCCL code path: communicator, RCCL's stream, async handles, ordering against the main and
        # the side gradient stream (tests/test_dp_gpu.py)
        self._force_dp = os.environ.get("QFX_DP_FORCE", "0") == "1" and dist.is_available() and dist.is_initialized()

    def _ensure_synced(self):
        """Rank 0's adapter + optimizer state reaches every rank before the first step AND again whenever the model's adapter set
        was re-injected or re-loaded since (dit._adapter_gen moves on add_adapter / load_lora_adapter / load_state_dict ONLY): a
        re-loaded adapter must not depend on every rank having produced identical weights.  Purely local actions that rebuild the
        plans (.to(), set_adapter, merge / unmerge, quantize_trunk: dit._version) do NOT trigger the collective -- a rank-0-only
        validation or merge would otherwise leave the ranks with mismatched collectives and hang the job; call
        resync() on every rank after such an action if the adapter weights themselves were changed by it."""
        ver = getattr(self.dit, "_adapter_gen", 0)
        if not self._synced or self._synced_version != ver:
            self._synced, self._synced_version = True, ver
            if self.world > 1:
                self.dit.lora_store
                self.broadcast_state()

    def resync(self):
        """Explicit collective resync (every rank must call it): rank 0's adapter + optimizer state to all ranks at the next step."""
        self._synced = False

    # ------------------------------------------------------------------ sampling (CPU RNG like the reference)
    def sample_timesteps(self, batch_size, u=None):
        if u is None:
            u = torch.rand(size=(batch_size,), device="cpu")  # compute_density_for_timestep_sampling("none")
        idx = (u * 1000).long()
        return self.timesteps_tbl[idx], self.sigmas_tbl[idx]

    def _prepare(self, embeddings, noise=None, u=None):
        dev = self.dit.device
        x0 = embeddings["image_latents"].to(self.weight_dtype).to(dev, non_blocking=True)
        ctrl = embeddings["control_latents"].to(self.weight_dtype).to(dev, non_blocking=True)
        pe = embeddings["prompt_embeds"].to(self.weight_dtype).to(dev, non_blocking=True)
        B = x0.shape[0]
        if noise is None:
            noise = torch.randn_like(x0, device=dev, dtype=self.weight_dtype)
        else:
            noise = noise.to(self.weight_dtype).to(dev)
        timesteps, sigmas = self.sample_timesteps(B, u)
        sig = sigmas.to(self.weight_dtype).to(dev, non_blocking=True)
        packed, target = ops.flowmatch_prepare(x0.contiguous(), noise.contiguous(), ctrl.contiguous(), sig)
        t_in = (timesteps / 1000).to(dev, non_blocking=True)
        return packed, target, pe, t_in, x0.shape[1]

    # ------------------------------------------------------------------ drop-in (autograd) path
    def compute_loss(self, embeddings, noise=None, u=None):
        packed, target, pe, t_in, S_t = self._prepare(embeddings, noise, u)
        mask = embeddings["prompt_embeds_mask"]
        txt_seq_lens = [pe.shape[1]] * pe.shape[0] if mask is None else mask.sum(dim=1).tolist()
        pred = self.dit(hidden_states=packed, timestep=t_in, guidance=None, encoder_hidden_states_mask=mask,
                        encoder_hidden_states=pe, img_shapes=embeddings["img_shapes"], txt_seq_lens=txt_seq_lens,
                        return_dict=False)[0]
        pred = pred[:, :S_t]
        el = (pred.float() - target.float()) ** 2            # MseLoss with weighting = 1 (mse_loss.py:71-81)
        if self.criterion == "mask_edit":                      # MaskEditLoss, reduction="mean" (edit_mask_loss.py:62-86)
            el = el * self._token_weights(embeddings, target.shape[0], S_t, el.device).unsqueeze(-1)
        return torch.mean(el.reshape(target.shape[0], -1), dim=1).mean()

    def _token_weights(self, embeddings, B, S_t, dev):
        m = embeddings.get("edit_mask")
        m = torch.ones(B, S_t) if m is None else m.float()
        return (m * self.fg + (1.0 - m) * self.bg).to(dev).contiguous()

    # ------------------------------------------------------------------ fused path
    def forward_backward(self, embeddings, noise=None, u=None, grad_scale=1.0, sync=True):
        """loss (device fp32 scalar); LoRA grads ACCUMULATE into the flat gradient buffer.
        sync=False = accelerator.accumulate()/no_sync micro-step (base_trainer.py:518): no gradient exchange is started; pass
        sync=True on the last micro-step of the window (the buckets then carry the accumulated sums)."""
        packed, target, pe, t_in, S_t = self._prepare(embeddings, noise, u)
        dit = self.dit
        plan = dit.get_plan(packed.shape[0], packed.shape[1], pe.shape[1], embeddings["img_shapes"], None)
        dit.lora_store  # make sure the flat buffers / grads are attached
        self._ensure_synced()
        pred = plan.run_forward(packed, pe, t_in)
        if self.criterion == "mask_edit":
            B = packed.shape[0]
            tw = self._token_weights(embeddings, B, S_t, pred.device)
            loss, dpred = ops.mse_token_weighted_fwd_bwd(pred, target, tw, S_t, 1.0 / (B * S_t), gscale=grad_scale)
        else:
            loss, dpred = ops.mse_loss_fwd_bwd(pred, target, S_t, gscale=grad_scale)
        self._mark_unexchanged()       # local gradients are added below: whatever exchange a drop-in backward did before is stale
        plan.run_backward(dpred, on_segment=self._bucket_hook() if ((self.world > 1 or self._force_dp) and sync) else None)
        return loss

    # ------------------------------------------------------------------ hipGraph replay of the DiT part of the step
    def capture_graph(self, embeddings):
        """Capture [LoRA operand refresh, forward program, loss, backward program] -- every launch of the step except the
        flow-match preparation (host RNG) and the 3 optimizer launches (host-side step count / lr) -- into ONE hipGraph for the
        shape of `embeddings`, side-stream gradient launches and their fork / join events included.  Returns
        step(embeddings, noise=None, u=None) -> loss, a full optimisation step that replays the graph: the ~1.45k ctypes calls
        of the Python replay become one hipGraphLaunch.  Inputs are staged through static buffers; all arena pointers are baked
        in, so the graph is valid until the plan is rebuilt (adapter set / quantisation / shape change -> capture again).
        Data-parallel: the gradient exchange runs as one all-reduce after the graph (the bucketed overlap needs the eager
        replay)."""
        if self.criterion == "mask_edit":
            raise NotImplementedError("capture_graph: the mask_edit criterion takes per-step token weights; use train_step")
        dit = self.dit
        packed, target, pe, t_in, S_t = self._prepare(embeddings)
        plan = dit.get_plan(packed.shape[0], packed.shape[1], pe.shape[1], embeddings["img_shapes"], None)
        dit.lora_store
        self._ensure_synced()      # outside the capture: a job driven only by the captured step still starts from rank 0's state
        version = dit._version
        static = [torch.empty_like(t) for t in (packed, target, pe, t_in)]
        for d, s_ in zip(static, (packed, target, pe, t_in)):
            d.copy_(s_)

        def body():
            pred = plan.run_forward(static[0], static[2], static[3])
            loss, dpred = ops.mse_loss_fwd_bwd(pred, static[1], S_t)
            plan.run_backward(dpred)
            return loss

        warm = torch.cuda.Stream(device=dit.device)     # one eager pass on a non-default stream (lazy module loads, pool warm-up)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            body()
        torch.cuda.current_stream().wait_stream(warm)
        self.zero_grad()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss_static = body()
        self.zero_grad()
        shapes = [tuple(t.shape) for t in static]

        def step(emb, noise=None, u=None):
            if dit._version != version or not any(p is plan for p in dit._plans.values()):
                raise RuntimeError("capture_graph: the plan this graph was captured on is gone (adapters / quantisation changed "
                                   "or it was evicted); capture again")
            ins = self._prepare(emb, noise, u)
            if [tuple(t.shape) for t in ins[:4]] != shapes:
                raise ValueError("capture_graph: batch shape differs from the captured one")
            for d, s_ in zip(static, ins[:4]):
                d.copy_(s_)
            self._ensure_synced()
            self._mark_unexchanged()
            graph.replay()
            self._finish_buckets = None
            self.optimizer_step(grad_scale=self.allreduce_grads())
            self.zero_grad()
            return loss_static.clone()      # the graph overwrites loss_static on every replay

        step.graph = graph
        return step

    # ------------------------------------------------------------------ bucketed all-reduce behind the backward
    def _bucket_hook(self):
        st = self.dit.lora_store
        ents = st.entries
        todo = set(range(len(ents)))
        acc = []          # entry indices final but not yet reduced
        self._pending, self._reduced = [], False

        def flush(force=False):
            nbytes = sum(((ents[i][3] + 63) // 64 * 64) * 4 for i in acc)
            if not acc or (not force and nbytes < self.bucket_bytes):
                return
            for lo, hi in _contiguous_runs(sorted(acc), ents):
                self._pending.append(dist.all_reduce(st.gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            acc.clear()

        # adapters of the conditioning head (AdaLN modulation linears img_mod.1 / txt_mod.1, FLUX norm*.linear, the embedders) live
        # under the block prefixes too, but their gradients are written by the LAST call of the backward program (the head's own
        # backward, fed by the d(modulation) column sums of every block): they are never final at a block mark and go out with
        # finish(), after the whole program
        cond_sfx = tuple(getattr(self.dit, "_COND_SUFFIXES", ()))
        late = {i for i in todo if cond_sfx and ents[i][0].split(".lora_")[0].endswith(cond_sfx)}

        def hook(prefix):
            done = [i for i in todo if i not in late and ents[i][0].startswith(prefix)]
            todo.difference_update(done)
            acc.extend(done)
            flush()

        def finish():
            acc.extend(sorted(todo))   # adapters outside the marked blocks, if any
            todo.clear()
            flush(force=True)

        self._finish_buckets = finish
        return hook

    def allreduce_grads(self):
        """Returns the factor the optimizer applies to the summed gradient (1/world)."""
        if self.world > 1 or self._force_dp:
            fin = getattr(self, "_finish_buckets", None)
            if fin is not None:
                fin()
                self._finish_buckets = None
                for w in self._pending:
                    w.wait()
                self._pending = []
            else:   # no bucketed backward of THIS object ran (drop-in autograd path, or the hipGraph replay)
                dp = getattr(self.dit, "_dp", None)
                if dp is not None and dp.exchanged:
                    # dit.enable_data_parallel(): the LAST loss.backward() since zero_grad already exchanged (overlapped) and averaged
                    # the accumulated gradient (dp.py sets the flag in finish(); merely being enabled proves nothing -- the captured
                    # graph replays plan.run_backward directly and never passes through the autograd node's exchange)
                    return 1.0
                dist.all_reduce(self.dit.lora_store.gflat, op=dist.ReduceOp.SUM, group=self.group)
            return 1.0 / self.world
        return 1.0

    def optimizer_step(self, grad_scale=1.0):
        st = self.dit.lora_store
        if self._m is None or self._m.numel() != st.pflat.numel() or self._m.device != st.pflat.device:
            self._m = torch.zeros_like(st.pflat)
            self._v = torch.zeros_like(st.pflat)
            self._gnorm = torch.zeros((), dtype=torch.float32, device=st.pflat.device)
        self.global_step += 1
        if getattr(self, "_gparts", None) is None or self._gparts.device != st.pflat.device:
            self._gparts = torch.zeros(1024, dtype=torch.float32, device=st.pflat.device)
        ops.sumsq_det(st.gflat, self._gnorm, self._gparts)     # fixed reduction order: every replica computes the same clip coefficient
        if self.optimizer == "prodigy":
            if self._pstate is None or self._ps.numel() != st.pflat.numel():
                self._ps = torch.zeros_like(st.pflat)
                self._p0 = st.pflat.detach().clone()     # parameters at the first step() call
                self._pstate = torch.zeros(L_PRODIGY_STATE, dtype=torch.float64, device=st.pflat.device)
                ops.prodigy_init_state(self._pstate, self.optimizer_args["d0"])
            ops.prodigy_step(st.pflat, st.gflat, self._m, self._v, self._ps, self._p0, self._pstate, lr=self.lr, betas=self.betas,
                             eps=self.eps, weight_decay=self.weight_decay, gnorm_sq=self._gnorm, max_norm=self.max_grad_norm,
                             grad_scale=grad_scale, **self.optimizer_args)
            return
        ops.adamw_step(st.pflat, st.gflat, self._m, self._v, self.lr, self.betas[0], self.betas[1], self.eps,
                       self.weight_decay, self.global_step, gnorm_sq=self._gnorm, max_norm=self.max_grad_norm,
                       grad_scale=grad_scale)

    def _mark_unexchanged(self):
        dp = getattr(self.dit, "_dp", None)
        if dp is not None:
            dp.exchanged = False

    def zero_grad(self):
        self.dit.lora_store.gflat.zero_()
        self._mark_unexchanged()

    # ------------------------------------------------------------------ optimizer / resume state (base_trainer.py:827-875,944-1002)
    def state_dict(self):
        """torch.optim.AdamW-style state: {"state": {i: {"step","exp_avg","exp_avg_sq"}}, "param_groups": [...]} with one entry
        per LoRA parameter in named_parameters() order (what accelerate's optimizer.bin holds for the reference)."""
        st = self.dit.lora_store
        state = {}
        if self.optimizer == "prodigy":
            return self._prodigy_state_dict()
        for i, (_, p, off, k) in enumerate(st.entries):
            if self._m is None:
                break
            state[i] = {"step": torch.tensor(float(self.global_step)), "exp_avg": self._m[off:off + k].view(p.shape).detach().cpu().clone(),
                        "exp_avg_sq": self._v[off:off + k].view(p.shape).detach().cpu().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "params": list(range(len(st.entries)))}
        return {"state": state, "param_groups": [group], "global_step": self.global_step}

    _PS_KEYS = ("d", "d_max", "d_numerator", "d_denom", "d_hat", "k")

    def _prodigy_state_dict(self):
        """prodigyopt layout: per-parameter {"step","s","p0","exp_avg","exp_avg_sq"}; the group carries d, d_max, d_numerator,
        d_denom, d_hat, k next to the init_args."""
        st = self.dit.lora_store
        state = {}
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay, **self.optimizer_args)
        d0 = self.optimizer_args["d0"]
        group.update(d=d0, d_max=d0, d_numerator=0.0, d_denom=0.0, d_hat=d0, k=0)
        if self._pstate is not None:
            vals = self._pstate.cpu().tolist()
            group.update({n: vals[i] for i, n in enumerate(self._PS_KEYS)})
            group["k"] = int(group["k"])
            for i, (_, p, off, k) in enumerate(st.entries):
                state[i] = {"step": group["k"], "s": self._ps[off:off + k].detach().cpu().clone(),
                            "p0": self._p0[off:off + k].detach().cpu().clone(),
                            "exp_avg": self._m[off:off + k].view(p.shape).detach().cpu().clone(),
                            "exp_avg_sq": self._v[off:off + k].view(p.shape).detach().cpu().clone()}
        group["params"] = list(range(len(st.entries)))
        return {"state": state, "param_groups": [group], "global_step": self.global_step}

    def _load_prodigy_state_dict(self, sd):
        st = self.dit.lora_store
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        for n in self.optimizer_args:
            if n in g:
                self.optimizer_args[n] = g[n]
        self.global_step = int(sd.get("global_step", g.get("k", 0)))
        self._gnorm = torch.zeros((), dtype=torch.float32, device=st.pflat.device)
        if not sd["state"]:
            self._pstate = None
            return
        self._m = torch.zeros_like(st.pflat); self._v = torch.zeros_like(st.pflat)
        self._ps = torch.zeros_like(st.pflat); self._p0 = torch.zeros_like(st.pflat)
        for i, (_, p, off, k) in enumerate(st.entries):
            e = sd["state"][i]
            self._m[off:off + k].copy_(e["exp_avg"].reshape(-1)); self._v[off:off + k].copy_(e["exp_avg_sq"].reshape(-1))
            self._ps[off:off + k].copy_(e["s"].reshape(-1))
            if e["p0"].numel() == k:            # the package stores a 0-dim zero for an all-zero parameter
                self._p0[off:off + k].copy_(e["p0"].reshape(-1))
        vals = [float(g[n]) for n in self._PS_KEYS] + [0.0] * (L_PRODIGY_STATE - len(self._PS_KEYS))
        self._pstate = torch.tensor(vals, dtype=torch.float64).to(st.pflat.device)

    def load_state_dict(self, sd):
        if self.optimizer == "prodigy":
            return self._load_prodigy_state_dict(sd)
        st = self.dit.lora_store
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        self._m = torch.zeros_like(st.pflat); self._v = torch.zeros_like(st.pflat)
        self._gnorm = torch.zeros((), dtype=torch.float32, device=st.pflat.device)
        step = sd.get("global_step", 0)
        for i, (_, p, off, k) in enumerate(st.entries):
            e = sd["state"].get(i)
            if e is None:
                continue
            self._m[off:off + k].copy_(e["exp_avg"].reshape(-1).to(self._m.device))
            self._v[off:off + k].copy_(e["exp_avg_sq"].reshape(-1).to(self._v.device))
            step = max(step, int(float(e["step"])))
        self.global_step = int(step)

    def save_checkpoint(self, save_dir, extra_state=None):
        """checkpoint-<e>-<step> folder of the reference (base_trainer.py:827-875): pytorch_lora_weights.safetensors (diffusers
        key style) + optimizer.bin + state.json."""
        import json
        os.makedirs(save_dir, exist_ok=True)
        self.dit.save_lora_weights(save_dir)
        torch.save(self.state_dict(), os.path.join(save_dir, "optimizer.bin"))
        with open(os.path.join(save_dir, "state.json"), "w") as f:
            json.dump(dict({"global_step": self.global_step, "lr": self.lr}, **(extra_state or {})), f, indent=2)

    def load_checkpoint(self, save_dir, adapter_name=None):
        """adapter_name: inject/overwrite that adapter from the saved weights first (None = the adapter is already in place)."""
        import json
        if adapter_name is not None:
            self.dit.load_lora_adapter(save_dir, adapter_name=adapter_name)
        self.load_state_dict(torch.load(os.path.join(save_dir, "optimizer.bin"), map_location="cpu", weights_only=False))
        self.broadcast_state()       # every replica continues from rank 0's copy of the checkpoint
        with open(os.path.join(save_dir, "state.json")) as f:
            return json.load(f)

    # ------------------------------------------------------------------ replica consistency (main.py:58, base_trainer.py:384-393)
    def _state_buffers(self):
        """Every flat buffer that must be identical on all ranks: adapter weights, then the optimizer's moment / Prodigy buffers."""
        st = self.dit.lora_store
        bufs = [("lora", st.pflat)]
        for name in ("_m", "_v", "_ps", "_p0", "_pstate"):
            t = getattr(self, name, None)
            if t is not None:
                bufs.append((name, t))
        return bufs

    def broadcast_state(self, src: int = 0):
        """Rank `src`'s adapter weights (+ optimizer state, + step count) to every rank: what DDP's constructor does for the
        reference's LoRA container (base_trainer.py:384-393; the reference otherwise relies on equal seeds, main.py:58).  Called
        once before the first step and after load_checkpoint / load_lora_adapter: a resumed or re-injected adapter set must not
        depend on every rank having read identical files.  Which optimizer buffers exist is agreed on first (rank `src` decides)."""
        if self.world <= 1:
            return
        dev = self.dit.lora_store.pflat.device
        have = torch.tensor([float(self._m is not None), float(self._pstate is not None), float(self.global_step)], device=dev)
        dist.broadcast(have, src=src, group=self.group)
        st = self.dit.lora_store
        have_m, have_p = bool(have[0].item()), bool(have[1].item())
        # the buffer list is derived from the AGREED flags on every rank (same collectives in the same order everywhere): buffers
        # rank `src` has are created where missing, buffers it lacks are dropped locally (a rank that had stepped before must not
        # carry moments the others do not have)
        if have_m:
            if self._m is None or self._m.numel() != st.pflat.numel():
                self._m, self._v = torch.zeros_like(st.pflat), torch.zeros_like(st.pflat)
                self._gnorm = torch.zeros((), dtype=torch.float32, device=dev)
        else:
            self._m = self._v = None
        if have_p:
            if self._pstate is None or self._ps is None or self._ps.numel() != st.pflat.numel():
                self._ps, self._p0 = torch.zeros_like(st.pflat), torch.zeros_like(st.pflat)
                self._pstate = torch.zeros(L_PRODIGY_STATE, dtype=torch.float64, device=dev)
        else:
            self._ps = self._p0 = self._pstate = None
        self.global_step = int(have[2].item())
        bufs = [st.pflat] + ([self._m, self._v] if have_m else []) + ([self._ps, self._p0, self._pstate] if have_p else [])
        for t in bufs:
            dist.broadcast(t, src=src, group=self.group)

    def check_replicas(self, what: str = "adapter weights"):
        """Raises if the adapter weights (and optimizer buffers) differ between ranks: two order-sensitive fp64 checksums per
        buffer, gathered and compared on every rank.  Cheap (one small all-gather); call it after loading state and periodically."""
        if self.world <= 1:
            return True
        st = self.dit.lora_store
        dev = st.pflat.device
        sums = []
        for name in ("lora", "_m", "_v", "_ps", "_p0", "_pstate"):      # fixed layout: a buffer a rank lacks is part of the verdict
            t = st.pflat if name == "lora" else getattr(self, name, None)
            if t is None:
                sums += [torch.zeros((), dtype=torch.float64, device=dev)] * 3
                continue
            d = t.detach().double().flatten()
            w = torch.arange(1, d.numel() + 1, device=d.device, dtype=torch.float64) % 8191
            sums += [torch.ones((), dtype=torch.float64, device=dev), d.sum(), (d * w).sum()]
        mine = torch.stack(sums)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        for r, o in enumerate(out):
            if not torch.equal(o, out[0]):
                raise RuntimeError(f"data-parallel replicas diverged ({what}): rank {r} differs from rank 0 "
                                   f"(checksums {o.tolist()} vs {out[0].tolist()}); call broadcast_state() after loading state")
        return True

    def train_step(self, embeddings, noise=None, u=None, micro_batches=None):
        """One full optimisation step; returns the (device) loss.  micro_batches: optional list of further embedding dicts
        accumulated before the step (gradient_accumulation_steps = 1 + len(micro_batches); mean over micro-steps)."""
        extra = list(micro_batches or [])
        k = 1 + len(extra)
        loss = self.forward_backward(embeddings, noise, u, sync=not extra)
        for j, mb in enumerate(extra):
            loss = loss + self.forward_backward(mb, sync=(j == len(extra) - 1))
        scale = self.allreduce_grads() / k
        self.optimizer_step(grad_scale=scale)
        self.zero_grad()
        return loss / k if k > 1 else loss

    def gather_loss(self, loss):
        """accelerator.gather(loss).mean() (base_trainer.py:539)."""
        if self.world > 1:
            out = [torch.zeros_like(loss) for _ in range(self.world)]
            dist.all_gather(out, loss, group=self.group)
            return torch.stack(out).mean()
        return loss


def optimizer_kwargs_from_config(class_path: str, init_args: dict | None = None) -> dict:
    """The reference's YAML `optimizer: {class_path, init_args}` (BaseTrainer.configure_optimizers, base_trainer.py:884-909) ->
    keyword arguments of QwenLoraTrainStep / FluxKontextTrainStep.
        torch.optim.AdamW                      -> optimizer="adamw"   (lr, betas, eps, weight_decay)
        bitsandbytes.optim.Adam8bit / Adam     -> optimizer="adam8bit": Adam with FP32 moments (see __init__; 8-bit states buy nothing
        bitsandbytes.optim.AdamW8bit / AdamW   -> optimizer="adamw"     next to 288 GB of HBM)
        prodigyopt.Prodigy                     -> optimizer="prodigy" + optimizer_args
    Unknown classes raise: silently training with a different optimizer is worse than stopping."""
    a = dict(init_args or {})
    name = class_path.rsplit(".", 1)[-1]
    out = {}
    for k in ("lr", "eps", "weight_decay"):
        if k in a:
            out[k] = float(a.pop(k))
    if "betas" in a:
        out["betas"] = tuple(float(b) for b in a.pop("betas"))
    if class_path in ("torch.optim.AdamW", "bitsandbytes.optim.AdamW8bit", "bitsandbytes.optim.AdamW", "bitsandbytes.optim.PagedAdamW8bit"):
        out["optimizer"] = "adamw"
    elif class_path in ("torch.optim.Adam", "bitsandbytes.optim.Adam8bit", "bitsandbytes.optim.Adam", "bitsandbytes.optim.PagedAdam8bit"):
        out["optimizer"] = "adam8bit" if "8bit" in name else "adam"
        out.setdefault("weight_decay", 0.0)
    elif class_path == "prodigyopt.Prodigy":
        out["optimizer"] = "prodigy"
        out["optimizer_args"] = {k: a.pop(k) for k in list(a) if k in ("beta3", "decouple", "use_bias_correction", "safeguard_warmup", "d0",
                                                                       "d_coef", "growth_rate")}
    else:
        raise NotImplementedError(f"optimizer {class_path!r} has no fused counterpart (use the drop-in path with the torch optimizer)")
    for k in ("min_8bit_size", "percentile_clipping", "block_wise", "optim_bits", "is_paged", "amsgrad", "foreach", "fused"):
        a.pop(k, None)       # knobs of the 8-bit state / torch dispatch: no meaning for the fused fp32 step
    if a:
        raise NotImplementedError(f"unsupported optimizer init_args for {class_path}: {sorted(a)}")
    return out


def get_scheduler(name: str, num_warmup_steps: int = 0, num_training_steps: int | None = None, num_cycles: float = 0.5):
    """lr multiplier(step) of diffusers.optimization.get_scheduler for the schedules the reference's configs use
    (base_trainer.py:900-916): constant, constant_with_warmup, linear, cosine.  Use: step.lr = base_lr * f(global_step)."""
    import math

    def warm(s):
        return float(s) / float(max(1, num_warmup_steps)) if s < num_warmup_steps else None

    if name == "constant":
        return lambda s: 1.0
    if name == "constant_with_warmup":
        return lambda s: (warm(s) if warm(s) is not None else 1.0)
    if name == "linear":
        return lambda s: (warm(s) if warm(s) is not None else
                          max(0.0, float(num_training_steps - s) / float(max(1, num_training_steps - num_warmup_steps))))
    if name == "cosine":
        def f(s):
            w = warm(s)
            if w is not None:
                return w
            prog = float(s - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * prog)))
        return f
    raise ValueError(f"unsupported lr scheduler {name!r}")


def _contiguous_runs(idx, ents):
    """Entry indices -> [lo, hi) element ranges of the flat buffer, merged where adjacent (64-element padded slots)."""
    runs = []
    for i in idx:
        lo = ents[i][2]
        hi = lo + (ents[i][3] + 63) // 64 * 64
        if runs and runs[-1][1] == lo:
            runs[-1][1] = hi
        else:
            runs.append([lo, hi])
    return [(a, b) for a, b in runs]


def init_distributed_from_env():
    """One process per GPU; backend "nccl" is RCCL on ROCm.  Returns (rank, local_rank, world).
    Test hooks: QFX_DIST_BACKEND=gloo and QFX_SHARE_GPU=1 let several ranks share device 0 (RCCL refuses duplicate devices), so
    the multi-rank code path can be exercised on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("QFX_SHARE_GPU") == "1":
        local = 0
    if torch.cuda.is_available():
        torch.cuda.set_device(local)   # before the process group: RCCL binds to the current device
    if (world > 1 or os.environ.get("QFX_BENCH_INIT_PG") == "1") and not dist.is_initialized():     # (one-rank group: tools/rccl_one_rank.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("QFX_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
