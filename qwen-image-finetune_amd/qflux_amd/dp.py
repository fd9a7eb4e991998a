"""Data-parallel gradient exchange for the DROP-IN path (the reference's own trainer loop around `dit(...)` / `loss.backward()`).

The reference wraps its container of LoRA layers in DDP (accelerator.prepare, src/qflux/trainer/base_trainer.py:384-393) and DDP
reduces a gradient when autograd ACCUMULATES it into the parameter.  Here the whole DiT is one autograd node whose kernels write
dA / dB straight into the flat gradient buffer -- autograd never touches the LoRA parameters, DDP's hooks never fire.  The model
therefore exchanges its gradients itself:

    dit.enable_data_parallel(process_group=None, bucket_mb=24)      # once, after add_adapter (no-op without torch.distributed)
    ...
    loss.backward()          # bucketed all-reduce (SUM) of the flat LoRA gradient behind the backward program, drained and
                             # averaged (DDP semantics) before backward() returns
    with dit.no_sync(): ...  # gradient-accumulation micro-steps (accelerate's `accumulate(self.dit)` finds this method by name)

The fused step (QwenLoraTrainStep.train_step) has its own copy of the same exchange with the 1/world factor folded into the
optimizer kernel.
"""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist


def contiguous_runs(idx, ents):
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


class LoraGradSync:
    def __init__(self, dit, process_group=None, bucket_mb: float = 24.0):
        self.dit, self.group = dit, process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.enabled = True
        self.exchanged = False     # set by finish(): the flat gradient currently holds the exchanged + averaged sum (cleared by the
                                   # trainer step's zero_grad / by the next local-only backward)
        self._pending, self._finish = [], None

    def hook(self):
        """on_segment callback of plan.run_backward: starts an async all-reduce on the contiguous slice of the blocks whose
        gradients are final once >= bucket_bytes have accumulated.  Conditioning-head adapters (their gradients are written by the
        LAST entries of the backward program) and anything outside the marked blocks go out with finish()."""
        st = self.dit.lora_store
        ents = st.entries
        todo = set(range(len(ents)))
        acc = []
        self._pending = []
        cond_sfx = tuple(getattr(self.dit, "_COND_SUFFIXES", ()))
        late = {i for i in todo if cond_sfx and ents[i][0].split(".lora_")[0].endswith(cond_sfx)}

        def flush(force=False):
            nbytes = sum(((ents[i][3] + 63) // 64 * 64) * 4 for i in acc)
            if not acc or (not force and nbytes < self.bucket_bytes):
                return
            for lo, hi in contiguous_runs(sorted(acc), ents):
                self._pending.append(dist.all_reduce(st.gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            acc.clear()

        def on_segment(prefix):
            done = [i for i in todo if i not in late and ents[i][0].startswith(prefix)]
            todo.difference_update(done)
            acc.extend(done)
            flush()

        def finish():
            acc.extend(sorted(todo))
            todo.clear()
            flush(force=True)

        self._finish = finish
        return on_segment

    def finish(self, average: bool = True):
        """Drain the buckets; average=True divides by the world size (what DDP hands to the optimizer)."""
        if self._finish is not None:
            self._finish()
            self._finish = None
        for w in self._pending:
            w.wait()
        self._pending = []
        if average and self.world > 1:
            self.dit.lora_store.gflat.mul_(1.0 / self.world)
        self.exchanged = True


class DataParallelMixin:
    """enable_data_parallel / no_sync for the drop-in DiT modules (see module docstring)."""

    _dp = None

    def enable_data_parallel(self, process_group=None, bucket_mb: float = 24.0):
        self._dp = LoraGradSync(self, process_group, bucket_mb) if (dist.is_available() and dist.is_initialized()
                                                                    and dist.get_world_size(process_group) > 1) else None
        return self

    @contextlib.contextmanager
    def no_sync(self):
        """DDP.no_sync look-alike: backward passes inside the context only accumulate locally."""
        dp = self._dp
        old = dp.enabled if dp is not None else None
        if dp is not None:
            dp.enabled = False
        try:
            yield
        finally:
            if dp is not None:
                dp.enabled = old

    def _dp_backward(self, plan, grad_out):
        """The autograd node's backward: launch program + (when enabled) the overlapped exchange."""
        dp = self._dp
        if dp is None or not dp.enabled:
            if dp is not None:
                dp.exchanged = False       # local-only gradients were just added on top of whatever the buffer held
            plan.run_backward(grad_out)
            return
        plan.run_backward(grad_out, on_segment=dp.hook())
        dp.finish(average=True)
