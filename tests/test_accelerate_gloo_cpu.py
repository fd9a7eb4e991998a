"""CPU, world_size 2 over gloo, through a REAL `accelerate.Accelerator` (VERDICT r5 #4 ii): the reference's loop
(base_trainer.py:508-561) around a toy model that behaves like the drop-in DiT -- ONE autograd node whose backward writes the adapter
gradients straight into the flat buffer (autograd never accumulates into the LoRA parameters) and exchanges them itself
(qflux_amd/dp.py).  What is executed, not asserted in prose:

  * `accelerator.prepare(container_of_lora_layers, optimizer)` wraps the container in DistributedDataParallel (base_trainer.py:384-388):
    the wrapper must be INERT (its hooks never fire, it must not double-reduce) and must leave the flat-buffer views intact;
  * `accelerator.accumulate(dit)` finds `dit.no_sync` BY NAME on the unwrapped model and enters it on non-sync micro-steps: those only
    accumulate locally, the sync micro-step all-reduces the accumulated sum and averages it (DDP semantics);
  * `accelerator.backward` (loss / k), `accelerator.clip_grad_norm_(dit.parameters(), ...)`, the AcceleratedOptimizer's skipped steps,
    `optimizer.zero_grad()` (set_to_none), `accelerator.gather(loss)`.

Expected values are computed by every rank for BOTH ranks with plain tensor arithmetic (the toy gradient is a closed form of the rank, the
micro-step and the upstream gradient), then AdamW + clip by hand."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytest.importorskip("accelerate")


def _grad_of(rank, it, n_entries):
    """Closed-form local gradient of entry e at micro-step `it` on `rank` for an upstream gradient of 1."""
    return [float((rank + 1) * (it + 1) * (e + 2)) * 1e-2 for e in range(n_entries)]


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      ACCELERATE_USE_CPU="true")
    from accelerate import Accelerator
    from accelerate.utils import DistributedType

    from qflux_amd.dp import DataParallelMixin, LoraGradSync
    from qflux_amd.modules import LoraStore, QfxLinear, QfxLoraLinear

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.to_q = QfxLoraLinear(QfxLinear(8, 8), 4, 8, "ad")

    class _Plan:
        """What _QwenPlan offers the autograd node: run_backward(grad_out, on_segment) = the launch program, block marks last-first."""

        def __init__(self, model):
            self.model, self.it, self.calls = model, 0, []

        def run_backward(self, grad_out, on_segment=None):
            st = self.model.lora_store
            up = float(grad_out.sum())
            vals = _grad_of(rank, self.it, len(st.entries))
            nb = len(self.model.transformer_blocks)
            for i in range(nb - 1, -1, -1):
                for e, (n, p, off, k) in enumerate(st.entries):
                    if n.startswith(f"transformer_blocks.{i}."):
                        st.gflat[off:off + k] += up * vals[e]          # the kernels ACCUMULATE into the flat buffer
                if on_segment is not None:
                    on_segment(f"transformer_blocks.{i}.")
            self.calls.append(on_segment is not None)

    class _Fn(torch.autograd.Function):
        """Same shape as _QwenDiTFn (models/transformer_qwenimage.py): LoRA parameters are inputs only so that autograd schedules the
        node; backward returns None for all of them."""

        @staticmethod
        def forward(ctx, model, plan, x, *lora_params):
            ctx.plan = plan
            return x.sum().reshape(1) * 0 + 1.0

        @staticmethod
        def backward(ctx, grad_out):
            ctx.plan.model._dp_backward(ctx.plan, grad_out.contiguous())
            return (None,) * len(ctx.needs_input_grad)

    class Toy(nn.Module, DataParallelMixin):
        def __init__(self):
            super().__init__()
            self.transformer_blocks = nn.ModuleList([Blk() for _ in range(3)])
            self._store = LoraStore(self)
            self._store.rebuild("cpu")
            self.plan = _Plan(self)

        device = torch.device("cpu")

        @property
        def lora_store(self):      # = QwenImageTransformer2DModel._ensure_lora_store: re-pack when something re-seated the parameters
            if not self._store.is_consistent(self.device):
                self._store.rebuild(self.device)
                self.rebuilds += 1
            self._store.ensure_grads()
            return self._store

        rebuilds = 0

        def forward(self, x):
            return _Fn.apply(self, self.plan, x, *[p for _, p in self.lora_store.params()])

    k, micro_steps, lr, wd, max_norm = 2, 4, 1e-2, 0.01, 0.5
    accelerator = Accelerator(gradient_accumulation_steps=k, mixed_precision="no", cpu=True)
    ok = accelerator.distributed_type == DistributedType.MULTI_CPU and accelerator.num_processes == world and accelerator.use_distributed
    torch.manual_seed(0)
    toy = Toy()
    with torch.no_grad():      # ranks start from DIFFERENT adapter values: DDP's constructor broadcast (rank 0 wins) must land in the flat buffer
        toy.lora_store.pflat.copy_(torch.randn(toy.lora_store.pflat.shape, generator=torch.Generator().manual_seed(50 + rank)))
    toy.enable_data_parallel(bucket_mb=1e-4)            # what add_adapter does by itself under an initialised process group
    ok = ok and isinstance(toy._dp, LoraGradSync) and toy._dp.world == world
    params = [p for p in toy.parameters() if p.requires_grad]
    optimizer = torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    lora_layers = nn.ModuleList([m for n, m in toy.named_modules() if n.endswith(("lora_A", "lora_B"))])
    wrapped, optimizer = accelerator.prepare(lora_layers, optimizer)                 # base_trainer.py:385-387
    ok = ok and isinstance(wrapped, torch.nn.parallel.DistributedDataParallel)
    toy = toy.to(accelerator.device)                                                 # :388
    # prepare() moved the container with module.to(accelerator.device): on this backend (`cpu:0` != `cpu`) that RE-SEATS every parameter
    # (as it does in the reference's flow, where the adapters sit on the CPU until prepare) -- the store must notice and re-pack
    st = toy.lora_store
    cons_after_prepare = st.is_consistent("cpu")
    ok = ok and cons_after_prepare
    p0 = torch.randn(st.pflat.shape, generator=torch.Generator().manual_seed(50))
    live = torch.zeros_like(st.pflat, dtype=torch.bool)
    for _, _, off, kk in st.entries:
        live[off:off + kk] = True
    ok = ok and torch.equal(st.pflat[live], p0[live])                             # rank 0's values everywhere, in place

    # ---- expectation: both ranks' closed-form gradients, accelerate semantics by hand, AdamW + clip in fp32
    ne = len(st.entries)
    p_ref = p0.clone()
    m_ref, v_ref = torch.zeros_like(p_ref), torch.zeros_like(p_ref)
    acc = [torch.zeros_like(p_ref) for _ in range(world)]
    t = 0
    syncs, calls_expected, gathered = [], [], []
    for it in range(micro_steps):
        x = torch.ones(2, 3)
        with accelerator.accumulate(toy):                                             # finds toy.no_sync by name
            loss = toy(x).sum()
            accelerator.backward(loss)
            if accelerator.sync_gradients:
                accelerator.clip_grad_norm_(toy.parameters(), max_norm)
            optimizer.step()
            optimizer.zero_grad()
        toy.plan.it += 1
        syncs.append(bool(accelerator.sync_gradients))
        calls_expected.append(bool(accelerator.sync_gradients))
        if accelerator.sync_gradients:
            gathered.append(float(accelerator.gather(loss.detach().reshape(1) * (rank + 1)).mean()))
        for r in range(world):
            vals = _grad_of(r, it, ne)
            for e, (_, _, off, kk) in enumerate(st.entries):
                acc[r][off:off + kk] += vals[e] / k
        if (it + 1) % k == 0:
            g = sum(acc) / world
            gn = float(g[live].norm())
            g = g * min(1.0, max_norm / (gn + 1e-6))
            t += 1
            p_ref[live] = p_ref[live] * (1 - lr * wd)
            m_ref = 0.9 * m_ref + 0.1 * g
            v_ref = 0.999 * v_ref + 0.001 * g * g
            p_ref[live] = p_ref[live] - lr * (m_ref[live] / (1 - 0.9 ** t)) / ((v_ref[live] / (1 - 0.999 ** t)).sqrt() + 1e-8)
            acc = [torch.zeros_like(p_ref) for _ in range(world)]
    dbg = dict(pre=bool(ok), repacked=cons_after_prepare, dt=str(accelerator.distributed_type), wrapped=type(wrapped).__name__, syncs=syncs, calls=toy.plan.calls, gathered=gathered)
    ok = ok and syncs == [False, True, False, True]
    ok = ok and toy.plan.calls == calls_expected               # non-sync micro-steps ran WITHOUT the exchange hook (no_sync was entered)
    ok = ok and all(abs(gv - (world + 1) / 2) < 1e-6 for gv in gathered)
    err = float((st.pflat[live] - p_ref[live]).abs().max())
    ok = ok and err < 2e-6 and st.is_consistent("cpu") and toy.rebuilds <= 1
    # replicas identical: all-gather the flat parameters
    allp = accelerator.gather(st.pflat.reshape(1, -1))
    ok = ok and torch.equal(allp[0], allp[1])
    q.put((rank, bool(ok), err, dbg))
    accelerator.wait_for_everyone()


def test_reference_loop_under_accelerate_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 39500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert [r[:2] for r in res] == [(0, True), (1, True)], res
