"""CPU, world_size 2 over gloo: the data-parallel exchange is ONE all-reduce of the flat LoRA gradient buffer,
averaged inside the optimizer's grad_scale; replicas stay identical."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qflux_amd.modules import LoraStore, QfxLinear, QfxLoraLinear
    from qflux_amd.trainer import QwenLoraTrainStep

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = QfxLoraLinear(QfxLinear(8, 6), 4, 8, "ad")
            self.b = QfxLoraLinear(QfxLinear(6, 10), 4, 8, "ad")
            self._store = LoraStore(self)
            self._store.rebuild("cpu")

        @property
        def lora_store(self):
            return self._store

        device = torch.device("cpu")

    torch.manual_seed(0)
    toy = Toy()
    st = toy.lora_store
    assert st.is_consistent("cpu") and st.pflat.numel() >= sum(p.numel() for _, p in st.params())
    # parameters are views of the flat buffer, grads of the flat gradient buffer
    with torch.no_grad():
        toy.a.A.fill_(1.0)
    assert st.pflat[: toy.a.A.numel()].eq(1.0).all()
    g_local = torch.full_like(st.gflat, float(rank + 1))
    st.gflat.copy_(g_local)
    step = QwenLoraTrainStep(toy)
    assert step.world == world
    scale = step.allreduce_grads()
    expect = sum(range(1, world + 1))
    ok = bool(st.gflat.eq(expect).all()) and abs(scale - 1.0 / world) < 1e-12
    # grads seen through the Parameters are the reduced ones
    ok = ok and bool(toy.b.B.grad.eq(expect).all())
    # loss gather = mean over ranks
    lg = step.gather_loss(torch.tensor(float(rank)))
    ok = ok and abs(lg.item() - (world - 1) / 2) < 1e-6
    # zero_grad(set_to_none) survives
    for _, p in st.params():
        p.grad = None
    st.ensure_grads()
    ok = ok and all(p.grad is not None and p.grad.data_ptr() == st.gflat.data_ptr() + 4 * off for _, p, off, _ in st.entries)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_flat_lora_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)], res


def _worker_buckets(rank, world, port, q):
    """Bucketed exchange: segments of a (fake) backward program finalise the gradients of whole blocks; the hook
    all-reduces contiguous ranges of the flat buffer asynchronously; allreduce_grads() drains them."""
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qflux_amd.modules import LoraStore, QfxLinear, QfxLoraLinear
    from qflux_amd.trainer import QwenLoraTrainStep

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.to_q = QfxLoraLinear(QfxLinear(8, 8), 4, 8, "ad")
            self.to_k = QfxLoraLinear(QfxLinear(8, 8), 4, 8, "ad")

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer_blocks = nn.ModuleList([Blk() for _ in range(5)])
            self.extra = QfxLoraLinear(QfxLinear(8, 8), 4, 8, "ad")   # an adapter outside the marked blocks
            self._store = LoraStore(self)
            self._store.rebuild("cpu")

        @property
        def lora_store(self):
            return self._store

        device = torch.device("cpu")

    toy = Toy()
    st = toy.lora_store
    # tiny bucket size: every block flushes on its own -> several async all-reduces in flight
    step = QwenLoraTrainStep(toy, bucket_mb=1e-4)
    hook = step._bucket_hook()
    calls = []
    for i in range(4, -1, -1):   # backward order: last block first
        for n, p in toy.named_parameters():
            if n.startswith(f"transformer_blocks.{i}.") and "lora_" in n:
                p.grad.fill_(float((rank + 1) * (i + 1)))
        hook(f"transformer_blocks.{i}.")
        calls.append(len(step._pending))
    for n, p in toy.named_parameters():
        if n.startswith("extra.") and "lora_" in n:
            p.grad.fill_(float(10 * (rank + 1)))
    scale = step.allreduce_grads()
    tot = sum(range(1, world + 1))
    ok = abs(scale - 1.0 / world) < 1e-12 and calls == sorted(calls) and calls[-1] >= 5 and not step._pending
    for n, p in toy.named_parameters():
        if "lora_" not in n:
            continue
        if n.startswith("extra."):
            ok = ok and bool(p.grad.eq(10.0 * tot).all())
        else:
            i = int(n.split(".")[1])
            ok = ok and bool(p.grad.eq(float(tot * (i + 1))).all())
    # a second step without a bucketed backward falls back to the single all-reduce
    st.gflat.fill_(float(rank + 1))
    step.allreduce_grads()
    ok = ok and bool(st.gflat.eq(float(tot)).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bucketed_allreduce_behind_backward_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_buckets, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)], res


def _toy_model(n_blocks=3, with_cond=True):
    from qflux_amd.modules import LoraStore, QfxLinear, QfxLoraLinear

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.to_q = QfxLoraLinear(QfxLinear(8, 8), 4, 8, "ad")
            if with_cond:      # AdaLN modulation linear of the block: name transformer_blocks.<i>.img_mod.1 (index 0 is the SiLU)
                self.img_mod = nn.Sequential(nn.Identity(), QfxLoraLinear(QfxLinear(8, 48), 4, 8, "ad"))

    class Toy(nn.Module):
        _COND_SUFFIXES = ("timestep_embedder.linear_1", "img_mod.1", "txt_mod.1", "norm_out.linear")

        def __init__(self):
            super().__init__()
            self.transformer_blocks = nn.ModuleList([Blk() for _ in range(n_blocks)])
            self._store = LoraStore(self)
            self._store.rebuild("cpu")

        @property
        def lora_store(self):
            return self._store

        device = torch.device("cpu")

    return Toy()


def _worker_cond_late(rank, world, port, q):
    """with adapters on the conditioning head (all-linear, FLUX norm*.linear regex) the gradients of
    transformer_blocks.<i>.img_mod.1 are written by the LAST call of the backward program, long after block i's mark; the bucket
    hook must not reduce them at the mark.  Emulated backward: block marks first (tiny buckets: every mark flushes), the
    conditioning-head gradients only afterwards; every adapter must end up with the cross-rank sum."""
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qflux_amd.trainer import QwenLoraTrainStep
    toy = _toy_model()
    step = QwenLoraTrainStep(toy, bucket_mb=1e-4)
    hook = step._bucket_hook()
    for i in range(2, -1, -1):
        for n, p in toy.named_parameters():
            if n.startswith(f"transformer_blocks.{i}.to_q.") and "lora_" in n:
                p.grad.fill_(float((rank + 1) * (i + 1)))
        hook(f"transformer_blocks.{i}.")
    for w in step._pending:      # let every early bucket complete BEFORE the head's gradients are written (worst case for the bug:
        w.wait()                 # the slice has already been reduced when the local gradient lands in it)
    for n, p in toy.named_parameters():
        if ".img_mod.1." in n and "lora_" in n:
            i = int(n.split(".")[1])
            p.grad.add_(float((rank + 1) * 100 * (i + 1)))       # cond_head.backward accumulates into the flat buffer
    step.allreduce_grads()
    tot = sum(range(1, world + 1))
    ok = True
    for n, p in toy.named_parameters():
        if "lora_" not in n:
            continue
        i = int(n.split(".")[1])
        want = float(tot * (i + 1)) * (100.0 if ".img_mod.1." in n else 1.0)
        ok = ok and bool(p.grad.eq(want).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def _spawn(fn, world, base_port, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = base_port + (os.getpid() % 2000)
    procs = [ctx.Process(target=fn, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    return sorted(res)


def test_conditioning_head_adapters_are_reduced_after_the_backward_world2():
    assert _spawn(_worker_cond_late, 2, 33500) == [(0, True), (1, True)]


def _worker_broadcast(rank, world, port, q):
    """Rank-0 broadcast of adapter + optimizer state: ranks start from DIFFERENT adapter values and optimizer
    buffers (a resumed run where only rank 0 read the checkpoint); check_replicas() must flag that, broadcast_state() must repair
    it -- including optimizer buffers that exist on rank 0 only."""
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qflux_amd.trainer import QwenLoraTrainStep
    toy = _toy_model(with_cond=False)
    st = toy.lora_store
    with torch.no_grad():
        st.pflat.copy_(torch.randn(st.pflat.shape, generator=torch.Generator().manual_seed(100 + rank)))
    step = QwenLoraTrainStep(toy)
    if rank == 0:      # only rank 0 carries optimizer state (it loaded optimizer.bin)
        step._m = torch.full_like(st.pflat, 0.25)
        step._v = torch.full_like(st.pflat, 0.5)
        step.global_step = 17
    flagged = False
    try:
        step.check_replicas()
    except RuntimeError:
        flagged = True
    step.broadcast_state()
    ok = flagged and step.check_replicas()
    ref = torch.randn(st.pflat.shape, generator=torch.Generator().manual_seed(100))
    ok = ok and torch.equal(st.pflat, ref) and step.global_step == 17
    ok = ok and step._m is not None and bool(step._m.eq(0.25).all()) and bool(step._v.eq(0.5).all())
    # parameters are still views of the flat buffer (the broadcast was in place)
    ok = ok and st.is_consistent("cpu")
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_rank0_broadcast_of_adapter_and_optimizer_state_world3():
    assert _spawn(_worker_broadcast, 3, 35500) == [(0, True), (1, True), (2, True)]


def _worker_ragged(rank, world, port, q, root):
    """cfg #5 on 4 ranks: every rank draws DIFFERENT bucket shapes per step (a continuum of padded lengths) from a dataset whose
    size is not a multiple of world x batch -- the loader must hand every rank the same number of batches, the step's only
    collectives (bucketed gradient exchange, loss gather) must not depend on a rank's shape, and the reduced gradient must be
    identical on all ranks after each of 3 steps."""
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qflux_amd.data import CachedEmbeddingDataset, PrefetchLoader
    from qflux_amd.trainer import QwenLoraTrainStep
    ds = CachedEmbeddingDataset(root)
    loader = PrefetchLoader(ds, batch_size=2, device="cpu", rank=rank, world=world, shuffle=True, seed=7, workers=2, drop_last=False)
    n_batches = len(loader)
    toy = _toy_model(n_blocks=4)
    st = toy.lora_store
    step = QwenLoraTrainStep(toy, bucket_mb=1e-4)
    sums = []
    steps = 0
    for batch in loader:
        if steps == 3:
            break
        # a "backward" whose per-rank work depends on the batch's padded shape (ragged buckets): the gradient VALUES differ per
        # rank and per shape, the collective sequence must not
        S_pad = int(batch["image_latents"].shape[1]) + int(batch["control_latents"].shape[1])
        hook = step._bucket_hook()
        for i in range(3, -1, -1):
            for n, p in toy.named_parameters():
                if n.startswith(f"transformer_blocks.{i}.to_q.") and "lora_" in n:
                    p.grad.fill_(float(S_pad * (i + 1) + rank))
            hook(f"transformer_blocks.{i}.")
        for n, p in toy.named_parameters():
            if ".img_mod.1." in n and "lora_" in n:
                p.grad.fill_(float(S_pad + 1000 * rank))
        scale = step.allreduce_grads()
        loss = step.gather_loss(torch.tensor(float(S_pad)))
        sums.append((float(st.gflat.double().sum()), float(loss), scale))
        step.zero_grad()
        steps += 1
    q.put((rank, n_batches, steps, sums))
    dist.destroy_process_group()


def test_world4_ragged_buckets_equal_step_counts_identical_reduced_gradients(tmp_path):
    from qflux_amd.data import write_cache_sample
    g = torch.Generator().manual_seed(5)
    # 27 samples (not a multiple of 4 ranks x 2): three bucket areas, square and non-square token grids (SURVEY 8d cfg #5)
    grids = [(20, 20), (32, 32), (40, 40), (40, 26), (26, 40), (24, 16)]
    for i in range(27):
        h, w = grids[i % len(grids)]
        n = h * w // 16            # scaled-down token counts, same raggedness
        tensors = dict(image_latents=torch.randn(n, 64, generator=g), control_latents=torch.randn(n, 64, generator=g),
                       prompt_embeds=torch.randn(5, 32, generator=g), prompt_embeds_mask=torch.ones(5, dtype=torch.int64))
        write_cache_sample(tmp_path, f"{i:04x}main", tensors, img_shapes=[(3, h * 16, w * 16)] * 2,
                           hashes={k: f"{i:04x}{k[:3]}" for k in tensors})
    res = _spawn(_worker_ragged, 4, 37500, extra=(str(tmp_path),))
    counts = {r[1] for r in res}
    assert len(counts) == 1 and counts.pop() == 4          # ceil(27 / 8) global batches: wrap-padded, the same count everywhere
    assert all(r[2] == 3 for r in res)
    for s in range(3):
        assert len({r[3][s] for r in res}) == 1, [r[3][s] for r in res]     # same reduced gradient sum, gathered loss and scale on all ranks


def _worker_advice_r3(rank, world, port, q):
    """(a) allreduce_grads() must not infer "already exchanged" from dit._dp being enabled (the captured-graph step and
    forward_backward never pass through the autograd node's exchange): only LoraGradSync.finish() vouches for it, and zero_grad /
    a local-only backward withdraw it; (b) broadcast_state() issues the SAME collectives on every rank when a non-src rank holds
    optimizer buffers src lacks (they are dropped); (c) the rank-0 broadcast repeats when the model's adapter set was rebuilt."""
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qflux_amd.dp import LoraGradSync
    from qflux_amd.trainer import QwenLoraTrainStep
    toy = _toy_model(with_cond=False)
    toy._dp = LoraGradSync(toy)            # what dit.enable_data_parallel() installs (add_adapter does it under a process group)
    toy._version = 0
    toy._adapter_gen = 0
    st = toy.lora_store
    step = QwenLoraTrainStep(toy)
    ok = True
    # (a) enabled, but no exchange ran: the full all-reduce must happen and the factor is 1/world
    st.gflat.fill_(float(rank + 1))
    f = step.allreduce_grads()
    tot = float(sum(range(1, world + 1)))
    ok = ok and f == 1.0 / world and bool(st.gflat.eq(tot).all())
    # after a drop-in backward's finish() the gradient IS exchanged and averaged: no second reduction
    step.zero_grad()
    ok = ok and toy._dp.exchanged is False
    st.gflat.fill_(float(rank + 1))
    toy._dp.hook()                          # arms finish()
    toy._dp.finish(average=True)
    ok = ok and toy._dp.exchanged is True and bool(st.gflat.eq(tot / world).all())
    f = step.allreduce_grads()
    ok = ok and f == 1.0 and bool(st.gflat.eq(tot / world).all())
    step.zero_grad()
    ok = ok and toy._dp.exchanged is False
    # (b) a NON-src rank carries moments, src has none: same collective sequence everywhere, the stray buffers are dropped
    if rank == 1:
        step._m = torch.full_like(st.pflat, 3.0)
        step._v = torch.full_like(st.pflat, 4.0)
    with torch.no_grad():
        st.pflat.fill_(float(10 + rank))
    step.broadcast_state()
    ok = ok and step._m is None and step._v is None and bool(st.pflat.eq(10.0).all())
    ok = ok and step.check_replicas()
    # (c) _ensure_synced: once per model version
    step._synced = False
    with torch.no_grad():
        st.pflat.fill_(float(20 + rank))
    step._ensure_synced()
    ok = ok and bool(st.pflat.eq(20.0).all())
    with torch.no_grad():
        st.pflat.fill_(float(30 + rank))
    step._ensure_synced()                   # same version: no broadcast
    ok = ok and bool(st.pflat.eq(30.0 + rank).all())
    toy._version += 1                       # a purely LOCAL plan rebuild (.to(), set_adapter, merge, quantize_trunk): NO collective --
    step._ensure_synced()                   # a rank-0-only validation / merge must not leave the ranks with mismatched broadcasts
    ok = ok and bool(st.pflat.eq(30.0 + rank).all())
    toy._adapter_gen += 1                   # add_adapter / load_lora_adapter / load_state_dict: rank 0's state goes out again
    step._ensure_synced()
    ok = ok and bool(st.pflat.eq(30.0).all())
    with torch.no_grad():
        st.pflat.fill_(float(40 + rank))
    step.resync()                           # the explicit form
    step._ensure_synced()
    ok = ok and bool(st.pflat.eq(40.0).all())
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_advice_r3_exchange_flag_broadcast_symmetry_and_resync_world2():
    assert _spawn(_worker_advice_r3, 2, 37500) == [(0, True), (1, True)]
