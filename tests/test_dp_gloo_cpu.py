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
