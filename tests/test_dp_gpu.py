"""Two data-parallel ranks sharing ONE GPU (gloo backend on device tensors: RCCL refuses duplicate devices): the real
launch programs, the bucketed all-reduce hooked into the segmented backward, fused clip+AdamW.  Replicas must stay
bit-identical and match a single-process reference that averages the two ranks' gradients."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _mk(device):
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    from common import TINY
    from parity_util import build_pair, tiny_embeddings
    _, hip = build_pair(dict(TINY), device=device)
    return hip, tiny_embeddings


def _worker(rank, world, port, q, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"      # RCCL: one device per rank; gloo: both ranks share device 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    hip, tiny_embeddings = _mk(dev)
    from qflux_amd.trainer import QwenLoraTrainStep
    step = QwenLoraTrainStep(hip, lr=1e-2, bucket_mb=1e-3)
    emb, noise, u = tiny_embeddings(seed=11 + rank)
    losses = []
    for _ in range(2):
        losses.append(step.train_step(emb, noise=noise, u=u).item())
    torch.cuda.synchronize()
    flat = hip.lora_store.pflat.detach().cpu().clone()
    q.put((rank, flat.numpy(), losses))   # by value: torch tensors would travel as shared-memory handles of a dead process
    dist.destroy_process_group()


def test_two_ranks_one_gpu_stay_identical_and_match_manual_average():
    _two_rank_check("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL variant needs two devices (RCCL refuses duplicate devices)")
def test_two_ranks_rccl_stay_identical_and_match_manual_average():
    """Same check over backend "nccl" (= RCCL over xGMI) with one device per rank: the bucketed async all-reduces run on RCCL's
    own stream, ordered against the main stream and the side gradient stream only by the events the step records
    (qwen_step.py _bucket_hook / allreduce_grads vs base_trainer.py:384-393).  Runs wherever >= 2 GPUs are visible."""
    _two_rank_check("nccl")


def _two_rank_check(backend):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (0 if backend == "gloo" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    f0, f1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(f0, f1), "replicas diverged"
    # single-process reference: average the two ranks' gradients by hand, same optimizer kernel
    hip, tiny_embeddings = _mk("cuda:0")
    from qflux_amd.trainer import QwenLoraTrainStep
    step = QwenLoraTrainStep(hip, lr=1e-2)
    embs = [tiny_embeddings(seed=11 + r) for r in range(2)]
    for _ in range(2):
        for emb, noise, u in embs:
            step.forward_backward(emb, noise=noise, u=u)       # grads accumulate (atomics into the flat buffer)
        step.optimizer_step(grad_scale=0.5)
        step.zero_grad()
    torch.cuda.synchronize()
    ref = hip.lora_store.pflat.detach().cpu()
    rel = ((f0 - ref).abs().max() / ref.abs().max()).item()
    print("2-rank vs manual average: rel", rel, "losses", res[0][2], res[1][2])
    assert rel < 1e-5    # fp32 atomics order differs, nothing else


def test_bench_two_ranks_share_one_gpu():
    """The driver's N>1 launch line (torch.distributed.run, one process per rank) on the bench itself, 2 DiT blocks, both ranks
    on device 0 over gloo: barrier / max-over-ranks timing / single JSON line from rank 0 / whole-job images per second."""
    import json
    import subprocess
    env = dict(os.environ, QFX_DIST_BACKEND="gloo", QFX_SHARE_GPU="1")
    port = 34500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "2",
           "--res", "256"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert abs(r["value"] - 2 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-3
    assert "cpu_baseline" not in r    # rank 0 at N=1 only


def test_bench_self_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts its own ranks (torch.distributed.run) and
    still prints exactly one JSON line carrying the data-parallel exchange report."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(QFX_DIST_BACKEND="gloo", QFX_SHARE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "2", "--res", "256"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2"
    dp = r["dp_exchange"]
    assert dp["bytes_per_step"] > 0 and dp["allreduce_alone_ms"] > 0 and dp["exposed_ms_per_step"] >= 0 and dp["hidden_ms_per_step"] >= 0
    assert dp["backend"] == "gloo" and dp["ranks"] == 2 and dp["replicas_checked"] is True


def _worker_dropin(rank, world, port, q, targets):
    """The reference's loop body on the drop-in module, two ranks: dit(...) -> loss.backward() -> clip_grad_norm_ ->
    torch.optim.AdamW.step() -> zero_grad(); the model exchanges its LoRA gradients inside backward (enable_data_parallel:
    DDP's autograd hooks never see kernel-written gradients).  One accumulation window of two micro-steps under no_sync()."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    from common import TINY
    from parity_util import build_pair, tiny_embeddings
    from qflux_amd.trainer import QwenLoraTrainStep
    _, hip = build_pair(dict(TINY), device="cuda:0", targets=targets)
    hip.enable_data_parallel(bucket_mb=1e-3)
    step = QwenLoraTrainStep(hip)          # used only for compute_loss (= the reference's _compute_loss on self.dit)
    params = hip.lora_parameters()
    opt = torch.optim.AdamW(params, lr=1e-2)
    e1 = tiny_embeddings(seed=11 + rank)
    e2 = tiny_embeddings(seed=21 + rank)
    with hip.no_sync():
        step.compute_loss(e1[0], noise=e1[1], u=e1[2]).backward()
    step.compute_loss(e2[0], noise=e2[1], u=e2[2]).backward()
    torch.cuda.synchronize()
    g = hip.lora_store.gflat.detach().cpu().clone()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    opt.zero_grad()
    torch.cuda.synchronize()
    q.put((rank, g.numpy(), hip.lora_store.pflat.detach().cpu().numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("targets", [("to_k", "to_q", "to_v", "to_out.0"), "all-linear"], ids=["attn", "all-linear"])
def test_dropin_autograd_two_ranks_exchange_inside_backward(targets):
    """Replicas identical after the step; the exchanged gradient equals the mean over ranks of the locally accumulated gradients
    (single-process reference), with adapters on the conditioning head too (their gradients are final only at the end of the
    backward program: ADVICE r2, high)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000) + (0 if isinstance(targets, tuple) else 7)
    procs = [ctx.Process(target=_worker_dropin, args=(r, 2, port, q, targets)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    g0, g1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(g0, g1) and torch.equal(torch.from_numpy(res[0][2]), torch.from_numpy(res[1][2])), "replicas diverged"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_util import build_pair, tiny_embeddings
    from common import TINY
    from qflux_amd.trainer import QwenLoraTrainStep
    _, hip = build_pair(dict(TINY), device="cuda:0", targets=targets)
    step = QwenLoraTrainStep(hip)
    for r in range(2):
        for seed in (11 + r, 21 + r):
            e = tiny_embeddings(seed=seed)
            step.compute_loss(e[0], noise=e[1], u=e[2]).backward()
    torch.cuda.synchronize()
    ref = hip.lora_store.gflat.detach().cpu() * 0.5
    rel = ((g0 - ref).abs().max() / ref.abs().max()).item()
    print("drop-in 2-rank exchanged gradient vs manual mean: rel", rel, "targets", targets)
    assert rel < 1e-5 and ref.abs().max() > 0


def _worker_rccl1(port, q, force):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if force:
        os.environ["QFX_DP_FORCE"] = "1"
    torch.cuda.set_device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    hip, tiny_embeddings = _mk("cuda:0")
    from qflux_amd.trainer import QwenLoraTrainStep
    step = QwenLoraTrainStep(hip, lr=1e-2, bucket_mb=1e-3)      # tiny buckets: one async RCCL all-reduce per DiT block
    assert step._force_dp == force
    losses = []
    for s_ in (11, 12, 13):
        emb, noise, u = tiny_embeddings(seed=s_)
        losses.append(step.train_step(emb, noise=noise, u=u).item())
    if force:
        assert step.broadcast_state() is None and step.check_replicas()
    torch.cuda.synchronize()
    q.put((force, hip.lora_store.pflat.detach().cpu().numpy(), losses))
    dist.destroy_process_group()


def test_rccl_code_path_on_one_rank_equals_the_plain_step():
    """RCCL ("nccl" backend) executes the exchange on a one-GPU box: a ONE-rank communicator with QFX_DP_FORCE=1 runs the bucketed
    async all-reduces (identities) on RCCL's own stream behind the segmented backward -- ordered against the main stream and the
    side gradient stream only by the events the step records -- plus broadcast_state / check_replicas; three optimisation steps
    must equal the same steps without any exchange up to the fp32 atomics' order."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for force in (True, False):
        q = ctx.Queue()
        p = ctx.Process(target=_worker_rccl1, args=(38500 + (os.getpid() % 2000) + int(force), q, force))
        p.start()
        f, flat, losses = q.get(timeout=300)
        p.join(60)
        res[f] = (torch.from_numpy(flat), losses)
    rel = ((res[True][0] - res[False][0]).abs().max() / res[False][0].abs().max()).item()
    print("RCCL one-rank exchange vs plain step: param rel", rel, "losses", res[True][1], res[False][1])
    assert rel < 1e-5 and all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(res[True][1], res[False][1]))
