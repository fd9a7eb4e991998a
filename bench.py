#!/usr/bin/env python
"""bench.py -- training images/s of the Qwen-Image-Edit LoRA step (BASELINE.json configs[1]) on MI355X.

A "step" = one full optimisation step of the hot path on one batch of synthetic cached embeddings
per GPU: flow-match prepare -> 60-block DiT forward -> criterion -> DiT backward (dX + LoRA dA/dB) ->
all-reduce of the LoRA gradients (N>1) -> global-norm clip + AdamW.  Inputs are resident in HBM when the
timed region starts.  Random-init weights of the real architecture (no network for checkpoints).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README/DESIGN for the field meanings).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))

BF = torch.bfloat16
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_flops(L, D, S_i, T, Jd, Cin, Cout, r, n_tgt):
    """MACs x 2, B = 1, frozen base (dX only), no recompute."""
    S = S_i + T
    f_lin = 2 * (L * ((S_i + T) * 12 * D * D + 2 * 6 * D * D) + S_i * Cin * D + T * Jd * D + 256 * D + 3 * D * D + S_i * D * Cout)
    f_attn = 2 * L * 2 * S * S * D
    lora = 2 * L * n_tgt * S_i * 2 * D * r
    fwd = f_lin + f_attn + lora
    bwd = f_lin + 2.5 * f_attn + 2 * lora
    return fwd, bwd


def gemm_flops_of(prog):
    """Algorithmic flops of every qfx_gemm_bf16 launch of a launch program (from its argument structs)."""
    from qflux_amd import _lib as L
    tot = 0
    n = 0
    for g in prog.keep:
        if isinstance(g, L.GemmArgs):
            tot += 2 * g.M * g.N * (g.K1 + g.K2)
            n += 1
        elif isinstance(g, C.Array) and len(g) and isinstance(g[0], L.GemmArgs):
            tot += sum(2 * x.M * x.N * (x.K1 + x.K2) for x in g)
            n += 1
    return tot, n


def gemm_bytes_of(*progs):
    """Algorithmic HBM bytes of the GEMM launches: every operand and output touched once (bf16), aux/bias included."""
    from qflux_amd import _lib as L
    tot = 0
    for prog in progs:
        for g in prog.keep:
            gs = [g] if isinstance(g, L.GemmArgs) else (list(g) if isinstance(g, C.Array) and len(g) and isinstance(g[0], L.GemmArgs) else [])
            for x in gs:
                k = x.K1 + x.K2
                outs = 2 if x.epi == L.EPI_GELU else 1
                aux = 1 if x.epi in (L.EPI_GATE_RES, L.EPI_DGELU) else 0
                tot += 2 * (x.M * k + x.N * k + (outs + aux) * x.M * x.N)
    return tot


def hbm_bytes_of_call(name, args):
    """Algorithmic HBM bytes of one launch of a bandwidth-bound entry point, from its argument structs (bf16 streams touched once;
    rank-side operands included, the few-KB outputs of the rank-r kernels ignored)."""
    from qflux_amd import _lib as L

    def structs(a, n=None):
        obj = getattr(a, "_obj", a)          # ctypes.byref(struct) keeps the struct in ._obj
        return [obj] if n is None else [obj[i] for i in range(n)]

    if name in ("qfx_lora_down", "qfx_lora_down_batch"):
        return sum(2 * x.M * x.K + 4 * x.R * x.K for x in structs(args[0], args[1] if name.endswith("batch") else None))
    if name in ("qfx_lora_grad", "qfx_lora_grad_batch"):
        return sum(2 * x.M * x.K + 4 * x.R * x.M + 4 * x.R * x.K for x in structs(args[0], args[1] if name.endswith("batch") else None))
    if name == "qfx_lora_head_reduce":       # (list, n): H per-head fp32 partial slabs of [M, R] read once (round 4, ABI 6)
        return sum(4 * x.H * x.M * x.R for x in structs(args[0], args[1]))
    if name == "qfx_ln_modulate_fwd_batch":
        return sum(4 * x.rows * x.D for x in structs(args[0], args[1]))
    if name == "qfx_ln_modulate_bwd_batch":
        return sum(2 * x.rows * x.D * (3 + (1 if x.dres else 0) + (1 if x.dyg else 0)) for x in structs(args[0], args[1]))
    if name == "qfx_mod_gemv":               # (temb, B, K, W, bias, nmat, N, silu, out)
        return 2 * args[5] * args[6] * args[2]
    if name in ("qfx_qk_norm_rope_fwd", "qfx_qk_norm_rope_bwd"):
        # q and k rows of the joint buffer.  Forward (out of place since round 2): ONE read of the pre-norm rows + ONE write of the
        # normalised / rotated rows = 2 passes (profiles/r03_pmc_hbm.json: 33.7 + 29.9 MB per launch; the round-3 line counted 3 and
        # overstated the rate by 1.5x -- VERDICT r3 weak #4).  Stand-alone backward (unused since its fusion into the attention
        # epilogues): gradient read + saved rows read + gradient written = 3.
        Bq, Sq, Hq, dh = args[7], args[8], args[10], args[11]
        return 2 * Bq * Sq * 2 * Hq * dh * (2 if name.endswith("fwd") else 3)
    return None


def run_profiled(prog, fn_target, hbm=None):
    """Replay a program with a HIP event pair around every launch of `fn_target` (returned as a list) and, when `hbm` is a dict,
    around every bandwidth-bound entry point too (hbm[name] -> [events, bytes, launches])."""
    st_obj = torch.cuda.current_stream()
    st = st_obj.cuda_stream
    evs = []
    for ent in prog.calls:
        fn, args = ent[0], ent[1]      # (side-stream entries carry a third field; the profiled replay keeps one stream)
        if fn is None:
            args()
            continue
        nbytes = hbm_bytes_of_call(fn.__name__, args) if hbm is not None else None
        if fn in fn_target or nbytes is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st_obj)
            rc = fn(*args, st)
            e1.record(st_obj)
            if fn in fn_target:
                evs.append((e0, e1))
            else:
                rec = hbm.setdefault(fn.__name__.replace("_batch", ""), [[], 0, 0])
                rec[0].append((e0, e1)); rec[1] += nbytes; rec[2] += 1
        else:
            rc = fn(*args, st)
        if rc != 0:
            raise RuntimeError(f"{fn.__name__} -> {rc}")
    return evs


def _physical_cores():
    try:                                   # all PHYSICAL cores of the box (SMT siblings add nothing to fp32 GEMMs)
        import psutil
        cores = psutil.cpu_count(logical=False) or (os.cpu_count() or 1)
    except Exception:  # noqa: BLE001
        cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))     # ... that this process may run on
    except AttributeError:
        pass
    return cores


def cpu_baseline(cfg_dims, blocks=2, warmup=1, steps=3):
    """The oracle (a CPU restatement of the reference's module graph, fp32 eager PyTorch) timed on this host's cores on a
    BOUNDED sample (BASELINE.md section 3 protocol): K=2 of the 60 blocks + head/tail at the full sequence length,
    forward+backward+AdamW on the LoRA params, 1 warm-up step (allocator, thread pool) + 3 timed steps; the per-block time is
    measured as (K-block step) / K with head/tail included, extrapolated x(60/K).  Thread counts: ALL physical cores (the
    protocol) and, when the box has more than 64, also 64 -- eager CPU GEMMs stop scaling (and start thrashing) beyond that;
    `value` is the FASTER of the two, the other one is reported next to it."""
    sys.path.insert(0, ROOT)
    from oracle import qwen_dit as O
    D_h, H, Jd, S_t, T = cfg_dims
    phys = _physical_cores()
    runs = {}
    for cores in sorted({phys, min(phys, 64)}, reverse=True):
        torch.set_num_threads(cores)
        torch.manual_seed(1234)
        m = O.OracleQwenDiT(num_layers=blocks, attention_head_dim=D_h, num_attention_heads=H, joint_attention_dim=Jd)
        O.add_lora(m, r=16, lora_alpha=16, adapter_name="default", seed=0)
        opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
        emb = dict(image_latents=torch.randn(1, S_t, 64), control_latents=torch.randn(1, S_t, 64),
                   prompt_embeds=torch.randn(1, T, Jd) * 4, prompt_embeds_mask=torch.ones(1, T, dtype=torch.int64),
                   img_shapes=[[(1, 32, 32), (1, 32, 32)]])
        times = []
        for i in range(warmup + steps):
            t0 = time.time()
            loss = O.qwen_compute_loss(m, emb, torch.randn(1, S_t, 64), torch.rand(1), torch.float32)
            loss.backward()
            opt.step()
            opt.zero_grad()
            if i >= warmup:
                times.append(time.time() - t0)
        runs[cores] = times
        del m, opt
    best = min(runs, key=lambda c: sum(runs[c]) / len(runs[c]))
    times = runs[best]
    per_step = sum(times) / len(times)
    full = per_step * (60.0 / blocks)
    out = {"value": 1.0 / full, "unit": "images/s", "cores": best, "kind": "port",
           "sample": f"{blocks} of 60 DiT blocks + head/tail, fp32 eager, B=1, 512^2 (S_i={2 * S_t},T={T}), fwd+bwd+AdamW, "
                     f"{warmup} warm-up + {steps} timed steps of {per_step:.2f} s (min {min(times):.2f}, max {max(times):.2f}), "
                     f"extrapolated x{60 // blocks}", "physical_cores": phys}
    for c, t in runs.items():
        if c != best:
            out["other_thread_counts"] = {str(c): round(1.0 / (sum(t) / len(t) * 60.0 / blocks), 6)}
    return out


def _git_blob_sha1(path):
    """`git hash-object` of a file (so the bench line names exactly which committed PMC summary `traffic` came from)."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_traffic(kernel="gemm256_kernel"):
    """HBM-side traffic of the dominant kernel: PMC counters cannot be collected from inside this process; the newest committed
    rocprofv3 summary profiles/rNN_pmc_hbm.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, converted
    by tools/pmc_to_json.py with the gfx950 FETCH_SIZE x2 correction) gives bytes per launch of the same kernel on the same
    workload.  Returns (bytes per launch | None, {"file", "git_blob"})."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_hbm.json")))
    if not files:
        return None, None
    path = files[-1]
    try:
        with open(path) as fh:
            t = json.load(fh)["kernels"][kernel]["traffic_bytes_per_launch"]
        return t, {"file": os.path.relpath(path, ROOT), "git_blob": _git_blob_sha1(path)}
    except Exception:  # noqa: BLE001
        return None, None


def live_pmc_traffic(kernel="gemm256_kernel", budget_s=200):
    """HBM-side traffic of the dominant kernel measured IN THIS RUN (the committed summary is a constant the driver
    never observes): two rocprofv3 counter passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: separate passes; --pmc with --kernel-trace
    only -- MI355X_MICROARCH.md) over a 2-step run of this same command as child processes, folded by tools/pmc_to_json.py's rules
    (FETCH_SIZE x 2 on gfx950, KB per dispatch).  Returns (bytes per launch, source dict) or (None, None) when rocprofv3 is missing,
    refuses, or the passes do not fit the time budget -- the caller then falls back to the committed summary."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_to_json
    except Exception:  # noqa: BLE001
        return None, None
    tmp = tempfile.mkdtemp(prefix="qfx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    inner = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-batch2", "--no-fp8",
             "--no-dropin", "--no-hostfed", "--sustained-steps", "0", "--no-live-traffic"]
    t0 = time.perf_counter()
    try:
        acc = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            left = budget_s - (time.perf_counter() - t0)
            if left < 30:
                return None, None
            d = os.path.join(tmp, ctr)
            r = subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + inner,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left)
            if r.returncode != 0:
                return None, None
            acc[ctr] = pmc_to_json.read_pass(d, ctr).get(kernel)
        if not acc["FETCH_SIZE"] or not acc["WRITE_SIZE"]:
            return None, None
        (fn, fkb), (wn, wkb) = acc["FETCH_SIZE"], acc["WRITE_SIZE"]
        traffic = int(round(fkb / fn * 1024 * 2)) + int(round(wkb / wn * 1024))
        return traffic, {"measured_in_this_run": True, "passes": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace over 2 steps of this command",
                         "launches_counted": int(fn), "seconds": round(time.perf_counter() - t0, 1)}
    except Exception:  # noqa: BLE001  (timeout, unreadable output ...: the committed summary is the fallback)
        return None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_spawn(n):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run (one process per GPU, RCCL) and
    relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--batch", type=int, default=1, help="per-GPU micro batch")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--rank", type=int, default=16, help="LoRA rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch2", action="store_true", help="skip the secondary per-GPU-batch-2 measurement")
    ap.add_argument("--no-fp8", action="store_true", help="skip the secondary MX-FP8 trunk measurement")
    ap.add_argument("--no-dropin", action="store_true", help="skip the secondary drop-in (autograd + torch optimizer) measurement")
    ap.add_argument("--no-hostfed", action="store_true", help="skip the secondary disk-cache + PCIe inclusive measurement")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC summary instead of two rocprofv3 passes of this run")
    ap.add_argument("--sustained-steps", type=int, default=150, help="steps of the secondary sustained-throughput line (0 = skip)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args.gpus))

    from qflux_amd.models import QwenImageTransformer2DModel
    from qflux_amd.modules import LoraConfig
    from qflux_amd import _lib as L
    from qflux_amd.trainer import QwenLoraTrainStep
    from qflux_amd.trainer.qwen_step import init_distributed_from_env
    import torch.distributed as dist

    rank, local, world = init_distributed_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.manual_seed(1234)          # identical frozen trunk on every rank (data-parallel replicas)

    with torch.device(dev):
        dit = QwenImageTransformer2DModel(num_layers=args.layers)
    with torch.no_grad():
        for n, p in dit.named_parameters():
            if p.ndim == 2:
                p.normal_(0.0, 0.02)
            elif "norm" in n:
                p.fill_(1.0)
            elif ".img_mod." in n or ".txt_mod." in n or "norm_out" in n:
                p.normal_(0.0, 0.02)
            else:
                p.zero_()
    gen = torch.Generator().manual_seed(1234)  # identical LoRA init on every rank (the reference relies on equal seeds)
    dit.add_adapter(LoraConfig(r=args.rank, lora_alpha=args.rank, init_lora_weights="gaussian"), "default", generator=gen)
    step = QwenLoraTrainStep(dit, lr=1e-4, max_grad_norm=1.0)

    B = args.batch
    side = args.res // 16
    S_t = side * side
    T = 384
    Jd = dit.config.joint_attention_dim
    torch.manual_seed(1234 + rank)   # rank-local synthetic micro-batch (main.py:58 seed + rank)
    emb = dict(image_latents=torch.randn(B, S_t, 64).half().to(dev), control_latents=torch.randn(B, S_t, 64).half().to(dev),
               prompt_embeds=(torch.randn(B, T, Jd) * 4).half().to(dev), prompt_embeds_mask=None,
               img_shapes=[[(1, side, side), (1, side, side)]] * B)

    if world > 1:
        # communicator / channel set-up outside the timed region even with --warmup 0: one collective of the gradient's size
        dummy = torch.zeros_like(dit.lora_store.gflat)
        dist.all_reduce(dummy)
        torch.cuda.synchronize()
        del dummy
    for _ in range(args.warmup):
        step.train_step(emb)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = step.train_step(emb)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    # ---- dominant kernel (gemm_kernel): per-launch HIP-event timing on the launch stream, one extra replayed step
    plan = list(dit._plans.values())[0]
    dit.refresh_lora_operands()
    gemm_fns = (L.lib.qfx_gemm_bf16, L.lib.qfx_gemm_grouped)
    hbm = {}
    ev = run_profiled(plan.fwd, gemm_fns, hbm)
    ev += run_profiled(plan.bwd, gemm_fns, hbm)
    torch.cuda.synchronize()
    gemm_ms = sum(a.elapsed_time(b) for a, b in ev)
    gf_f, n_f = gemm_flops_of(plan.fwd)
    gf_b, n_b = gemm_flops_of(plan.bwd)
    n_launch = n_f + n_b
    achieved = (gf_f + gf_b) / (gemm_ms * 1e-3) / 1e12
    step.zero_grad()

    cfgd = dit.config
    fwd_fl, bwd_fl = algorithmic_flops(cfgd.num_layers, dit.inner_dim, 2 * S_t, T, Jd, cfgd.in_channels, dit.proj_out.out_features,
                                       args.rank, 4)
    traffic, traffic_src = (None, None)
    if B == 1 and args.layers == 60 and args.res == 512 and args.rank == 16:
        traffic, traffic_src = pmc_traffic()

    # ---- data-parallel exchange: what the bucketed all-reduce costs on top of the step, and how much of it the backward hides
    dp = None
    if world > 1:
        try:      # adapter weights + optimizer state identical on every rank after the timed steps (every rank reaches the same verdict)
            replicas_ok = bool(step.check_replicas())
        except RuntimeError as e:
            replicas_ok = False
            if rank == 0:
                print(f"[bench] replica check failed: {e}", file=sys.stderr)
        gflat = dit.lora_store.gflat
        k2 = max(2, min(args.steps, 10))
        step.world = 1                      # same step without the gradient exchange (measurement only: replicas would diverge)
        step.train_step(emb)
        dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            step.train_step(emb)
        dist.barrier(); torch.cuda.synchronize()
        noex = (time.perf_counter() - t1) / k2 * 1e3
        step.world = world
        dist.all_reduce(gflat); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(gflat)
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t1) / 5 * 1e3
        vals = torch.tensor([noex, alone], dtype=torch.float64, device=dev)
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        noex, alone = vals.tolist()
        exposed = max(0.0, ms_per_step - noex)
        dp = {"collective": "all_reduce(SUM) of the flat fp32 LoRA gradient, bucketed behind the backward", "backend": dist.get_backend(),
              "ranks": dist.get_world_size(), "replicas_checked": replicas_ok,
              "bytes_per_step": gflat.numel() * 4, "bucket_mb": step.bucket_bytes / (1 << 20),
              "ms_per_step_without_exchange": round(noex, 3), "allreduce_alone_ms": round(alone, 3),
              "exposed_ms_per_step": round(exposed, 3), "hidden_ms_per_step": round(max(0.0, alone - exposed), 3)}
        step.zero_grad()
    out = {
        "metric": "train images/sec, Qwen-Image-Edit LoRA r=16 bf16 512^2, cached-embed",
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (random-init weights of the real architecture, synthetic cached embeddings)",
        "config": {"workload": f"Qwen-Image-Edit DiT LoRA step: {cfgd.num_layers} blocks, D={dit.inner_dim}, 24x128 heads, "
                               f"{args.res}x{args.res} target + 1 control (S_i={2 * S_t}), T={T}, LoRA r={args.rank} on "
                               f"to_q/to_k/to_v/to_out.0, AdamW + clip, no recompute",
                   "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                   "step_tflop_algorithmic": round((fwd_fl + bwd_fl) * B / 1e12, 2),
                   "whole_step_tflops_per_gpu": round((fwd_fl + bwd_fl) * B / (ms_per_step * 1e-3) / 1e12, 1),
                   "loss": float(loss.item())},
        "roofline": {"bound": "mfma", "kernel": "gemm256_kernel / gemm_kernel (qfx_gemm_grouped + qfx_gemm_bf16, all epilogue variants)",
                     "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                     "traffic": traffic, "traffic_unit": "bytes per launch, L2 fabric side incl. Infinity-Cache hits (committed PMC pass)",
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": int(gemm_bytes_of(plan.fwd, plan.bwd) / n_launch),
                     "launches_per_step": n_launch, "avg_launch_us": round(gemm_ms * 1e3 / n_launch, 2),
                     "gemm_share_of_step": round(gemm_ms / ms_per_step, 3),
                     # rounds 4-5: what actually limits these launches (ablations, not this run)
                     "measured_limiter": "operand stream on the L2 -> CU vector-memory path (~32 B/clk/CU): 92 % of a launch remains with the matrix pipe idle, 94.5 % with the same requests as plain loads that never touch the LDS (profiles/r04_gemm_operand_stream.json, r05_gemm_stream_path.json); priced against the MFMA peak as the contract asks"},
        # the bandwidth-bound kernels of the same replayed step (serial replay: the weight-gradient launches are timed alone here,
        # in the step they overlap the main stream): algorithmic bytes / summed launch time against the 8 TB/s HBM3E peak
        "hbm_kernels": {k: {"GBps": round(v[1] / (sum(a.elapsed_time(b) for a, b in v[0]) * 1e-3) / 1e9, 1),
                            "frac_of_8TBps": round(v[1] / (sum(a.elapsed_time(b) for a, b in v[0]) * 1e-3) / 8e12, 3),
                            "ms_per_step": round(sum(a.elapsed_time(b) for a, b in v[0]), 3), "launches": v[2]}
                        for k, v in sorted(hbm.items())},
    }
    if dp is not None:
        out["dp_exchange"] = dp
    if world == 1 and B == 1 and not args.no_dropin:
        # secondary line: the DROP-IN path a reference user gets after the two-line swap of INTEGRATION.md -- the reference's own
        # loop body (base_trainer.py:508-561): loss = criterion(dit(...)[0]); loss.backward(); clip_grad_norm_(trainable, 1.0);
        # torch.optim.AdamW.step(); zero_grad() -- autograd node around the same launch programs, torch's optimizer over the per-tensor
        # LoRA parameters (views of the flat buffer)
        params = dit.lora_parameters()
        opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
        def dropin_step():
            l = step.compute_loss(emb)
            l.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            opt.zero_grad()
            return l
        for _ in range(3):
            dropin_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nd = max(4, min(args.steps, 10))
        for _ in range(nd):
            dropin_step()
        torch.cuda.synchronize()
        dtd = time.perf_counter() - t1
        out["dropin_autograd"] = {"value": round(nd / dtd, 4), "unit": "images/s", "ms_per_step": round(dtd / nd * 1e3, 3), "steps": nd,
                                  "vs_fused_step": round((dtd / nd * 1e3) / ms_per_step, 4),
                                  "note": "dit(...)[0] -> MSE -> loss.backward() -> clip_grad_norm_ -> torch.optim.AdamW.step() -> zero_grad(), "
                                          "the reference's loop body (base_trainer.py:508-561) on the drop-in module; `value` above is the fused step"}
        del opt
        step.zero_grad()
    if world == 1 and B == 1 and args.sustained_steps > 0:
        # secondary line: SUSTAINED throughput (`value` is a 20-40 step burst).  >= 150 back-to-back steps with no host
        # synchronisation inside the loop; per-step GPU time from one HIP event per step on the launch stream (durations between
        # consecutive events), wall clock around the whole run, package power / shader clock sampled by rocm-smi meanwhile.
        import gc
        import subprocess
        import threading
        ns = args.sustained_steps
        smp, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                try:
                    o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                    c = json.loads(o).get("card0", {})
                    pw = [float(v) for k_, v in c.items() if "power" in k_.lower() and "(w)" in k_.lower()]
                    ck = [float(str(v).strip("()").lower().replace("mhz", "")) for k_, v in c.items() if "sclk" in k_.lower() and "mhz" in str(v).lower()]
                    smp.append((pw[0] if pw else None, ck[0] if ck else None))
                except Exception:  # noqa: BLE001
                    pass
                stop.wait(0.5)
        th = threading.Thread(target=poll, daemon=True)
        for _ in range(5):
            step.train_step(emb)
        gc.collect()
        gc.freeze()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(ns + 1)]
        torch.cuda.synchronize()
        th.start()
        t1 = time.perf_counter()
        evs[0].record()
        for i in range(ns):
            step.train_step(emb)
            evs[i + 1].record()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        stop.set(); th.join(6)
        gc.unfreeze()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(ns))
        pws = [a for a, _ in smp if a is not None]
        cks = [b for _, b in smp if b is not None]
        out["sustained"] = {"value": round(ns / dts, 4), "unit": "images/s", "steps": ns, "ms_per_step_wall": round(dts / ns * 1e3, 3),
                            "ms_median": round(per[ns // 2], 3), "ms_p95": round(per[int(ns * 0.95)], 3), "ms_max": round(per[-1], 3),
                            "ms_min": round(per[0], 3), "p95_over_median": round(per[int(ns * 0.95)] / per[ns // 2], 4),
                            "vs_burst": round((dts / ns * 1e3) / ms_per_step, 4),
                            "power_w_mean": round(sum(pws) / len(pws), 1) if pws else None, "power_w_max": max(pws) if pws else None,
                            "sclk_mhz_mean": round(sum(cks) / len(cks), 1) if cks else None, "sclk_mhz_min": min(cks) if cks else None,
                            "samples": len(smp),
                            "note": "same resident batch as `value`, no host sync in the loop; per-step times = gaps between per-step HIP events"}
        step.zero_grad()
    if world == 1 and B == 1 and not args.no_hostfed:
        # secondary line: the step fed the way a training job feeds it -- the reference's on-disk embedding cache (fp16 .pt files,
        # SURVEY 8f1) -> PrefetchLoader (worker thread, pinned staging, upload on a side stream) -> train_step.  `value` above is
        # measured with the batch already resident in HBM; this is the disk + PCIe inclusive rate of the same step.
        import shutil
        import tempfile
        from qflux_amd.data import CachedEmbeddingDataset, PrefetchLoader, convert_img_shapes_to_latent_space, write_cache_sample
        root = tempfile.mkdtemp(prefix="qfx_bench_cache_")
        try:
            g = torch.Generator().manual_seed(0)
            for i in range(16):
                write_cache_sample(root, f"{i:032x}", dict(image_latents=torch.randn(S_t, 64, generator=g), control_latents=torch.randn(S_t, 64, generator=g),
                                                          prompt_embeds=torch.randn(T, Jd, generator=g) * 4, prompt_embeds_mask=torch.ones(T)),
                                   img_shapes=[(3, args.res, args.res), (3, args.res, args.res)])
            loader = PrefetchLoader(CachedEmbeddingDataset(root), batch_size=1, device=dev)
            nh = max(8, min(args.steps, 16))
            done, t1, nbytes = 0, None, 0
            while done < nh + 4:
                for b_ in loader:
                    e_ = dict(image_latents=b_["image_latents"], control_latents=b_["control_latents"], prompt_embeds=b_["prompt_embeds"],
                              prompt_embeds_mask=b_["prompt_embeds_mask"].long(), img_shapes=convert_img_shapes_to_latent_space(b_["img_shapes"]))
                    step.train_step(e_)
                    done += 1
                    if done == 4:
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                        nbytes = sum(v.numel() * v.element_size() for v in b_.values() if isinstance(v, torch.Tensor))
                    if done >= nh + 4:
                        break
            torch.cuda.synchronize()
            dth = (time.perf_counter() - t1) / nh
            out["host_fed"] = {"value": round(1.0 / dth, 4), "unit": "images/s", "ms_per_step": round(dth * 1e3, 3), "steps": nh,
                               "vs_resident": round(dth * 1e3 / ms_per_step, 4), "host_bytes_per_step": int(nbytes),
                               "note": "16-sample on-disk cache in the reference's layout -> CachedEmbeddingDataset -> PrefetchLoader "
                                       "(pinned staging, side-stream H2D) -> QwenLoraTrainStep.train_step; disk + PCIe inclusive"}
        finally:
            shutil.rmtree(root, ignore_errors=True)
        step.zero_grad()
    if world == 1 and B == 1 and not args.no_batch2:
        # secondary line: the reference's own default micro-batch for this config is 2 (configs/face_seg_config.yaml:31, and its
        # README numbers are quoted at bs 2).  At B=2 every GEMM runs >= 2 rounds per CU, so the per-launch fixed cost (pipeline
        # fill + exposed epilogue, ~12 us of a 45-170 us launch at B=1: profiles/r02_gemm_fixed_cost.json) is amortised.
        emb2 = {k: (torch.cat([v, v.flip(1)], 0) if isinstance(v, torch.Tensor) else v) for k, v in emb.items()}
        emb2["img_shapes"] = emb["img_shapes"] * 2
        for _ in range(3):
            step.train_step(emb2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n2 = max(4, min(args.steps, 10))
        for _ in range(n2):
            step.train_step(emb2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        # the GEMM launches of the B = 2 programs, timed per launch as for the headline (the narrow launches get the 256x256 tile here)
        plan2 = [pl for pl in dit._plans.values() if pl is not plan][-1]
        ev2 = run_profiled(plan2.fwd, gemm_fns, {})
        ev2 += run_profiled(plan2.bwd, gemm_fns, {})
        torch.cuda.synchronize()
        gemm_ms2 = sum(a.elapsed_time(b) for a, b in ev2)
        gf2 = gemm_flops_of(plan2.fwd)[0] + gemm_flops_of(plan2.bwd)[0]
        step.zero_grad()
        out["per_gpu_batch_2"] = {"value": round(2 * n2 / dt2, 4), "unit": "images/s", "ms_per_step": round(dt2 / n2 * 1e3, 3), "steps": n2,
                                  "gemm_frac_of_peak": round(gf2 / (gemm_ms2 * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "note": "same workload at the reference's default micro-batch (batch_size: 2); `value` above stays B=1"}
    if world == 1 and B == 1 and not args.no_fp8:
        # secondary line: the low-precision trunk (the reference's `model.quantize: true` analogue): forward + dX GEMMs of the block
        # linears in MX-FP8 on the block-scaled MFMA; NOT the headline precision (`dtype` above stays bf16)
        dit.quantize_trunk("mxfp8-fb")
        for _ in range(3):
            step.train_step(emb)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n8 = max(4, min(args.steps, 10))
        for _ in range(n8):
            step.train_step(emb)
        torch.cuda.synchronize()
        dt8 = time.perf_counter() - t1
        dit.quantize_trunk(None)
        out["mxfp8_trunk"] = {"value": round(n8 / dt8, 4), "unit": "images/s", "ms_per_step": round(dt8 / n8 * 1e3, 3), "steps": n8,
                              "mode": "quantize_trunk('mxfp8-fb'): MX-FP8 forward + dX GEMMs, bf16 everywhere else",
                              "note": "reduced-precision option, reported beside the bf16 headline, never as `value`"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline((cfgd.attention_head_dim, cfgd.num_attention_heads, Jd, S_t, T))
        except Exception as e:  # the baseline is a reported side number; never let it kill the bench line
            out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
    if rank == 0 and world == 1 and traffic is not None and not args.no_live_traffic:
        lt, lsrc = live_pmc_traffic()
        if lt is not None:
            out["roofline"]["traffic_committed_summary"] = {"bytes_per_launch": traffic, **(traffic_src or {})}
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = lt, lsrc
            out["roofline"]["traffic_unit"] = "bytes per launch, L2 fabric side incl. Infinity-Cache hits (rocprofv3 PMC passes of this run)"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
