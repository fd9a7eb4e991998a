#!/usr/bin/env python
"""What the bucketed exchange costs on the step when RCCL really runs it (one-rank communicator, QFX_DP_FORCE=1: the collectives
are identities, so this is the framework cost -- async launches on RCCL's stream, event joins, handle draining -- not the xGMI
transfer):  python tools/rccl_one_rank.py  -> one JSON line (profiles/r03_rccl_one_rank.json)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for force in ("0", "1"):
    env = dict(os.environ, QFX_DP_FORCE=force, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", QFX_BENCH_INIT_PG="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-batch2",
                          "--no-fp8", "--no-dropin"], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
    res["forced_exchange" if force == "1" else "plain"] = json.loads(line[0])["ms_per_step"]
res["overhead_ms"] = round(res["forced_exchange"] - res["plain"], 3)
res["note"] = "one-rank RCCL communicator; 4 async all_reduce calls (24 MB buckets) per step behind the backward + drain; identities, no xGMI traffic"
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "rccl_one_rank.json"), "w"), indent=1)
