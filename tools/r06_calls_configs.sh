#!/bin/bash
# round 6: the non-headline configurations at the final HEAD (Qwen sweep of tools/configs_r02.py, cfg #4 also under the one-pass lever, FLUX / cfg #5)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/configs_r02.py 2>&1 | grep '^{' > gpurun_out/r06_configs_qwen.jsonl; cat gpurun_out/r06_configs_qwen.jsonl | cut -c1-260
QFX_ATTN_BWD=1pass timeout 600 python - <<'PY' 2>&1 | grep '^{' | tee gpurun_out/r06_cfg4_1pass.json
import sys, os
sys.argv = ["x"]
src = open("tools/configs_r02.py").read()
head = src[:src.index("dit = model()\nstep = QwenLoraTrainStep(dit, lr=1e-4)\nfor B in (1, 2, 4):")]
head = head.replace('open(out_path, "w").close()', '')
exec(compile(head, "configs_head", "exec"))
dit = model(); step = QwenLoraTrainStep(dit, lr=1e-4)
dt, loss = timeit(step, embeddings(1, 64, 384), warm=2, n=4)
print(json.dumps({"case": "cfg#4 shape 1024^2 (S=8576) r=16 B=1, QFX_ATTN_BWD=1pass", "ms_per_step": round(dt * 1e3, 1), "loss": round(loss, 4)}))
PY
timeout 400 python tools/flux_bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/r06_flux_shared_final.json | cut -c1-300
timeout 400 python tools/flux_bench.py --steps 10 --warmup 3 --multires 20x20,40x40 2>&1 | tail -1 | tee gpurun_out/r06_flux_multires_final.json | cut -c1-300
timeout 400 python tools/flux_bench.py --steps 10 --warmup 3 --multires 20x20,32x32 2>&1 | tail -1 | tee gpurun_out/r06_flux_multires_b_final.json | cut -c1-300
