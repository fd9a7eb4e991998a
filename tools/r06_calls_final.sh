#!/bin/bash
# round 6, final-state records: kernel stats + PMC passes of the bench command, the default bench line, per-call profile, det-vs-atomics A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round.sh r06_final > gpurun_out/r06_final_profile.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r06_bench_final.json; cut -c1-400 gpurun_out/r06_bench_final.json
timeout 400 python tools/stepprof.py 2>&1 | grep -i "gemm\|attn" | head -24; cp gpurun_out/stepprof.json gpurun_out/r06_stepprof_final.json
timeout 900 python tools/step_plan_ab.py base,QFX_GRAD_DET=0 --steps 20 --rounds 3 --out gpurun_out/r06_grad_det_final.json 2>&1 | tail -3
