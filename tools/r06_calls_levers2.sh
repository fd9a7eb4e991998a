#!/bin/bash
# round 6: the plan-build-time levers re-measured with per-variant plans (tools/step_plan_ab.py) -- tools/step_ablate.py does not rebuild
# the launch programs, so its KEY=VAL variants only reach levers the LIBRARY reads per launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/step_plan_ab.py base,QFX_FUSE_QKNORM_BWD=0,QFX_ATTN_BWD=1pass --steps 20 --rounds 3 --out gpurun_out/r06_step_levers_v2.json 2>&1 | tail -6
