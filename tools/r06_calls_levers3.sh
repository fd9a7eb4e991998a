#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/step_plan_ab.py base,QFX_GRAD_DET=0,QFX_SIDE_MOD=1 --steps 20 --rounds 3 --out gpurun_out/r06_step_levers_v2b.json 2>&1 | tail -6
