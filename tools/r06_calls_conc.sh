#!/bin/bash
# round 6: attention backward dQ || dK/dV on two streams (QFX_ATTN_BWD_CONC=1), and the non-temporal-store lever already measured
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/step_plan_ab.py base,QFX_ATTN_BWD_CONC=1 --steps 20 --rounds 3 --out gpurun_out/r06_attn_bwd_conc.json 2>&1 | tail -5
QFX_ATTN_BWD_CONC=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fulldepth_gpu.py -x -q 2>&1 | tail -4
