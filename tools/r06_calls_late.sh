#!/bin/bash
# round 6: the side-stream gradient launches of block i in front of block i-1's attention backward instead of its feed-forward GEMMs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/step_plan_ab.py base,QFX_SIDE_AT_ATTN=1 --steps 20 --rounds 3 --out gpurun_out/r06_side_at_attn.json 2>&1 | tail -4
QFX_SIDE_AT_ATTN=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fulldepth_gpu.py -x -q 2>&1 | tail -3
