#!/bin/bash
# round 6: three copies of the side launches' scratch operands: join and fork as adjacent packets (QFX_SIDE_COPIES=3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/step_plan_ab.py base,QFX_SIDE_COPIES=3 --steps 20 --rounds 3 --out gpurun_out/r06_side_copies.json 2>&1 | tail -4
QFX_SIDE_COPIES=3 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fulldepth_gpu.py -x -q 2>&1 | tail -3
