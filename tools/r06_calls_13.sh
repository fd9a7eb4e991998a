#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -x -q 2>&1 | tail -3
timeout 1200 python tools/step_ablate.py --variants full,QFX_SIDE_MOD=0 --steps 20 --rounds 4 --out gpurun_out/r06_step_side_mod.json 2>&1 | tail -3
