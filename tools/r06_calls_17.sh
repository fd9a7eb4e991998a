#!/bin/bash
cd $GRAFT_REPO_ROOT
QFX_SHARE_GPU=1 QFX_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --layers 8 2>&1 | tail -1 | cut -c1-1300
timeout 900 python tools/step_lib_ab.py base,gch256,gch1k --steps 20 --rounds 3 --gflat-repro --out gpurun_out/r06_step_grad_ch2.json 2>&1 | tail -7
