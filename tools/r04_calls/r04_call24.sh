#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
echo "== default policy"; timeout 600 python tools/step_ab.py nom32,m32 --layers 6 --reps 5 --only gemm 2>&1 | grep "n=2\|n=6\|TOTAL\|SUM\|class"
echo "== legacy tiles"; QFX_GEMM_TILES=legacy timeout 600 python tools/step_ab.py nom32,m32 --layers 6 --reps 5 --only gemm 2>&1 | grep "n=2\|n=6\|TOTAL\|SUM"
