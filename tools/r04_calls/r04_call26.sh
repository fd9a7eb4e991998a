#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/step_ab.py cur,nld4 --layers 6 --reps 5 --only gemm 2>&1 | grep "n=2\|n=6\|TOTAL\|SUM\|class"
