#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 300 python tools/step_ab.py cur,pf4,pf8 --layers 6 --reps 5 --only gemm 2>&1 | tail -22 | grep -v "n=1"
