#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
echo "== default"; timeout 300 python tools/step_ab.py cur --layers 6 --reps 5 --only "gemm n=2 N=12288" 2>&1 | grep "N=12288"
echo "== no wide tile"; QFX_GEMM_TILES="160x192,256x128" timeout 300 python tools/step_ab.py cur --layers 6 --reps 5 --only "gemm n=2 N=12288" 2>&1 | grep "N=12288"
