#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or lora_linear" 2>&1 | tail -4 )
timeout 600 python tools/step_ab.py nom32,m32 --layers 6 --reps 5 --only gemm --out gpurun_out/step_ab_m32.json 2>&1 | tail -24
