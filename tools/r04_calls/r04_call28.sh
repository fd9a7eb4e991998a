#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_mxfp8_gpu.py -x -q -k "gemm or lora_linear or mx" 2>&1 | tail -3 )
timeout 600 python tools/step_lib_ab.py cur,nld4 --steps 20 --rounds 3 --out gpurun_out/step_lib_ab_nld4.json 2>&1 | tail -2
