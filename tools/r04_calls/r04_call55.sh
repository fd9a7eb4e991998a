#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "head_lora or out_projection" 2>&1 | tail -2 )
timeout 300 python tools/step_ab.py hr8,base --layers 6 --reps 5 --only "head_reduce" 2>&1 | tail -4
