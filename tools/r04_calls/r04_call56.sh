#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "norm_rope" 2>&1 | tail -2 )
timeout 300 python tools/step_ab.py qk64,base --layers 6 --reps 5 --only "qk_norm" 2>&1 | tail -4
