#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or head_lora or out_projection" 2>&1 | tail -5 )
timeout 600 python tools/attn_var_bench.py r3,base,narrow,eager --S 2432,8576 2>&1 | tail -1
