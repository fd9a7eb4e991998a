#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -2 )
timeout 600 python tools/attn_var_bench.py u24,rot --entry fwd --S 2432,8576 2>&1 | tail -1
