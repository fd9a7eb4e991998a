#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "import_before or attention" 2>&1 | tail -3 )
timeout 600 python tools/attn_var_bench.py head0,base --entry dkv --S 2432 2>&1 | tail -1
