#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_fullsize_cfgs_gpu.py tests/test_kernels_gpu.py -x -q -k "cfg4 or attention or norm_rope" 2>&1 | tail -3 )
timeout 300 python tools/attn_var_bench.py vscalar,base --S 2432 2>&1 | tail -1 | cut -c1-700
