#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -4 )
timeout 600 python tools/step_lib_ab.py r3,prev,pk --steps 20 --rounds 3 --out gpurun_out/step_lib_ab_pk.json 2>&1 | tail -2
timeout 300 python tools/attn_var_bench.py prev,pk --S 2432 2>&1 | tail -1
