#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/gemm_prod_bench.py 2>&1 | tail -10
