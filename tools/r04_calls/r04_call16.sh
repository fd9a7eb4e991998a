#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/attn_var_bench.py r3,base --S 1280:48,2560:24,5120:12,10240:6 2>&1 | tail -1
