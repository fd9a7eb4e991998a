#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 300 python tools/attn_var_bench.py cur-,cur --entry dq,dkv --S 2432 --qk 2>&1 | tail -1
