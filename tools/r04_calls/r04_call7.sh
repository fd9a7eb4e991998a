#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 300 python tools/attn_var_bench.py base,dqpf --entry dq 2>&1 | tail -1 ) > gpurun_out/c7_attn_var.log 2>&1
cat gpurun_out/c7_attn_var.log
