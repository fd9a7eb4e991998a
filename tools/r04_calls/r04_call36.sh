#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 900 python tools/find_nondet.py --blocks 2 --passes 4 2>&1 | tail -8
