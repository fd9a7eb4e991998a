#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/find_nondet.py --blocks 3 --passes 4 --res 512 2>&1 | grep -v "^  " | tail -9
timeout 600 python tools/find_nondet.py --blocks 2 --passes 3 --res 1024 2>&1 | grep -v "^  " | tail -7
