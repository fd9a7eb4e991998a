#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/enqueue_time.py 2>&1 | tail -1
