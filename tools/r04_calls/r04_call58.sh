#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 900 python tools/soak.py 2>&1 | tail -2 | cut -c1-900
