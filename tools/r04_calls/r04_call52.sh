#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 1200 python tools/configs_r02.py 2>&1 | grep "^{" | cut -c1-260
timeout 600 python tools/flux_bench.py --steps 8 --warmup 3 2>&1 | tail -2 | cut -c1-400
