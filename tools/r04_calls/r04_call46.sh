#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 300 python tools/step_ab.py cur,linw --layers 6 --reps 5 --only "ln_down" 2>&1 | tail -5
