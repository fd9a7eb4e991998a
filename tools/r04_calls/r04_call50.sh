#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 300 python tools/step_ab.py cur,lnt --layers 6 --reps 5 --only "ln_" 2>&1 | tail -7
timeout 600 python tools/step_lib_ab.py cur,lnt --steps 15 --rounds 3 2>&1 | tail -2
