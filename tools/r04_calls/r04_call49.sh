#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 300 python tools/step_ab.py cur,gnt --layers 60 --reps 3 --only "mod_gemv" 2>&1 | tail -4
timeout 600 python tools/step_lib_ab.py cur,gnt --steps 15 --rounds 3 2>&1 | tail -2
