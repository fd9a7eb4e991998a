#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
timeout 600 python tools/step_ab.py r3,prev,pk --layers 6 --reps 5 --out gpurun_out/step_ab_pk.json 2>&1 | tail -40
