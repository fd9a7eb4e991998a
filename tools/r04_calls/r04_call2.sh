#!/bin/bash
set -u
R=$(pwd)
mkdir -p gpurun_out
export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -5 ) > gpurun_out/c2_gemm_tests.log 2>&1
( timeout 300 python tools/step_lib_ab.py pd0,base,pd1,pd2 --steps 20 --rounds 3 --out gpurun_out/c2_pd_lib_ab.json 2>&1 | tail -8 ) > gpurun_out/c2_pd_lib_ab.log 2>&1
( timeout 300 python tools/step_ab.py pd0,base,pd1,pd2 --layers 8 --reps 5 --only gemm --out gpurun_out/c2_pd_step_ab.json 2>&1 | tail -30 ) > gpurun_out/c2_pd_step_ab.log 2>&1
( timeout 300 python tools/tiles_ab.py --steps 12 --rounds 3 --policies legacy,n160 2>&1 | tail -3 | cut -c1-300 ) > gpurun_out/c2_tiles_ab.log 2>&1
cat gpurun_out/c2_gemm_tests.log | tail -3; cat gpurun_out/c2_pd_lib_ab.log; cat gpurun_out/c2_pd_step_ab.log
