#!/bin/bash
# round 4, GPU call 1: new full-depth parity tests, per-geometry GEMM test, tile-policy A/B on the headline step, LN rows-per-wave A/B, power trace
set -u
R=$(pwd)
mkdir -p gpurun_out
export PYTHONPATH=$R/qwen-image-finetune_amd:$R
( timeout 900 python -m pytest tests/test_fulldepth_gpu.py -x -q -s 2>&1 | tail -40 ) > gpurun_out/c1_fulldepth.log 2>&1
( timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -15 ) > gpurun_out/c1_gemm_tests.log 2>&1
( timeout 400 python tools/tiles_ab.py --steps 12 --rounds 3 2>&1 | tail -5 ) > gpurun_out/c1_tiles_ab.log 2>&1
( timeout 300 python tools/step_lib_ab.py base,lnr2,lnr4 --steps 20 --rounds 3 --out gpurun_out/c1_ln_lib_ab.json 2>&1 | tail -8 ) > gpurun_out/c1_ln_lib_ab.log 2>&1
( timeout 300 python tools/step_ab.py base,lnr2,lnr4 --layers 8 --reps 5 --only ln --out gpurun_out/c1_ln_step_ab.json 2>&1 | tail -30 ) > gpurun_out/c1_ln_step_ab.log 2>&1
( timeout 200 python tools/power_watch.py 2>&1 | tail -40 ) > gpurun_out/c1_power.log 2>&1
tail -3 gpurun_out/c1_fulldepth.log gpurun_out/c1_gemm_tests.log
tail -2 gpurun_out/c1_tiles_ab.log | cut -c1-1500
cat gpurun_out/c1_ln_lib_ab.log
