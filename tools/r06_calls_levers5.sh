#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout 900 python -m pytest tests/test_fulldepth_gpu.py -x -q 2>&1 | tail -2; done
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "lora_grad or bit_reproducible" 2>&1 | tail -2
timeout 1200 python tools/step_plan_ab.py base,QFX_GRAD_DET=0 --steps 20 --rounds 3 --out gpurun_out/r06_step_grad_det_v2.json 2>&1 | tail -3
