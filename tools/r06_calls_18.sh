#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "lora_grad" 2>&1 | tail -2
timeout 900 python tools/step_lib_ab.py base,gch1k --steps 20 --rounds 3 --gflat-repro --out gpurun_out/r06_step_grad_ch2.json 2>&1 | tail -5
