#!/bin/bash
# round 6: d(GELU) epilogue with the pre-activation rows through an LDS landing buffer -- parity, then A/B on the real step and per call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -6
timeout 600 python tools/step_lib_ab.py base,noauxdma,auxdma1 --steps 20 --rounds 3 --out gpurun_out/r06_auxdma_ab.json 2>&1 | tail -6
timeout 400 python tools/stepprof.py 2>&1 | grep -i "gemm" | head -30; cp gpurun_out/stepprof.json gpurun_out/r06_stepprof_auxdma.json
