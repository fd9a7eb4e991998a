#!/bin/bash
# round 6: non-temporal stores for the GEMM outputs that are only read in the backward (GELU pre-activation, pre-gate output)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/step_lib_ab.py base,nt1,nt3 --steps 20 --rounds 3 --out gpurun_out/r06_nt_saved_ab.json 2>&1 | tail -4
