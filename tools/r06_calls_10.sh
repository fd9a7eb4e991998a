#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/step_lib_ab.py base,gch4k,gch1k --steps 20 --rounds 3 --gflat-repro --out gpurun_out/r06_step_grad_ch.json 2>&1 | tail -8
