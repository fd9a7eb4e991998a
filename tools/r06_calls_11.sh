#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/step_ablate.py --variants full,QFX_FUSE_QKNORM_BWD=0,QFX_ATTN_BWD=1pass,QFX_FUSE_QKNORM_BWD=0+QFX_ATTN_BWD=1pass --steps 20 --rounds 3 --out gpurun_out/r06_step_levers.json 2>&1 | tail -8
