#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/attn_onepass_bench.py --S 2432 --rounds 6 --libs a1,a3,a7,a15,a31,a4,a16 2>&1 | tail -2
