#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/nondet_bisect.py base,pk1,pk1noslp,pk1wz,pk1o1 --S 2432 --reps 12 --entry dq --out nondet_bisect_flags.json > gpurun_out/nondet5.log 2>&1
tail -c 6000 gpurun_out/nondet5.log
