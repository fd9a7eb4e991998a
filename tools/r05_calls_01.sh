#!/bin/bash
# round 5, call 1: bisect of the dQ-epilogue run-to-run differences + a same-box baseline bench
cd $GRAFT_REPO_ROOT
V=base,pk1,pk1f0,pk1f1,pk1f2,pk1f3,pk1f4,pk1f5,pk1f6,pk1f7,pk1ns
timeout 600 python tools/nondet_bisect.py $V --S 8576 --reps 30 --out nondet_bisect_s8576.json > gpurun_out/nondet1.log 2>&1
timeout 300 python tools/nondet_bisect.py base,pk1 --S 8576 --reps 30 --hl 0 --out nondet_bisect_s8576_nohl.json > gpurun_out/nondet2.log 2>&1
timeout 300 python tools/nondet_bisect.py base,pk1 --S 2432 --reps 60 --out nondet_bisect_s2432.json > gpurun_out/nondet3.log 2>&1
timeout 300 python tools/nondet_bisect.py base,pk1 --S 8576 --reps 30 --busy 1 --out nondet_bisect_s8576_busy.json > gpurun_out/nondet4.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed > gpurun_out/bench_call01.json 2> gpurun_out/bench_call01.err
tail -3 gpurun_out/nondet*.log; cat gpurun_out/bench_call01.json | cut -c1-400
