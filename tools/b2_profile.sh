#!/bin/bash
# Per-kernel comparison of the step at per-GPU batch 1 vs 2 (rocprofv3 kernel trace of a short bench run each); run through gpurun.
set -u
R=$(pwd); OUT=$R/gpurun_out/b2prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in 1 2; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/b$B -o p -- python $R/bench.py --batch $B --steps 6 --warmup 3 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed > $OUT/b$B.log 2>&1
  python $R/tools/trace_gaps.py $(find $OUT/b$B -name "*kernel_trace.csv" | head -1) -o $OUT/b${B}_step.json > /dev/null
  find $OUT/b$B -name "*kernel_trace.csv" -delete
done
tail -2 $OUT/b1.log $OUT/b2.log
