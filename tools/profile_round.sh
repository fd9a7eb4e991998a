#!/bin/bash
# Round profile of the bench command on one MI355X (run through gpurun from the repo root):
#   kernel-trace stats, then the PMC passes (each on its own: never combined with other trace domains).
# Outputs under gpurun_out/<tag>/ ; summaries are copied / converted into profiles/ by the caller (tools/pmc_to_json.py).
set -u
TAG=${1:-prof}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed --sustained-steps 0 --no-live-traffic"
CMD2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed --sustained-steps 0 --no-live-traffic"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- $CMD2 > $OUT/pmc_$C.log 2>&1
done
timeout 600 rocprofv3 --pmc MfmaUtil VALUBusy --kernel-trace --output-format csv -d $OUT/pmc_mfma -o p -- $CMD2 > $OUT/pmc_mfma.log 2>&1
# keep the merged-back payload small: drop the per-dispatch kernel traces of the PMC passes, keep stats + counter tables
find $OUT -name "*kernel_trace.csv" -path "*pmc_*" -delete
ls -la $OUT $OUT/stats 2>/dev/null | head -40
