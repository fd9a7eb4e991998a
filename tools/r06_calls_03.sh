#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_onepass_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/attn_onepass_bench.py --S 2432,8576,2432:24:2 2>&1 | tail -4
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/op_prof -o p -- python $GRAFT_REPO_ROOT/tools/attn_onepass_bench.py --S 2432 --rounds 3 > $GRAFT_REPO_ROOT/gpurun_out/op_prof.log 2>&1 )
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/op_prof/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if float(r['Percentage']) > 0.5: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
find gpurun_out/op_prof -name "*kernel_trace.csv" -delete
