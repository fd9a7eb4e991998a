#!/bin/bash
# round-6 final records on one box: default bench, rocprofv3 kernel stats + PMC passes of the bench command
cd $GRAFT_REPO_ROOT
R=$(pwd); mkdir -p gpurun_out
( timeout 900 python bench.py 2>gpurun_out/r06_bench.err | tail -1 ) > gpurun_out/r06_bench_n1.json; cut -c1-300 gpurun_out/r06_bench_n1.json
bash tools/profile_round.sh r06prof > gpurun_out/r06_prof.log 2>&1; tail -3 gpurun_out/r06_prof.log
cp $(find gpurun_out/r06prof/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r06_rocprofv3_kernel_stats.csv
python tools/pmc_to_json.py gpurun_out/r06prof/pmc_FETCH_SIZE gpurun_out/r06prof/pmc_WRITE_SIZE -o gpurun_out/r06_pmc_hbm.json 2>&1 | tail -2
python tools/pmc_to_json.py --mfma gpurun_out/r06prof/pmc_mfma -o gpurun_out/r06_pmc_mfma.json 2>&1 | tail -2
rm -rf gpurun_out/r06prof/stats/*kernel_trace* 2>/dev/null; find gpurun_out/r06prof -name "*.csv" -size +4M -delete
