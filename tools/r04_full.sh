#!/bin/bash
# full round-end validation on one MI355X: GPU test suite, smoke, default bench, rocprofv3 passes
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
rm -f gpurun_out/parity_observed.json
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) | tee gpurun_out/full_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2 ) | tee gpurun_out/full_smoke.log
( timeout 900 python bench.py 2>gpurun_out/full_bench.err | tail -1 ) > gpurun_out/full_bench.json; cut -c1-400 gpurun_out/full_bench.json
bash tools/profile_round.sh r04prof > gpurun_out/full_prof.log 2>&1; tail -5 gpurun_out/full_prof.log
