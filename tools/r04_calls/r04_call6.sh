#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/c6_gputests.log 2>&1
( timeout 600 python bench.py 2>&1 | tail -1 ) > gpurun_out/c6_bench.json 2>gpurun_out/c6_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c6_smoke.log 2>&1
tail -4 gpurun_out/c6_gputests.log; tail -2 gpurun_out/c6_smoke.log; cut -c1-1500 gpurun_out/c6_bench.json; cat gpurun_out/parity_observed.json
