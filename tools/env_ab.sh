#!/bin/bash
# A/B of an environment switch on the headline step: tools/env_ab.sh VAR valA valB [rounds] [extra bench flags]
# (alternating bench.py runs on one box; prints ms per step and the loss of each run)
VAR=$1; A=$2; B=$3; R=${4:-2}; shift 4 2>/dev/null
for i in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$VAR=$v', 'ms_per_step', d['ms_per_step'], 'loss', d['config'].get('loss'), 'lora_down ms/step', d.get('hbm_kernels', {}).get('qfx_lora_down', {}).get('ms_per_step'))
"
  done
done
