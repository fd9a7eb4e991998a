#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-batch2 --no-fp8 --no-dropin --no-hostfed"
for m in 1 0; do
  QFX_FUSE_HEAD_LORA=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c9_stats_$m -o p -- $B > $R/gpurun_out/c9_stats_$m.log 2>&1
done
cd $R
python - <<'P'
import csv,glob
for m in ('1','0'):
    f=glob.glob(f'gpurun_out/c9_stats_{m}/*kernel_stats.csv')[0]
    print('== fuse', m)
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if any(k in n for k in ('attn_','lora_down','head_reduce','ln_down','lora_grad')):
            print(f"  {n.split('(')[0][-50:]:52s} calls {r['Calls']:>5} avg_us {float(r['AverageNs'])/1e3:8.1f} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
P
find gpurun_out -name "*kernel_trace.csv" -path "*c9_*" -delete
