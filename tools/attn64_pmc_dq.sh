#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | head -40 > $R/gpurun_out/pmc_list.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  i=$((i+1))
  QFX_ATTN_DQ64=1 QFX_ATTN_FWD64=1 timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/dqpmc_$i -o p -- python $R/tools/attn_dq_only.py ${1:-2432} 3 > $R/gpurun_out/dqpmc_$i.log 2>&1
done
cd $R
python - <<'P'
import csv, glob, collections, json, re
out = {}
for f in glob.glob('gpurun_out/dqpmc_*/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r'attn_\w+', r['Kernel_Name'])
        if not m: continue
        acc[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
json.dump(out, open('gpurun_out/attn64_pmc_dq.json', 'w'), indent=1)
for k, v in out.items(): print(k, json.dumps({a: round(b) for a, b in v.items()}))
P
cat gpurun_out/pmc_list.txt | tr '\n' ' '; tail -2 gpurun_out/dqpmc_3.log | cut -c1-200
rm -rf gpurun_out/dqpmc_?
