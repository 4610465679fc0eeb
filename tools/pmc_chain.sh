#!/bin/bash
# SQ counters of the chain kernel on the caterpillar tree (where the time of a step goes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rm -rf $O/pmc_chain[0-9]*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_chain$i -- python3 bench.py --tree caterpillar --steps 3 --warmup 1 --no-cpu > $O/pmc_chain$i.log 2>&1 || { tail -5 $O/pmc_chain$i.log; exit 1; }
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_chain[0-9]')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'chain' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
