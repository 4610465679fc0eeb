#!/bin/bash
# rocprofv3 counter sets for the kernels of one C4 shard step (tools/c4_projection.py --shard-only 1): pmc_shard.sh <kernel substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KSUB=$1
O=$R/gpurun_out/pmc_shard; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/c4_projection.py --shard-only 1 --steps 3 --warmup 1 > $O/p$i.log 2>&1 || echo "set $i failed"
done
KSUB="$KSUB" python3 - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_shard"
ks=os.environ["KSUB"]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(O+"/p*/")):
    fs=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        if ks not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"][:52]+" g"+r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn,c in acc.items():
    print(kn)
    for k,v in c.items(): print("   %-34s %14.0f" % (k, sum(v)/len(v)))
PY
