#!/bin/bash
# usage: pmc_sets.sh <outdir-name> <kernel substring> -- bench args...   (each counter set under its own timeout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ $# -lt 3 ]; then echo "usage: pmc_sets.sh <outdir-name> <kernel substring> -- bench args..." >&2; exit 2; fi
NAME=$1; KSUB=$2; shift 3
O=$R/gpurun_out/$NAME; mkdir -p $O
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_MFMA" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_INSTS_WAVE32_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  echo "set $i: $set"
  timeout -k 5 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu > $O/pmc_$i.log 2>&1 || echo "pmc set $i failed"
done
NAME=$NAME KSUB="$KSUB" python3 - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"+os.environ["NAME"]
ks=os.environ["KSUB"]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(O+"/pmc_*/")):
    fs=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        if ks not in r["Kernel_Name"]: continue
        kn=r["Kernel_Name"][:52]+" g"+r["Grid_Size"]
        acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn,c in acc.items():
    print(kn)
    for k,v in c.items(): print("   %-42s %14.0f" % (k, sum(v)/len(v)))
PY
