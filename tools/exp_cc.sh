cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cc; mkdir -p $O
export PLL_AMD_FUSE_GENERIC=1 PLL_AMD_MFMA_MIN_STATES=17
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --config c3 --steps 10 --no-cpu > $O/log.txt 2>&1
python3 $R/tools/kstats.py $O/tr
python3 $R/tools/trace_steps.py $O/tr > $O/steps.txt; head -12 $O/steps.txt | cut -c1-40,60-140
rocprofv3 -L > $O/counters.txt 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_EA0_WRREQ_STALL_sum TCP_TCC_WRITE_REQ_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu > $O/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/cc"
for d in sorted(glob.glob(O+"/pmc_*/")):
    fs=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not fs: print(d,"no csv"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        kn=r["Kernel_Name"][:40]+" g"+r["Grid_Size"]
        acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn,c in acc.items():
        if "k_partials" in kn: print(kn, {k: round(sum(v)/len(v)) for k,v in c.items()})
PY
