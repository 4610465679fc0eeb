#!/bin/bash
# HBM counters of every launch of a traversal on an irregular tree against the bytes the library says the
# launches have to move: tools/pmc_tree.sh random|caterpillar
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-random}
O=gpurun_out
rm -rf $O/pt_fetch $O/pt_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pt_fetch -- python3 bench.py --tree $T --steps 3 --warmup 1 --no-cpu > $O/pt_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pt_write -- python3 bench.py --tree $T --steps 3 --warmup 1 --no-cpu > $O/pt_write.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, json, re
def rows(d, name):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r["Counter_Name"]==name and ("chain" in r["Kernel_Name"] or "dna_cc" in r["Kernel_Name"]):
                k=re.sub(r"\(.*","",r["Kernel_Name"].replace("void ",""))
                acc[(k,int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc
f=rows("gpurun_out/pt_fetch","FETCH_SIZE"); w=rows("gpurun_out/pt_write","WRITE_SIZE")
tot=0
for k in sorted(set(f)|set(w)):
    fm=2*sum(f.get(k,[0]))/max(1,len(f.get(k,[0])))*1024/1e6; wm=sum(w.get(k,[0]))/max(1,len(w.get(k,[0])))*1024/1e6
    print(k, "fetch MB", round(fm,1), "write MB", round(wm,1), "launches", len(f.get(k,[])))
    tot+=fm+wm
b=json.loads([l for l in open("gpurun_out/pt_fetch.log").read().split("\n") if l.startswith("{")][-1])
print("sum over distinct launches MB", round(tot,1), "algorithmic GB/s", b["roofline"]["full_traversal"]["algorithmic_GBps"], "ms", b["roofline"]["full_traversal"]["ms"])
PY
