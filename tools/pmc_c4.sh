cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rm -rf $O/c4_fetch $O/c4_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c4_fetch -- python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu > $O/c4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c4_write -- python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu > $O/c4_write.log 2>&1
python3 - <<'PY'
import csv, glob, collections
def rows(d, name):
    out=[]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r["Counter_Name"]==name and "k_partials_dna<false, false, true>" in r["Kernel_Name"]:
                out.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return out
f=rows("gpurun_out/c4_fetch","FETCH_SIZE"); w=rows("gpurun_out/c4_write","WRITE_SIZE")
agg=collections.defaultdict(lambda:[0,0,0,0])
for g,v in f: agg[g][0]+=v; agg[g][1]+=1
for g,v in w: agg[g][2]+=v; agg[g][3]+=1
for g,(fs,fn,ws,wn) in sorted(agg.items()):
    print(g, "fetch MB/launch", round(2*fs/fn*1024/1e6,1), "write MB/launch", round(ws/max(wn,1)*1024/1e6,1), "launches", fn)
PY
