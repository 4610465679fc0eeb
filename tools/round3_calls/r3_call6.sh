#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3f"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma61_probe.hip -o /tmp/mfma61_probe > /dev/null 2>&1 && timeout -k 5 200 /tmp/mfma61_probe > "$O/mfma61_probe.txt" 2>&1
cat "$O/mfma61_probe.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/clk" -- python3 "$R/bench.py" --config c5 --steps 5 --blocks 2 --warmup 3 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3f"
f=glob.glob(O+"/clk/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]!="GRBM_GUI_ACTIVE": continue
    dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-9
    acc[r["Kernel_Name"][:40]+" g"+r["Grid_Size"]].append((float(r["Counter_Value"])/8/dur/1e9, dur*1e6))
for k,v in acc.items():
    v=v[len(v)//2:]
    print("%-60s clock %.2f GHz  dur %.1f us  (n=%d)"%(k, sum(x[0] for x in v)/len(v), sum(x[1] for x in v)/len(v), len(v)))
PY
rm -rf "$O/clk"
