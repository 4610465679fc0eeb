#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3g"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
for n in 2500 5000 10000 20000 40000 80000; do
  python3 bench.py --config c5 --sites $n --steps 10 --blocks 3 --no-cpu > "$O/c5_n$n.json" 2>> "$O/c5.err"
done
python3 - <<'PY'
import json,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3g"
for n in (2500,5000,10000,20000,40000,80000):
    d=json.load(open(O+"/c5_n%d.json"%n)); r=d["roofline"]
    print(n, d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["avg_launch_ms"], "us/ksite", round(r["avg_launch_ms"]*1e3/n*1000,3), r["hbm_GBps"])
PY
