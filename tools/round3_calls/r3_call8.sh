#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3i"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest.log"
tail -6 "$O/pytest.log"
python3 bench.py --config c5 --steps 10 > "$O/c5_bench.json" 2> "$O/c5_bench.err"
python3 bench.py --config c3 --steps 10 > "$O/c3_bench.json" 2> "$O/c3_bench.err"
python3 bench.py --config c4 --steps 10 --no-cpu > "$O/c4_bench.json" 2> "$O/c4_bench.err"
cd /tmp && export TMPDIR=/tmp
for c in c3 c4 c5; do
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$c" -- python3 "$R/bench.py" --config $c --steps 10 --blocks 2 --no-cpu > /dev/null 2>&1
find "$O/prof_$c" -name "*kernel_stats.csv" -exec cp {} "$O/${c}_kernel_stats.csv" \;
rm -rf "$O/prof_$c"
done
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3i"
for c in ("c3","c4","c5"):
    d=json.load(open(O+"/%s_bench.json"%c)); r=d["roofline"]
    print(c, d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["avg_launch_ms"], d.get("repeats_update_ms"))
PY
head -12 "$O/c4_kernel_stats.csv"; head -8 "$O/c3_kernel_stats.csv"
