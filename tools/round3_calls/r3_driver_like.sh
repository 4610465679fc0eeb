#!/bin/bash
# what the driver runs at round end: smoke(), then the default bench line
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/driverlike"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1 || { tail -20 "$O/smoke.log"; exit 1; }
tail -2 "$O/smoke.log"
SECONDS=0; python3 bench.py > "$O/bench.json" 2> "$O/bench.err" || { tail -20 "$O/bench.err"; exit 1; }
echo "bench wall: ${SECONDS}s"
python3 -c "
import json; p=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(p['metric'], p['value'], p['n_gpus'], p['steps'], p['warmup'], p['ms_per_step'], p['scaling'], p['dtype'], p['roofline']['frac'], p['roofline']['traffic'], p['roofline'].get('traffic_source',{}).get('file'), p['cpu_baseline']['value'], p['lnl_rel_err_pinned'])"
python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu | python3 -c "import sys,json; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('explicit', p['value'], p['steps'], p['warmup'])"
