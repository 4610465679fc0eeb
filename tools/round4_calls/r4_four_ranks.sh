#!/bin/bash
# bench.py --gpus 4 / 6 starting its own ranks on ONE device (gloo control plane, shared-memory exchange): the N > 2 flow
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_ranks"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
for N in 4 6; do
  PLL_BENCH_SAME_DEVICE=1 timeout -k 10 500 python3 bench.py --gpus $N --backend gloo --sites 400000 --steps 5 --blocks 2 --warmup 2 > "$O/n$N.json" 2> "$O/n$N.err"; echo "N=$N rc=$?"
  python3 -c "
import json; d=json.load(open('$O/n$N.json')); print(d['n_gpus'], d['value'], d['t1_ms'], d['tN_ms'], d['speedup'], d['lnl_rel_err_vs_unsharded'], d['config']['sites_per_gpu'], d['exchange'])" || tail -5 "$O/n$N.err"
done
