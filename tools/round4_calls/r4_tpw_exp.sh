#!/bin/bash
# tiles per wave of the 4 x 4 update launches at sizes where more than 4096 workgroups exist
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_tpw_exp"; mkdir -p "$O"; cd "$R"
run() { # label, cfg, sites, env...
  local label=$1 cfg=$2 sites=$3; shift 3
  env "$@" python3 bench.py --config $cfg --sites $sites --steps 10 --blocks 3 --no-cpu > "$O/x.json" 2> "$O/x.err" || { echo "$label FAILED"; tail -3 "$O/x.err"; return; }
  python3 - "$O/x.json" "$label" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:44s} ms/step {d['ms_per_step']:.4f}  {r['kernel'][:28]:28s} {r['avg_launch_ms']*1e3:7.1f} us  {r['achieved']:.0f} GB/s")
PY
}
for rep in 1 2; do
for T in 0 1 2; do
  X="PLL_AMD_X_TPW=$T"; [ $T = 0 ] && X="A=1"
  run "c2 400k tpw=$T" c2 400000 $X
  run "c2 400k one producer level tpw=$T" c2 400000 PLL_AMD_NO_FUSE_CC=1 $X
  run "c2 400k no fusion tpw=$T" c2 400000 PLL_AMD_NO_FUSE=1 $X
  run "c4 1M tpw=$T" c4 1000000 $X
  run "c2 100k tpw=$T" c2 100000 $X
done
done
