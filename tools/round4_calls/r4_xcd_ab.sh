#!/bin/bash
# A/B of the XCD-aware workgroup order of the store-bound launches (PLL_AMD_NO_XCD_ORDER=1 = natural order): bench lines
# alternated on one box; usage: r4_xcd_ab.sh <tag> <config> [bench args ...]
R="$GRAFT_REPO_ROOT"; TAG=$1; CFG=$2; shift 2
O="$R/gpurun_out/r4_xcd_$TAG"; mkdir -p "$O"; cd "$R"
for rep in 1 2 3; do
  for mode in xcd natural; do
    if [ $mode = natural ]; then export PLL_AMD_NO_XCD_ORDER=1; else unset PLL_AMD_NO_XCD_ORDER; fi
    python3 bench.py --config $CFG "$@" --steps 20 --no-cpu > "$O/$mode.$rep.json" 2> "$O/$mode.$rep.err" || { echo "bench failed"; tail -5 "$O/$mode.$rep.err"; exit 1; }
    python3 - "$O/$mode.$rep.json" $mode <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:8s} value {d['value']:10.1f} ms/step {d['ms_per_step']:.4f} [{d['ms_per_step_min']:.4f} {d['ms_per_step_max']:.4f}]  {r['kernel']} {r['avg_launch_ms']*1e3:.1f} us frac {r['frac']:.3f}  lnl_err {d.get('lnl_rel_err_pinned')}")
PY
  done
done
