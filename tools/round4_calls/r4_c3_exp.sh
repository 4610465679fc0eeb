#!/bin/bash
# C3's group launch (k_partials_mfma_cc): workgroup order x items per wave; per-kernel times from rocprofv3
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_c3_exp"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cherry" 2>&1 | tail -2
run() { # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --config c3 --steps 20 --no-cpu > "$O/x.json" 2> "$O/x.err" || { echo "$label FAILED"; tail -3 "$O/x.err"; return; }
  python3 - "$O/x.json" "$label" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:30s} value {d['value']:8.1f} ms/step {d['ms_per_step']:.4f} [{d['ms_per_step_min']:.4f} {d['ms_per_step_max']:.4f}] lnl_err {d.get('lnl_rel_err_pinned')}")
PY
}
for rep in 1 2; do
  run "xcd order" A=1
  run "natural" PLL_AMD_NO_XCD_ORDER=1
  run "xcd ipw 4" PLL_AMD_X_IPW=4
  run "xcd ipw 2" PLL_AMD_X_IPW=2
  run "xcd ipw 1" PLL_AMD_X_IPW=1
  run "natural ipw 2" PLL_AMD_X_IPW=2 PLL_AMD_NO_XCD_ORDER=1
done
cd /tmp && export TMPDIR=/tmp
for mode in xcd natural; do
  if [ $mode = natural ]; then export PLL_AMD_NO_XCD_ORDER=1; else unset PLL_AMD_NO_XCD_ORDER; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tr_$mode" -- python3 "$R/bench.py" --config c3 --steps 10 --no-cpu > "$O/tr_$mode.log" 2>&1
  f=$(find "$O/tr_$mode" -name "*kernel_stats.csv" | head -1); echo "== $mode"; head -8 "$f" | cut -c1-150
done
rm -rf "$O"/tr_*/
