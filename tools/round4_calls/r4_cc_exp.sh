#!/bin/bash
# experiments on k_partials_dna_cc at sizes beyond the Infinity Cache: store policy x tiles per wave x workgroup order
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_cc_exp"; mkdir -p "$O"; cd "$R"
run() { # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --config c2 --sites $SITES --steps 10 --blocks 3 --no-cpu > "$O/x.json" 2> "$O/x.err" || { echo "$label FAILED"; tail -3 "$O/x.err"; return; }
  python3 - "$O/x.json" "$label" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:34s} ms/step {d['ms_per_step']:.4f}  cc {r['avg_launch_ms']*1e3:7.1f} us  {r['achieved']:.0f} GB/s")
PY
}
for SITES in 400000 1000000 100000; do
  echo "== $SITES sites"
  run "default" A=1
  run "natural order" PLL_AMD_NO_XCD_ORDER=1
  run "all plain" PLL_AMD_X_STORES=1
  run "all plain, natural" PLL_AMD_X_STORES=1 PLL_AMD_NO_XCD_ORDER=1
  run "all nt" PLL_AMD_X_STORES=2
  run "tpw 1" PLL_AMD_X_TPW=1
  run "tpw 2" PLL_AMD_X_TPW=2
  run "tpw 8" PLL_AMD_X_TPW=8
  run "tpw 1, all plain" PLL_AMD_X_TPW=1 PLL_AMD_X_STORES=1
  run "tpw 2, all plain" PLL_AMD_X_TPW=2 PLL_AMD_X_STORES=1
  run "default" A=1
done
