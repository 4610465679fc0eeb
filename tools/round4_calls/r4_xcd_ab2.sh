#!/bin/bash
# parity of everything the XCD-aware order touched, then A/B (PLL_AMD_NO_XCD_ORDER=1 = natural order) over the configurations
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_xcd_ab2"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_repeats.py -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?; tail -3 "$O/pytest.log"; [ $rc -eq 0 ] || exit $rc
run() { # label, args..., -- env...
  local label=$1; shift
  local args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" python3 bench.py "${args[@]}" --steps 20 --no-cpu > "$O/x.json" 2> "$O/x.err" || { echo "$label FAILED"; tail -3 "$O/x.err"; return; }
  python3 - "$O/x.json" "$label" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:36s} value {d['value']:9.1f} ms/step {d['ms_per_step']:.4f} [{d['ms_per_step_min']:.4f} {d['ms_per_step_max']:.4f}] {r['kernel'][:26]:26s} {r['avg_launch_ms']*1e3:7.1f} us")
PY
}
for rep in 1 2; do
  for mode in xcd natural; do
    X="A=1"; [ $mode = natural ] && X="PLL_AMD_NO_XCD_ORDER=1"
    run "c2 $mode" --config c2 -- $X
    run "c2 random tree $mode" --config c2 --tree random -- $X
    run "c2 caterpillar $mode" --config c2 --tree caterpillar -- $X
    run "c3 $mode" --config c3 -- $X
    run "c3 no groups $mode" --config c3 -- $X PLL_AMD_FUSE_GENERIC=0
    run "c5 $mode" --config c5 -- $X
    run "c4 1M $mode" --config c4 -- $X
  done
done
