#!/bin/bash
# fifteen-op groups (complete 16-tip subtrees): parity, then A/B against PLL_AMD_FUSE_CC16=0 on one box
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_cc16"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?; tail -3 "$O/pytest.log"; [ $rc -eq 0 ] || exit $rc
run() { # label, args..., -- env...
  local label=$1; shift
  local args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" python3 bench.py "${args[@]}" --steps 20 --no-cpu > "$O/x.json" 2> "$O/x.err" || { echo "$label FAILED"; tail -3 "$O/x.err"; return; }
  python3 - "$O/x.json" "$label" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:30s} value {d['value']:9.1f} ms/step {d['ms_per_step']:.4f} [{d['ms_per_step_min']:.4f} {d['ms_per_step_max']:.4f}] {r['kernel'][:26]:26s} {r['avg_launch_ms']*1e3:7.1f} us launches {r['full_traversal']['launches']} lnl_err {d.get('lnl_rel_err_pinned')}")
PY
}
for rep in 1 2 3; do
  run "c2 fifteen-op groups" --config c2 -- PLL_AMD_FUSE_CC16=1
  run "c2 seven-op groups" --config c2 -- PLL_AMD_FUSE_CC16=0
done
run "c2 400k fifteen" --config c2 --sites 400000 -- PLL_AMD_FUSE_CC16=1
run "c2 400k seven" --config c2 --sites 400000 -- PLL_AMD_FUSE_CC16=0
run "c2 128 taxa fifteen" --config c2 --taxa 128 -- PLL_AMD_FUSE_CC16=1
run "c2 128 taxa seven" --config c2 --taxa 128 -- PLL_AMD_FUSE_CC16=0
