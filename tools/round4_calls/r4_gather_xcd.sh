#!/bin/bash
# XCD-aware order for the gathering 4 x 4 launches (site repeats): parity, then the C4 projection with and without
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_gather_xcd"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout -k 10 600 python3 -m pytest tests/test_gpu_repeats.py tests/test_gpu_c4_sharded.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  for mode in xcd natural; do
    if [ $mode = natural ]; then export PLL_AMD_X_NO_GATHER_XCD=1; else unset PLL_AMD_X_NO_GATHER_XCD; fi
    python3 tools/c4_projection.py --steps 20 > "$O/proj_$mode.$rep.json" 2> "$O/proj_$mode.$rep.err"
    python3 -c "
import json; p=json.load(open('$O/proj_$mode.$rep.json')); print('$mode', p['t1_ms'], p['shard_ms'], p['projected_tN_ms'], p['projected_speedup'])"
  done
done
