#!/bin/bash
# shard cuts of the 1M-site configuration on 8 ranks: equal sizes, cost-balanced (round 3's form), cost-balanced with a size cap
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_cuts"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
for rep in 1 2; do
  for mode in equal uncapped cap1.04 cap1.03 cap1.05; do
    case $mode in
      equal) CUT=equal; unset PLL_SHARD_MAX_RATIO;;
      uncapped) CUT=balanced; export PLL_SHARD_MAX_RATIO=0;;
      cap*) CUT=balanced; export PLL_SHARD_MAX_RATIO=${mode#cap};;
    esac
    python3 tools/c4_projection.py --steps 20 --cut $CUT > "$O/proj_$mode.$rep.json" 2> "$O/proj_$mode.$rep.err"
    python3 -c "
import json; p=json.load(open('$O/proj_$mode.$rep.json')); print('$mode', p['t1_ms'], p['shard_sites'], p['shard_ms'], p['projected_tN_ms'], p['projected_speedup'])"
  done
done
