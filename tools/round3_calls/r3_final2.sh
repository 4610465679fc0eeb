#!/bin/bash
# tools/profile_round.sh alone (the first final call ran it right behind the stress test: a warm chip, clocks down)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/final"; mkdir -p "$O"
cd "$R"
bash tools/profile_round.sh > "$O/profile_round.log" 2>&1 || { tail -20 "$O/profile_round.log"; exit 1; }
for c in c2 c3 c3r c4 c5 c5_codes c5_dense; do python3 -c "
import json; p=json.loads(open('$R/gpurun_out/round/${c}_bench.json').read().strip().splitlines()[-1]); print('$c', p['value'], p['ms_per_step'], p['roofline']['achieved'], p['roofline']['frac'], p.get('repeats_update_ms'))"; done
python3 -c "
import json
for f in ('c4_projection','c4_projection_balanced_cuts'):
    p=json.load(open('$R/gpurun_out/round/%s.json'%f)); print(f, p['t1_ms'], p['shard_ms'], p['exchange_us'], p['projected_tN_ms'], p['projected_speedup'])"
cat "$R/gpurun_out/round/group_latency.txt"
