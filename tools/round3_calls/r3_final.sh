#!/bin/bash
# the round's final evidence: full GPU suite, hand-off stress, then tools/profile_round.sh (one box)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/final"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > "$O/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a "$O/pytest.log"
tail -4 "$O/pytest.log"
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/handoff_stress.py > "$O/handoff_stress.txt" 2>&1 || { tail -5 "$O/handoff_stress.txt"; exit 1; }
tail -3 "$O/handoff_stress.txt"
bash tools/profile_round.sh > "$O/profile_round.log" 2>&1 || { tail -20 "$O/profile_round.log"; exit 1; }
for c in c2 c3 c3r c4 c5 c5_codes c5_dense; do python3 -c "
import json; p=json.loads(open('$R/gpurun_out/round/${c}_bench.json').read().strip().splitlines()[-1]); print('$c', p['value'], p['ms_per_step'], p['roofline']['achieved'], p['roofline']['frac'], p.get('repeats_update_ms'))"; done
python3 -c "
import json
for f in ('c4_projection','c4_projection_balanced_cuts'):
    p=json.load(open('$R/gpurun_out/round/%s.json'%f)); print(f, p['t1_ms'], p['shard_ms'], p['exchange_us'], p['projected_tN_ms'], p['projected_speedup'])"
cat "$R/gpurun_out/round/group_latency.txt"
