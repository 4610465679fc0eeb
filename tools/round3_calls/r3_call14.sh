#!/bin/bash
# full GPU suite, then the default bench line and the C5 line (epilogue reorder, 8-wave workgroups)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3o"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > "$O/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a "$O/pytest.log"
tail -5 "$O/pytest.log"
[ $rc -eq 0 ] || exit 1
for c in c2 c5 c3; do
  python3 bench.py --config $c > "$O/${c}_bench.json" 2> "$O/${c}_bench.err" || exit 1
  python3 -c "
import json; p=json.loads(open('$O/${c}_bench.json').read().strip().splitlines()[-1]); print('$c', p['value'], p['ms_per_step'], p['ms_per_step_min'], p['ms_per_step_max'], p['roofline']['achieved'], p['roofline']['frac'], p['roofline']['kernel'])"
done
