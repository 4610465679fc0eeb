#!/bin/bash
# round 4, first GPU call: the whole GPU suite (incl. the new multi-rank collective tests), default bench, C4 projection
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4a"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a "$O/pytest.log"
tail -8 "$O/pytest.log"
[ $rc -eq 0 ] || exit $rc
python3 bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.err"; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('$O/c2_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
python3 tools/c4_projection.py --steps 20 > "$O/c4_proj.json" 2> "$O/c4_proj.err"
python3 -c "
import json; p=json.load(open('$O/c4_proj.json')); print(p['t1_ms'], p['shard_ms'], p['exchange_us'], p['projected_tN_ms'], p['projected_speedup'])"
