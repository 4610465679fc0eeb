#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4e"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python3 -m pytest tests/test_gpu_repeats.py tests/test_gpu_c4_sharded.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?; tail -3 "$O/pytest.log"; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
python3 tools/c4_projection.py --steps 20 > "$O/proj.$rep.json" 2> "$O/proj.$rep.err"
python3 -c "
import json; p=json.load(open('$O/proj.$rep.json')); print(p['t1_ms'], p['shard_ms'], p['projected_tN_ms'], p['projected_speedup'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_shard" -- python3 "$R/tools/c4_projection.py" --shard-only 1 --steps 10 --warmup 3 --blocks 1 > "$O/trace_shard.log" 2>&1
python3 "$R/tools/trace_steps.py" "$O/trace_shard" > "$O/shard_steps.txt" 2>&1
rm -rf "$O/trace_shard"; head -12 "$O/shard_steps.txt"
