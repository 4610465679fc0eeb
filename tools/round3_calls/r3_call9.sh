#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3j"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 900 python3 -m pytest tests/test_gpu_repeats.py tests/test_gpu_core_seam.py tests/test_gpu_c4_sharded.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q -x > "$O/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest.log"
tail -6 "$O/pytest.log"
python3 bench.py --config c4 --steps 10 --no-cpu > "$O/c4_bench.json" 2> "$O/c4_bench.err"
python3 -c "
import json; d=json.load(open('$O/c4_bench.json')); print('c4', d['value'], d['ms_per_step'], 'repeats_update_ms', d['repeats_update_ms'], d['traversal_with_class_maps_ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c4" -- python3 "$R/bench.py" --config c4 --steps 5 --blocks 1 --no-cpu > /dev/null 2>&1
find "$O/prof_c4" -name "*kernel_stats.csv" -exec cp {} "$O/c4_kernel_stats.csv" \;
rm -rf "$O/prof_c4"
grep "k_rep\|fillBuffer" "$O/c4_kernel_stats.csv"
