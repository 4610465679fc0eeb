#!/bin/bash
# 8-wave vs 4-wave workgroups of k_partials_mfma_wide: parity, then the C5 bench under both with per-kernel stats
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3n"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_matrix or mfma or 61" > "$O/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a "$O/pytest.log"
tail -4 "$O/pytest.log"
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
for w in 1 4 1 4; do
  PLL_AMD_MFMA_WIDE=$w python3 "$R/bench.py" --config c5 --steps 20 --warmup 5 > "$O/c5_w$w.json" 2> "$O/c5_w$w.err" || exit 1
  python3 -c "
import json; p=json.loads(open('$O/c5_w$w.json').read().strip().splitlines()[-1]); print('w=$w', p['value'], p['ms_per_step'], p['roofline']['achieved'], p['roofline']['frac'])"
done
for w in 1 4; do
  PLL_AMD_MFMA_WIDE=$w rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_w$w" -- python3 "$R/bench.py" --config c5 --steps 10 --warmup 3 --blocks 1 > "$O/prof_w$w.log" 2>&1 || exit 1
  f=$(find "$O/prof_w$w" -name "*kernel_stats.csv" | head -1); head -6 "$f"; cp "$f" "$O/c5_w${w}_kernel_stats.csv"
  python3 "$R/tools/trace_steps.py" "$O/prof_w$w" > "$O/steps_w$w.txt" 2>&1; head -30 "$O/steps_w$w.txt"
  rm -rf "$O/prof_w$w"
done
