#!/bin/bash
# the round's final library under the long-running checks: extended fuzz, hand-off stress, lifecycle soak
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_stress"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
PLL_FUZZ_SEEDS=400 timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q > "$O/fuzz400.log" 2>&1; echo "fuzz rc=$?"; tail -2 "$O/fuzz400.log"
PLL_AMD_FUSE_CC16=1 PLL_FUZZ_SEEDS=200 timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q > "$O/fuzz200_cc16.log" 2>&1; echo "fuzz cc16 rc=$?"; tail -2 "$O/fuzz200_cc16.log"
timeout -k 10 600 python3 tools/handoff_stress.py > "$O/handoff_stress.txt" 2>&1; echo "handoff rc=$?"; tail -3 "$O/handoff_stress.txt"
timeout -k 10 900 python3 tools/soak.py 200 > "$O/soak.txt" 2>&1; echo "soak rc=$?"; tail -3 "$O/soak.txt"
