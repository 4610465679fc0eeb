#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_repeats.py tests/test_gpu_core_seam.py -x -q -s > $O/t_gpu.txt 2>&1 || { tail -60 $O/t_gpu.txt; exit 1; }
grep "us per call" $O/t_gpu.txt; tail -3 $O/t_gpu.txt
