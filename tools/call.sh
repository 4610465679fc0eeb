#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_repeats.py -x -q > $O/t_gpu.txt 2>&1 || { tail -60 $O/t_gpu.txt; exit 1; }
tail -3 $O/t_gpu.txt
