#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_core_seam.py tests/test_gpu_c_caller.py tests/test_gpu_sharding_product.py tests/test_gpu_reference_programs.py -x -q -s > $O/t_gpu.txt 2>&1 || { tail -40 $O/t_gpu.txt; exit 1; }
grep "us per call" $O/t_gpu.txt; tail -3 $O/t_gpu.txt
