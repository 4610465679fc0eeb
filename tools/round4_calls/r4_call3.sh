#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4c"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python3 -m pytest tests/test_gpu_bench_flow.py tests/test_gpu_c_caller.py tests/test_gpu_sharding_product.py -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?; tail -25 "$O/pytest.log"; exit $rc
