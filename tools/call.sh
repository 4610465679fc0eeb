#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cherry or matrix_pipe or group" > $O/t_gpu.txt 2>&1 || { tail -40 $O/t_gpu.txt; exit 1; }
tail -3 $O/t_gpu.txt
for v in new base new base new base; do L=$R/libpll-2_amd/csrc/libpll_amd.so; [ $v = base ] && L=$R/libpll-2_amd/csrc/libpll_amd_base.so; PLL_AMD_LIB=$L timeout -k 10 300 python bench.py --config c3 --no-cpu --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['lnl'])"; done
