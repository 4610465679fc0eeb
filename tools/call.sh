#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tail_fusion" > $O/t_gpu.txt 2>&1 || { tail -60 $O/t_gpu.txt; exit 1; }
tail -3 $O/t_gpu.txt
for v in 0 1 0 1 0 1; do PLL_AMD_NO_TAIL_FUSION=$v timeout -k 10 300 python bench.py --config c3 --no-cpu --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_TAIL_FUSION=$v', d['value'], d['ms_per_step'], d['lnl'])"; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 $R/bench.py --config c3 --no-cpu --steps 10 > $O/prof_c3.log 2>&1
cd $R; python tools/kstats.py $O/prof_c3
