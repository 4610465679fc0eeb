#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "chain or tail or fusion or dna" > $O/t_gpu.txt 2>&1 || { tail -40 $O/t_gpu.txt; exit 1; }
tail -3 $O/t_gpu.txt
for v in 0 1 0 1 0 1; do PLL_AMD_NO_CHAIN_PAIR=$v timeout -k 10 300 python bench.py --no-cpu --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_CHAIN_PAIR=$v', d['value'], d['ms_per_step'], d['ms_per_step_min'], d['lnl'])"; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/bench.py --no-cpu --steps 20 > $O/prof_c2.log 2>&1
cd $R; python tools/kstats.py $O/prof_c2
