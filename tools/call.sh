#!/bin/bash
# scratch driver for one gpurun call (overwritten from call to call; results under gpurun_out/)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_repeats.py -x -q > $O/t_repeats.txt 2>&1 || { tail -30 $O/t_repeats.txt; exit 1; }
tail -2 $O/t_repeats.txt
S="'' PLL_AMD_REP_HINTS=0 PLL_AMD_REP_WGS=8 PLL_AMD_REP_WGS=32 PLL_AMD_REP_WGS=64 PLL_AMD_REP_RANGES=2 PLL_AMD_REP_RANGES=8"
eval timeout -k 10 300 python tools/rep_ab.py 1000000 bench $S > $O/ab_1m.txt 2>&1; cat $O/ab_1m.txt
eval timeout -k 10 300 python tools/rep_ab.py 125000 bench $S > $O/ab_125k.txt 2>&1; cat $O/ab_125k.txt
eval timeout -k 10 300 python tools/rep_ab.py 125000 mutated $S > $O/ab_125k_mut.txt 2>&1; cat $O/ab_125k_mut.txt
eval timeout -k 10 300 python tools/rep_ab.py 1000000 mutated "''" PLL_AMD_REP_RANGES=8 > $O/ab_1m_mut.txt 2>&1; cat $O/ab_1m_mut.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 $R/tools/rep_ab.py 1000000 bench > $O/prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4s -- python3 $R/tools/rep_ab.py 125000 bench > $O/prof_c4s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rt -- python3 $R/tools/rep_ab.py 125000 mutated > $O/prof_rt.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
