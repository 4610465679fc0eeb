#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_repeats.py tests/test_gpu_c4_sharded.py tests/test_gpu_fuzz.py -x -q > $O/t_gpu.txt 2>&1 || { tail -40 $O/t_gpu.txt; exit 1; }
tail -3 $O/t_gpu.txt
S="'' PLL_AMD_REP_BITS=0"
eval timeout -k 10 300 python tools/rep_ab.py 1000000 bench $S > $O/ab_1m.txt 2>&1; cat $O/ab_1m.txt
eval timeout -k 10 300 python tools/rep_ab.py 125000 bench $S > $O/ab_125k.txt 2>&1; cat $O/ab_125k.txt
eval timeout -k 10 300 python tools/rep_ab.py 125000 mutated $S > $O/ab_125k_mut.txt 2>&1; cat $O/ab_125k_mut.txt
eval timeout -k 10 300 python tools/rep_ab.py 1000000 mutated $S > $O/ab_1m_mut.txt 2>&1; cat $O/ab_1m_mut.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 $R/tools/rep_ab.py 1000000 bench > $O/prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4s -- python3 $R/tools/rep_ab.py 125000 bench > $O/prof_c4s.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
