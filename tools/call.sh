#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t_gpu.txt 2>&1 || { tail -40 $O/t_gpu.txt; exit 1; }
tail -3 $O/t_gpu.txt
timeout -k 10 600 python tools/c4_projection.py > $O/c4_projection.json 2> $O/c4_projection.err || { tail -20 $O/c4_projection.err; exit 1; }
cat $O/c4_projection.json
S="'' PLL_AMD_REP_LEVEL_SYNC=1 PLL_AMD_REP_HINTS=0"
eval timeout -k 10 300 python tools/rep_ab.py 1000000 bench $S > $O/ab_1m.txt 2>&1; cat $O/ab_1m.txt
eval timeout -k 10 300 python tools/rep_ab.py 125000 bench $S > $O/ab_125k.txt 2>&1; cat $O/ab_125k.txt
