#!/bin/bash
# scratch driver for one gpurun call (overwritten from call to call; results under gpurun_out/)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t_gpu.txt 2>&1 || { tail -40 $O/t_gpu.txt; exit 1; }
tail -3 $O/t_gpu.txt
timeout -k 10 300 python bench.py --config c4 --steps 10 > $O/c4_bench.json 2> $O/c4_bench.err || { tail -20 $O/c4_bench.err; exit 1; }
cat $O/c4_bench.json
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
