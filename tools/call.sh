#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
( timeout -k 10 250 python tools/rep_stress.py 40000 300000 bench; timeout -k 10 250 python tools/rep_stress.py 30000 125000 mutated; PLL_AMD_REP_WGS=64 PLL_AMD_REP_RANGES=16 timeout -k 10 250 python tools/rep_stress.py 15000 200000 mutated ) > $O/rep_stress.txt 2>&1
cat $O/rep_stress.txt
