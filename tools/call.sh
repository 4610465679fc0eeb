set -e
mkdir -p gpurun_out/r5/tl125 gpurun_out/r5/tl1m
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/tl125 -o t -- python3 $GRAFT_REPO_ROOT/tools/rep_ab.py 125000 bench "" > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/tl1m -o t -- python3 $GRAFT_REPO_ROOT/tools/rep_ab.py 1000000 bench "" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
echo "== 1M sites (tools/rep_ab.py 1000000 bench under rocprofv3 --kernel-trace: the last traversal with class maps; bench.py's C4 alignment, unsorted)" > gpurun_out/r5/timeline.txt
python3 tools/rep_timeline.py gpurun_out/r5/tl1m >> gpurun_out/r5/timeline.txt
echo "== 125k sites (its 125k-site prefix)" >> gpurun_out/r5/timeline.txt
python3 tools/rep_timeline.py gpurun_out/r5/tl125 >> gpurun_out/r5/timeline.txt
python3 tools/rep_ab.py 1000000 bench "" PLL_AMD_REP_LEVEL_SYNC=1 PLL_AMD_REP_HINTS=0 PLL_AMD_REP_FUSE=0 PLL_AMD_SUB_PACK_ALWAYS=1 PLL_AMD_REP_BITS=0 "" > gpurun_out/r5/ab_1m.txt
python3 tools/rep_ab.py 125000 bench "" PLL_AMD_REP_LEVEL_SYNC=1 PLL_AMD_REP_HINTS=0 PLL_AMD_REP_FUSE=0 PLL_AMD_SUB_PACK_ALWAYS=1 PLL_AMD_REP_BITS=0 "" > gpurun_out/r5/ab_125k.txt
cat gpurun_out/r5/ab_1m.txt gpurun_out/r5/ab_125k.txt
