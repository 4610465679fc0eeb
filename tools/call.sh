set -e
python -m pytest tests/test_gpu_repeats.py tests/test_gpu_c4_sharded.py -x -q -m gpu 2>&1 | tail -2
mkdir -p gpurun_out/r5/tl1m gpurun_out/r5/tl125
python3 tools/rep_ab.py 1000000 bench "" ""
python3 tools/rep_ab.py 125000 bench "" ""
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/tl1m -o t -- python3 $GRAFT_REPO_ROOT/tools/rep_ab.py 1000000 bench "" > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/tl125 -o t -- python3 $GRAFT_REPO_ROOT/tools/rep_ab.py 125000 bench "" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/rep_timeline.py gpurun_out/r5/tl1m | head -3
python3 tools/rep_timeline.py gpurun_out/r5/tl125 | head -3
