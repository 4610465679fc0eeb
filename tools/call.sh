set -e
python3 tools/seam_loop.py 2000 1000
PLL_AMD_PINNED_STAGING=0 python3 tools/seam_loop.py 2000 1000
python3 tools/seam_loop.py 2000 1000
PLL_AMD_PINNED_STAGING=0 python3 tools/seam_loop.py 2000 1000
python -m pytest tests -x -q -m gpu > gpurun_out/r5/full2.log 2>&1; tail -3 gpurun_out/r5/full2.log
