set -e
python3 tools/rep_ab.py 125000 bench "" PLL_AMD_REP_FUSE=0
python3 tools/rep_ab.py 1000000 bench ""
python tools/c4_projection.py > gpurun_out/r5/c4_proj_f3.json 2> gpurun_out/r5/c4_proj_f3.err
python -m pytest tests -x -q -m gpu > gpurun_out/r5/full3.log 2>&1; tail -3 gpurun_out/r5/full3.log
