set -e
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_repeats.py tests/test_gpu_c4_sharded.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r5/t2.log 2>&1 || { tail -30 gpurun_out/r5/t2.log; exit 1; }
tail -2 gpurun_out/r5/t2.log
python3 tools/rep_ab.py 125000 bench ""
python3 tools/rep_ab.py 1000000 bench ""
python tools/c4_projection.py > gpurun_out/r5/c4_proj_fast2.json 2> gpurun_out/r5/c4_proj_fast2.err
