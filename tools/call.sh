set -e
python -m pytest tests/test_gpu_transfers.py tests/test_gpu_core_seam.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
python3 tools/seam_loop.py 2000 1000
