set -e
python -m pytest tests/test_gpu_transfers.py tests/test_gpu_core_seam.py -x -q -m gpu 2>&1 | tail -5
