python -m pytest tests/test_gpu_repeats.py -x -q -m gpu -k "class_maps_match" 2>&1 | tail -12
