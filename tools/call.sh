PLL_FUZZ_SEEDS=96 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4
