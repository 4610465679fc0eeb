cd $GRAFT_REPO_ROOT && timeout -k 10 900 python3 -m pytest tests/test_gpu_repeats.py tests/test_gpu_c4_sharded.py tests/test_gpu_sharding_product.py -q -x -m gpu 2>&1 | tail -4
bash tools/ab_config.sh c4 -- PLL_AMD_NO_FUSE_GG=1
python3 tools/c4_projection.py --steps 20 | cut -c1-700
