set -e
mkdir -p gpurun_out/r5/ttx0 gpurun_out/r5/ttx1
for i in 1 2 3; do
python bench.py --config c5 --steps 10 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('natural', d['value'], d['ms_per_step'])"
PLL_AMD_MFMA_TT_XCD=1 python bench.py --config c5 --steps 10 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('xcd    ', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/ttx0 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 10 --no-cpu > /dev/null 2>&1
export PLL_AMD_MFMA_TT_XCD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/ttx1 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 10 --no-cpu > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
grep "k_partials_mfma<16, true, true" gpurun_out/r5/ttx0/t_kernel_stats.csv | cut -c1-140
grep "k_partials_mfma<16, true, true" gpurun_out/r5/ttx1/t_kernel_stats.csv | cut -c1-140
PLL_AMD_MFMA_TT_XCD=1 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5 or 61" 2>&1 | tail -2
