cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lean3; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_repeats.py tests/test_gpu_parity.py tests/test_gpu_derivatives.py -q -x -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
cd /tmp
for mode in lean fma lean fma; do
  unset PLL_AMD_NO_LEAN
  [ $mode = fma ] && export PLL_AMD_NO_LEAN=1
  echo "== c3r $mode"
  python3 $R/bench.py --config c3r --steps 20 --no-cpu | cut -c1-130
done
unset PLL_AMD_NO_LEAN
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --config c3r --steps 10 --no-cpu > $O/log.txt 2>&1
python3 $R/tools/trace_steps.py $O/tr > $O/steps.txt; sed -n 1,8p $O/steps.txt | cut -c1-50,60-140
