cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c5b; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "cherry" > $O/tests.txt 2>&1; tail -8 $O/tests.txt
cd /tmp
for mode in fused plain fused plain; do
  unset PLL_AMD_NO_FUSE
  [ $mode = plain ] && export PLL_AMD_NO_FUSE=1
  echo "== c5 codes $mode"
  python3 $R/bench.py --config c5 --tips states --steps 20 --no-cpu | cut -c1-130
done
unset PLL_AMD_NO_FUSE
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --config c5 --tips states --steps 10 --no-cpu > $O/log.txt 2>&1
python3 $R/tools/trace_steps.py $O/tr > $O/steps.txt; sed -n 1,10p $O/steps.txt | cut -c1-50,60-140
