cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c3r3; mkdir -p $O
for mode in fma mfma; do
  unset PLL_AMD_MFMA_MIN_STATES
  [ $mode = mfma ] && export PLL_AMD_MFMA_MIN_STATES=17
  echo "== $mode"
  python3 $R/bench.py --config c3r --steps 20 --no-cpu | cut -c1-130
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$mode -- python3 $R/bench.py --config c3r --steps 10 --no-cpu > $O/log.txt 2>&1
  python3 $R/tools/trace_steps.py $O/tr$mode > $O/steps$mode.txt; sed -n 1,12p $O/steps$mode.txt | cut -c1-50,60-140
done
