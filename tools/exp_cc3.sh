cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cc3; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "cherry" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
cd /tmp
for mode in A E2 A E2; do
  unset PLL_AMD_FUSE_GENERIC
  case $mode in
    E2) export PLL_AMD_FUSE_GENERIC=2;;
  esac
  echo "== $mode"
  python3 $R/bench.py --config c3 --steps 20 --no-cpu | cut -c1-130
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$mode -- python3 $R/bench.py --config c3 --steps 10 --no-cpu > $O/log.txt 2>&1
  python3 $R/tools/trace_steps.py $O/tr$mode > $O/steps$mode.txt; sed -n 1,9p $O/steps$mode.txt | cut -c1-40,60-140
done
