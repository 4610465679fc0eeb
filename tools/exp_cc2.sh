cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cc2; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "cherry_groups" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
cd /tmp
for mode in A E2 E2i4 E2i16 M; do
  unset PLL_AMD_FUSE_GENERIC PLL_AMD_MFMA_MIN_STATES PLL_AMD_MFMA_IPW
  case $mode in
    E2) export PLL_AMD_FUSE_GENERIC=2;;
    E2i4) export PLL_AMD_FUSE_GENERIC=2 PLL_AMD_MFMA_IPW=4;;
    E2i16) export PLL_AMD_FUSE_GENERIC=2 PLL_AMD_MFMA_IPW=16;;
    M) export PLL_AMD_FUSE_GENERIC=1 PLL_AMD_MFMA_MIN_STATES=17;;
  esac
  echo "== $mode"
  python3 $R/bench.py --config c3 --steps 20 --no-cpu | cut -c1-130
done
export PLL_AMD_FUSE_GENERIC=2; unset PLL_AMD_MFMA_IPW PLL_AMD_MFMA_MIN_STATES
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --config c3 --steps 10 --no-cpu > $O/log.txt 2>&1
python3 $R/tools/kstats.py $O/tr
python3 $R/tools/trace_steps.py $O/tr > $O/steps.txt; sed -n 1,10p $O/steps.txt | cut -c1-40,60-140
