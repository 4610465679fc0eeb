cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ipw
for ipw in 1 2 4 8; do
  echo "== ipw $ipw"
  PLL_AMD_MFMA_IPW=$ipw PLL_AMD_MFMA_MIN_STATES=17 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ipw/t$ipw -- python3 $R/bench.py --config c3 --steps 10 --no-cpu > $R/gpurun_out/ipw/log$ipw.txt 2>&1
  python3 $R/tools/kstats.py $R/gpurun_out/ipw/t$ipw
  python3 $R/tools/trace_steps.py $R/gpurun_out/ipw/t$ipw | head -12 | cut -c1-40,60-140
done
echo "== fma"; python3 $R/bench.py --config c3 --steps 10 --no-cpu | cut -c1-120
