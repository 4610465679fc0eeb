cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lean2; mkdir -p $O
for ipb in 2 4 8 16 32; do
  export PLL_AMD_LEAN_IPB=$ipb
  for cfg in c3 c3r; do
  echo "== $cfg ipb $ipb"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$cfg$ipb -- python3 $R/bench.py --config $cfg --steps 10 --no-cpu > $O/log.txt 2>&1
  python3 $R/tools/trace_steps.py $O/tr$cfg$ipb > $O/steps$cfg$ipb.txt; sed -n 2,6p $O/steps$cfg$ipb.txt | cut -c1-44,60-80,96-112
  done
done
