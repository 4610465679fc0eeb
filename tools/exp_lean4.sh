cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lean4; mkdir -p $O
cd $R && timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
cd /tmp
export PLL_AMD_NO_FUSE=1
for mode in plain fma; do
  unset PLL_AMD_LEAN_PLAIN
  [ $mode = plain ] && export PLL_AMD_LEAN_PLAIN=1
  for tree in balanced random; do
  echo "== c3 levels-only $mode $tree"
  python3 $R/bench.py --config c3 --steps 20 --no-cpu --tree $tree | cut -c1-130
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$mode -- python3 $R/bench.py --config c3 --steps 10 --no-cpu > $O/log.txt 2>&1
  python3 $R/tools/trace_steps.py $O/tr$mode > $O/steps$mode.txt; sed -n 1,8p $O/steps$mode.txt | cut -c1-50,60-140
done
