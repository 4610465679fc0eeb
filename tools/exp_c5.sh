cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c5; mkdir -p $O
for v in states clv; do
python3 $R/bench.py --config c5 --tips $v --steps 20 --no-cpu | cut -c1-130
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$v -- python3 $R/bench.py --config c5 --tips $v --steps 10 --no-cpu > $O/log.txt 2>&1
python3 $R/tools/trace_steps.py $O/tr$v > $O/steps$v.txt; sed -n 1,12p $O/steps$v.txt | cut -c1-46,60-140
done
