cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c5c; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests -q -x -m gpu -k "61 or codon or mfma or s61 or fullsize or derivat or asc or mixture" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
cd /tmp
for v in states clv; do
echo "== c5 $v"
python3 $R/bench.py --config c5 --tips $v --steps 20 --no-cpu | cut -c1-130
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --config c5 --tips states --steps 10 --no-cpu > $O/log.txt 2>&1
python3 $R/tools/trace_steps.py $O/tr > $O/steps.txt; sed -n 1,11p $O/steps.txt | cut -c1-50,60-140
