set -e
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu > gpurun_out/r5/full4.log 2>&1; tail -3 gpurun_out/r5/full4.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err; python3 -c "import json; d=json.loads(open('gpurun_out/r5/bench_default.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step']['frac'], d['cpu_baseline']['value'])"
