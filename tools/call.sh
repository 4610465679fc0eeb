set -e
python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 20 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('split', d['value'], d['ms_per_step'], d['roofline']['step']['frac'])"
PLL_AMD_CHAIN_SPLIT=0 python bench.py --steps 20 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plain', d['value'], d['ms_per_step'], d['roofline']['step']['frac'])"
done
python bench.py --tree random --steps 20 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('random split', d['value'], d['ms_per_step'])"
PLL_AMD_CHAIN_SPLIT=0 python bench.py --tree random --steps 20 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('random plain', d['value'], d['ms_per_step'])"
python bench.py --tree caterpillar --steps 20 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ladder split', d['value'], d['ms_per_step'])"
PLL_AMD_CHAIN_SPLIT=0 python bench.py --tree caterpillar --steps 20 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ladder plain', d['value'], d['ms_per_step'])"
