#!/bin/bash
# Round evidence for the default bench (C2): kernel-trace summary + the two PMC passes, run on the GPU
# box through gpurun from the repository root. Outputs land in gpurun_out/prof_*; tools/pmc_traffic.py
# turns the PMC passes into profiles/traffic_c2.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 20 --no-cpu > $O/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/prof_write.log 2>&1
find $O/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 tools/pmc_traffic.py --fetch $O/prof_fetch --write $O/prof_write --kernel 'k_partials_dna_cc<5, 5>' \
  --algorithmic 745600000 --out $O/traffic_c2.json --trim $O/pmc
tail -c 400 $O/prof_stats.log
