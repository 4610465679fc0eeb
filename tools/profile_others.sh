#!/bin/bash
# rocprofv3 kernel summaries of the other BASELINE shapes (C3, C4 shard, C5); run through gpurun
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for c in c3 c4 c5; do
  rm -rf $O/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python3 bench.py --config $c --steps 10 --no-cpu > $O/prof_$c.log 2>&1
  find $O/prof_$c -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$c.csv \;
done
ls -la $O/kernel_stats_c*.csv
