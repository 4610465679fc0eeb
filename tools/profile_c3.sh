#!/bin/bash
# kernel-trace summary of C3 (20 states): tools/profile_c3.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-c3}
O=gpurun_out
rm -rf $O/prof_$T
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 bench.py --config c3 --steps 10 --no-cpu > $O/prof_$T.log 2>&1 || { tail -5 $O/prof_$T.log; exit 1; }
find $O/prof_$T -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$T.csv \;
head -4 $O/kernel_stats_$T.csv | cut -c1-160
tail -c 250 $O/prof_$T.log | head -c 10 > /dev/null
