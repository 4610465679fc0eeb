#!/bin/bash
# kernel-trace summary of the C2 shape on an irregular tree: tools/profile_tree.sh random|caterpillar
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-random}
O=gpurun_out
rm -rf $O/prof_tree_$T
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tree_$T -- python3 bench.py --tree $T --steps 20 --no-cpu > $O/prof_tree_$T.log 2>&1 || { tail -5 $O/prof_tree_$T.log; exit 1; }
find $O/prof_tree_$T -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$T.csv \;
head -8 $O/kernel_stats_$T.csv
