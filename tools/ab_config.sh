#!/bin/bash
# A/B on one box: tools/ab_config.sh <config> [bench args] -- VAR=val ...   runs bench.py for the config twice with and twice
# without the environment settings after "--", then prints the launch sequence of one step (rocprofv3 kernel trace) for both
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=$1; shift
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
[ "$1" = "--" ] && shift
SETTINGS=("$@")
O=$R/gpurun_out/ab_$CFG; mkdir -p $O
for rep in 1 2; do
  for mode in with without; do
    echo "== $CFG $mode ${SETTINGS[*]}"
    if [ $mode = with ] && [ ${#SETTINGS[@]} -gt 0 ]; then env "${SETTINGS[@]}" python3 $R/bench.py --config $CFG "${ARGS[@]}" --steps 20 --no-cpu | cut -c1-130
    else python3 $R/bench.py --config $CFG "${ARGS[@]}" --steps 20 --no-cpu | cut -c1-130; fi
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --config $CFG "${ARGS[@]}" --steps 10 --no-cpu > $O/log.txt 2>&1
python3 $R/tools/trace_steps.py $O/tr > $O/steps.txt; sed -n 1,12p $O/steps.txt | cut -c1-50,60-140
