#!/bin/bash
# PMC traffic of the dominant kernel for another bench config: tools/pmc_config.sh c3 'k_partials_tiled<20, false, false, false>' <algorithmic bytes per launch>
set -e
CFG=$1; KERNEL=$2; ALG=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rm -rf $O/pmc_${CFG}_fetch $O/pmc_${CFG}_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${CFG}_fetch -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu > $O/pmc_${CFG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${CFG}_write -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu > $O/pmc_${CFG}_write.log 2>&1
python3 tools/pmc_traffic.py --fetch $O/pmc_${CFG}_fetch --write $O/pmc_${CFG}_write --kernel "$KERNEL" --algorithmic $ALG --out $O/traffic_$CFG.json
