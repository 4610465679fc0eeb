#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
bash tools/pmc_sets.sh r5/pmc_c3 k_partials_mfma_cc -- --config c3 > gpurun_out/r5/pmc_c3.txt 2>&1
tail -45 gpurun_out/r5/pmc_c3.txt
bash tools/pmc_sets.sh r5/pmc_c5 k_partials_mfma -- --config c5 > gpurun_out/r5/pmc_c5.txt 2>&1
tail -80 gpurun_out/r5/pmc_c5.txt
find gpurun_out/r5/pmc_c3 gpurun_out/r5/pmc_c5 -name "*.csv" -size +5M -delete
