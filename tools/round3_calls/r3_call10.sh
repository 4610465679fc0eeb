#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3k"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
bash "$R/tools/pmc_sets.sh" r3k/pmc_c4 k_rep_mark -- --config c4 --sites 1000000 > "$O/pmc_c4_mark.txt" 2>&1
rm -rf "$O"/pmc_c4/pmc_*/
cat "$O/pmc_c4_mark.txt" | grep -v "^set"
