#!/bin/bash
# round 3, first GPU call: GPU suite, default bench, C4 projection (balanced / equal cuts), 61-state PMC sets, shard trace
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3a"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest.log"
tail -5 "$O/pytest.log"
python3 bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.err"; echo "bench rc=$?"
python3 tools/c4_projection.py --steps 20 --cut balanced > "$O/c4_proj_balanced.json" 2> "$O/c4_proj_balanced.err"; echo "proj rc=$?"
python3 tools/c4_projection.py --steps 20 --cut equal > "$O/c4_proj_equal.json" 2> "$O/c4_proj_equal.err"
python3 tools/c4_projection.py --steps 20 --cut balanced --driver python > "$O/c4_proj_balanced_py.json" 2> /dev/null
python3 bench.py --config c5 --steps 10 --no-cpu > "$O/c5_bench.json" 2> "$O/c5_bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_shard" -- python3 "$R/tools/c4_projection.py" --shard-only 1 --steps 10 --warmup 3 --blocks 1 > "$O/trace_shard.log" 2>&1
python3 "$R/tools/trace_steps.py" "$O/trace_shard" > "$O/shard_steps.txt" 2>&1
bash "$R/tools/pmc_sets.sh" r3a/pmc_c5 k_partials_mfma -- --config c5 > "$O/pmc_c5.txt" 2>&1
rm -rf "$O/trace_shard" "$O"/pmc_c5/pmc_*/
ls "$O"
