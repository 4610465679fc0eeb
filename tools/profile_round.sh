#!/bin/bash
# Round evidence (rounds 2-3), run on the GPU box through gpurun from the repository root:
#   default bench (C2): bench line with cpu_baseline, rocprofv3 kernel summary, the two PMC passes -> traffic_c2.json
#   the other shapes: bench lines + kernel summaries (c3, c3r, c4 = the whole 1M-site alignment, c5 both tip forms)
#   C4 on N GPUs: the one-GPU projection (tools/c4_projection.py)
# Outputs land in gpurun_out/round/; copy what is to be judged into profiles/ (named per round).
set -e
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/round"
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" > "$O/c2_bench.json" 2> "$O/c2_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2" -- python3 "$R/bench.py" --steps 20 --no-cpu > "$O/c2_prof.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
find "$O/prof_c2" -name "*kernel_stats.csv" -exec cp {} "$O/c2_kernel_stats.csv" \;
PLL_COMMIT="${PLL_COMMIT:-$(cat "$R/tools/.commit" 2>/dev/null)}" python3 "$R/tools/pmc_traffic.py" --fetch "$O/pmc_fetch" --write "$O/pmc_write" --kernel 'k_partials_dna_cc<5, 5>' \
  --algorithmic 745600000 --out "$O/traffic_c2.json" --trim "$O/c2_pmc" > /dev/null
# C3: HBM bytes per launch of the group kernel (the dominant launch of the 20-state step)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc3_fetch" -- python3 "$R/bench.py" --config c3 --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc3_write" -- python3 "$R/bench.py" --config c3 --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
PLL_COMMIT="${PLL_COMMIT:-$(cat "$R/tools/.commit" 2>/dev/null)}" python3 "$R/tools/pmc_traffic.py" --fetch "$O/pmc3_fetch" --write "$O/pmc3_write" --kernel 'k_partials_mfma_cc<5>' \
  --algorithmic 1548800000 --out "$O/traffic_c3.json" --trim "$O/c3_pmc" > /dev/null
# (outputs stay under gpurun_out/: copy traffic_c2.json / traffic_c3.json into profiles/ deliberately, with the commit they were measured at)
for c in c3 c3r c4 c5; do
  python3 "$R/bench.py" --config $c --steps 10 > "$O/${c}_bench.json" 2> "$O/${c}_bench.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$c" -- python3 "$R/bench.py" --config $c --steps 10 --no-cpu > /dev/null 2>&1
  find "$O/prof_$c" -name "*kernel_stats.csv" -exec cp {} "$O/${c}_kernel_stats.csv" \;
done
python3 "$R/bench.py" --config c5 --tips states --steps 10 --no-cpu > "$O/c5_codes_bench.json" 2>/dev/null
PLL_AMD_NO_TIP_CODES=1 python3 "$R/bench.py" --config c5 --steps 10 --no-cpu > "$O/c5_dense_bench.json" 2>/dev/null   # one-hot tip CLVs NOT recognised: 30 inner x inner ops
python3 "$R/tools/c4_projection.py" --steps 20 --cut equal > "$O/c4_projection.json" 2> "$O/c4_projection.err"
python3 "$R/tools/c4_projection.py" --steps 20 --cut balanced > "$O/c4_projection_balanced_cuts.json" 2>> "$O/c4_projection.err"
gcc -O2 "$R/tools/group_latency.c" -o /tmp/group_latency -ldl && for n in 2 4 8; do /tmp/group_latency "$R/libpll-2_amd/csrc/libpll_amd.so" $n 200000; done > "$O/group_latency.txt" 2>&1
# the N > 1 flow itself, two ranks on this one device (gloo control plane, shared-memory exchange)
PLL_BENCH_SAME_DEVICE=1 python3 "$R/bench.py" --gpus 2 --backend gloo --steps 10 > "$O/c4_two_ranks_one_device.json" 2> "$O/c4_two_ranks_one_device.err"
rm -rf "$O"/prof_* "$O"/pmc_fetch "$O"/pmc_write "$O"/pmc3_fetch "$O"/pmc3_write
ls -la "$O"
