#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r4_cc16_prof"; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for S in 100000 200000 400000; do for mode in fifteen seven; do
  if [ $mode = seven ]; then export PLL_AMD_FUSE_CC16=0; else export PLL_AMD_FUSE_CC16=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tr" -- python3 "$R/bench.py" --config c2 --sites $S --steps 20 --no-cpu > "$O/log.txt" 2>&1
  f=$(find "$O/tr" -name "*kernel_stats.csv" | head -1)
  echo "== $S sites, $mode: $(grep -o '"ms_per_step": [0-9.]*' "$O/log.txt" | head -1)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_" in n: print(f"   {n[:60]:60s} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
PY
  rm -rf "$O/tr"
done; done
