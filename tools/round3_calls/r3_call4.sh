#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3d"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_matrix_pipe or matrix_pipe_kernels or cherry_groups or cherry_tables" > "$O/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest.log"
tail -5 "$O/pytest.log"
if grep -q "pytest rc=0" "$O/pytest.log"; then
for v in "0 0" "2 0" "4 0" "2 1" "0 0" "2 0" "4 0"; do
  set -- $v
  PLL_AMD_MFMA_WIDE=$1 PLL_AMD_MFMA_PAD=$2 python3 bench.py --config c5 --steps 10 --blocks 3 --no-cpu > "$O/c5_w$1_p$2_$RANDOM.json" 2>> "$O/c5.err"
done
python3 - <<'PY'
import json,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3d"
for f in sorted(glob.glob(O+"/c5_w*.json"), key=os.path.getmtime):
    d=json.load(open(f)); r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["avg_launch_ms"], d.get("lnl_rel_err_pinned"))
PY
PLL_AMD_MFMA_WIDE=2 bash "$R/tools/pmc_sets.sh" r3d/pmc_c5 k_partials_mfma -- --config c5 > "$O/pmc_c5_w2.txt" 2>&1
rm -rf "$O"/pmc_c5/pmc_*/
fi
