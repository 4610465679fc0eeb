#!/bin/bash
# Round 6, "one ceiling, one box": tools/store_probe3.hip, the C2 and the C3 bench under rocprofv3 --kernel-trace --stats in
# ONE gpurun call; every figure a kernel duration of the same box. Writes gpurun_out/r6/store_ceiling.txt (copied to
# profiles/r6_store_ceiling.txt). Run from the repository root: bash tools/store_ceiling.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6
mkdir -p $O
rm -rf $O/sc_probe $O/sc_c2 $O/sc_c3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/store_probe3.hip -o /tmp/stp3 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sc_probe -- /tmp/stp3 > $O/sc_probe.log 2>&1 || { tail -5 $O/sc_probe.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sc_c2 -- python3 bench.py --steps 20 --no-cpu > $O/sc_c2.log 2>&1 || { tail -5 $O/sc_c2.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sc_c3 -- python3 bench.py --config c3 --steps 10 --no-cpu > $O/sc_c3.log 2>&1 || { tail -5 $O/sc_c3.log; exit 1; }
python3 - "$O" > $O/store_ceiling.txt <<'PY'
import csv, glob, sys
O = sys.argv[1]
def rows(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return list(csv.DictReader(open(f)))
print("one box, one call: kernel durations by rocprofv3 --kernel-trace --stats (avg / min / max us over the calls)")
print()
mb = 56 * 1563 * 8192 / 1e6
print("tools/store_probe3.hip - C2's store shape, %.1f MB per launch (+ 22.4 MB of scaler words in the *_scalers variants)" % mb)
for r in sorted(rows(O + "/sc_probe"), key=lambda r: r["Name"]):
    if not r["Name"].startswith("p3_"):
        continue
    avg = float(r["AverageNs"]) / 1e3
    b = mb + (22.4 if "scalers" in r["Name"] else 0.0)
    print("  %-44s calls %3s  avg %7.1f  min %7.1f  max %7.1f   %5.2f TB/s (avg)  %5.2f (min)" % (
        r["Name"].split("(")[0], r["Calls"], avg, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, b * 1e6 / (avg * 1e-6) / 1e12,
        b * 1e6 / (float(r["MinNs"]) * 1e-9) / 1e12))
for tag, title, pick in (("sc_c2", "bench.py (C2)", ("k_partials_dna_cc", "k_edge_dna_chain")), ("sc_c3", "bench.py --config c3", ("k_partials_mfma_cc", "k_partials_tiled", "k_edge_tiled"))):
    print()
    print(title)
    for r in rows(O + "/" + tag):
        if any(p in r["Name"] for p in pick):
            print("  %-70s calls %4s  avg %7.1f  min %7.1f  max %7.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cat $O/store_ceiling.txt
grep -h '"metric"' $O/sc_c2.log $O/sc_c3.log | cut -c1-300
