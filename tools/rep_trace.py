"""Per-launch durations of the class-map kernels (k_rep_*) from a rocprofv3 kernel trace, in launch order:
python tools/rep_trace.py <dir> [max launches]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
out, prev_end = [], None
for r in rows:
    name = r["Kernel_Name"]
    if "k_rep_" not in name and not out:
        continue
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%-28s %8.1f us  gap %6.1f  grid %s wg %s lds %s" % (name[:28], (b - a) / 1e3, (a - prev_end) / 1e3 if prev_end else 0.0,
                                                                  r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?")))
    prev_end = b
    if len(out) >= lim:
        break
print("\n".join(out))
