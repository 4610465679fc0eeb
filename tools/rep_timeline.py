"""The kernels of the last complete traversal with class maps in a rocprofv3 kernel trace (start, duration, gap to the
kernel before): python tools/rep_timeline.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_rep_mark" in r["Kernel_Name"] and "k_rep_" not in rows[i - 1]["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0, prev = int(rows[a]["Start_Timestamp"]), None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  %-44s %7.1f us gap %6.1f grid %s" % ((s - t0) / 1e3, r["Kernel_Name"][:44], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, r["Grid_Size_X"]))
    prev = e
