#!/usr/bin/env python3
"""print the kernel sequence of two steady-state steps from a rocprofv3 kernel trace csv (launch order, grid,
VGPRs, duration, gap to the previous kernel): tools/trace_steps.py <dir-or-csv> [anchor kernel substring]"""
import csv
import glob
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_edge_"
f = path if path.endswith(".csv") else glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0, i1 = marks[len(marks) // 2], marks[len(marks) // 2 + 2]
prev = None
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:58]:58s} grid={r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:>4} vgpr={r['VGPR_Count']:>4} lds={r['LDS_Block_Size']:>6} "
          f"dur={(e - s) / 1000:7.1f}us gap={(s - prev) / 1000 if prev else 0:7.1f}us")
    prev = e
