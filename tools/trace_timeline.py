#!/usr/bin/env python3
"""The last n dispatches of a rocprofv3 --kernel-trace CSV in time order: name, duration, gap to the previous kernel's end, grid.
usage: trace_timeline.py <directory holding *kernel_trace.csv> <n>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-int(sys.argv[2]):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-58s dur %6.2f us  gap %6.2f us  grid %s" % (r["Kernel_Name"][:58], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev = e
