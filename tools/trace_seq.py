#!/usr/bin/env python3
"""Print the last `n` kernel launches of a rocprofv3 kernel-trace CSV in time order: duration and the gap before each."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-70s %8.1f us  gap %7.1f us  grid %s wg %s" % (r["Kernel_Name"][:70], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0,
                                                   r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
    prev = e
print("span of these launches: %.1f us" % ((int(rows[-1]["End_Timestamp"]) - int(rows[-n]["Start_Timestamp"])) / 1e3))
