#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(d.items()):
    print("%-70s %-14s n=%4d mean=%14.1f" % (k, c, len(v), sum(v) / len(v)))
