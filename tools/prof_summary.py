#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / avg / min / max duration (us)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
print("%-100s %6s %10s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "%"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-100s %6d %10.1f %10.1f %10.1f %6.1f" % (k, len(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
