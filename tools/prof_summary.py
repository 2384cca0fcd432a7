#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / avg / min / max duration (us).
usage: prof_summary.py <kernel_trace.csv> [--skip N]
--skip N drops the first N dispatches of EVERY kernel name before averaging: with N = spin-up + warm-up steps of a
bench.py run the summary covers the timed steps only (so that launches x avg <= ms_per_step of the same run)."""
import collections, csv, sys
args = sys.argv[1:]
skip = 0
if "--skip" in args:
    i = args.index("--skip"); skip = int(args[i + 1]); del args[i:i + 2]
rows = list(csv.DictReader(open(args[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if skip:
    d = {k: v[skip:] for k, v in d.items() if len(v) > skip}
    print("# first %d dispatches of every kernel dropped (spin-up + warm-up); kernels with fewer dispatches omitted" % skip)
tot = sum(sum(v) for v in d.values())
print("%-100s %6s %10s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "%"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-100s %6d %10.1f %10.1f %10.1f %6.1f" % (k, len(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
