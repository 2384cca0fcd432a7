#!/usr/bin/env python3
"""StokesPCApply0 at 128^3 (power-law state, README inner limits) under `rocprofv3 --kernel-trace`: run 3 warm + 6 traced applies.
`pcapply_trace.py analyse <csv>`: kernel totals per apply, the time between launches (gaps) by the kernel that precedes them."""
import os, sys, csv, collections
if len(sys.argv) > 2 and sys.argv[1] == "analyse":
    rows = [r for r in csv.DictReader(open(sys.argv[2]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the traced applies: after the LAST marker kernel (randperm)
    last = max(i for i, r in enumerate(rows) if "randperm" in r["Kernel_Name"].lower() or "bitonic" in r["Kernel_Name"].lower() or "sort" in r["Kernel_Name"].lower())
    rows = rows[last + 1:]
    napply = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
    span = (t1 - t0) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
    print("%d launches over %.1f us: %.1f us per apply, kernels %.1f us per apply (serialised by the profiler), between launches %.1f us per apply" % (
        len(rows), span, span / napply, busy / napply, (span - busy) / napply))
    d = collections.defaultdict(list); g = collections.defaultdict(list)
    prev = None
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if prev is not None:
            g[prev[0]].append((int(r["Start_Timestamp"]) - prev[1]) / 1e3)
        prev = (name, int(r["End_Timestamp"]))
    print("%-72s %6s %9s %9s | gap after it: %6s %9s" % ("kernel", "n/app", "avg_us", "us/apply", "avg_us", "us/apply"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        gv = g.get(k, [0.0])
        print("%-72s %6.1f %9.1f %9.1f | %19.1f %9.1f" % (k, len(v) / napply, sum(v) / len(v), sum(v) / napply, sum(gv) / len(gv), sum(gv) / napply))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
P = 128
op = sp.StokesOp((P, P, P)); op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
c = np.cos(np.pi * np.arange(1, P - 1) / (P - 1))
X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
bub = (1 - X * X) * (1 - Y * Y) * (1 - Z * Z)
stt = np.stack([bub * np.sin(2 * Y + Z), bub * np.cos(X - 2 * Z), bub * np.sin(X + Y), 0 * bub], axis=-1).ravel()
x = torch.from_numpy(stt).cuda(); y = torch.empty_like(x)
op.function(x, y)
r = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); z = torch.empty_like(r)
M = sp.StokesSaddlePc(op, 0); M.setup()
for _ in range(3): M.apply(r, z)
torch.cuda.synchronize()
torch.randperm(1000, device="cuda"); torch.cuda.synchronize()
for _ in range(6): M.apply(r, z)
torch.cuda.synchronize()
