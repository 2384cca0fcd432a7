#!/usr/bin/env python3
"""Which kernels of a 128^3 power-law StokesMatMult pay for a bad placement?  Run under `rocprofv3 --kernel-trace`: handle A (made in
a fresh process) runs 30 callbacks, a marker kernel, then gigabytes are allocated and freed and handle C runs 30 callbacks.
`placement_trace.py analyse <csv>` prints the per-kernel mean durations of the two phases."""
import os, sys, csv, collections
if len(sys.argv) > 2 and sys.argv[1] == "analyse":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    phase, acc = 0, [collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(list)]
    for r in rows:
        name = r["Kernel_Name"]
        if "k_st_fill" in name and int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) >= 0 and "MARK" in os.environ.get("X", "MARK"):
            pass
        if "randperm" in name.lower() or "bitonic" in name.lower() or "sort" in name.lower():
            phase = min(phase + 1, 2) if not acc[phase] == {} else phase
            continue
        if "k_st_" in name or "cheb_" in name:
            acc[phase][name.split("(")[0][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for ph in range(3):
        if not acc[ph]:
            continue
        print("phase %d" % ph)
        tot = 0.0
        for k, v in sorted(acc[ph].items(), key=lambda kv: -sum(kv[1])):
            print("  %-90s n=%4d mean %.1f us" % (k, len(v), sum(v) / len(v))); tot += sum(v)
        print("  total %.1f us per callback (30)" % (tot / 30))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
sp.set_option("stokes_separate_allocs", 1)
P = 128
def make():
    op = sp.StokesOp((P, P, P)); op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
    return op
A = make()
x = torch.randn(A.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
A.function(x, y)
for _ in range(20): A.mult(x, y)
torch.cuda.synchronize()
torch.randperm(1000, device="cuda"); torch.cuda.synchronize()          # marker
for _ in range(30): A.mult(x, y)
torch.cuda.synchronize()
big = [torch.empty(2 * 2**30 // 8, dtype=torch.float64, device="cuda").zero_() for _ in range(6)]
del big; torch.cuda.empty_cache()
C = make(); C.function(x, y)
for _ in range(20): C.mult(x, y)
torch.cuda.synchronize()
torch.randperm(1000, device="cuda"); torch.cuda.synchronize()          # marker
for _ in range(30): C.mult(x, y)
torch.cuda.synchronize()
