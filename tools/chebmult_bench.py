#!/usr/bin/env python3
"""Sustained-loop timing of single ChebMult sweeps (cheb_apply, the chebyshev.c:142-199 drop-in) per direction.
16 B/point compulsory HBM traffic, P flop/point on the f64 matrix cores.  usage: chebmult_bench.py [P ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
if os.environ.get("CHEBHIP_LIB_PATH"):      # diagnostic / A-B builds (tools/v4_overlap_ab.sh)
    sp.LIB_PATH = os.environ["CHEBHIP_LIB_PATH"]
for P in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    shape = (P, P, P)
    x = torch.randn(shape, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for tr in range(3):
        plan = sp.ChebPlan(shape, tr)
        reps = 200
        for _ in range(reps):
            plan.mult(x, y)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                plan.mult(x, y)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        print("ChebMult %d^3 tr=%d: %7.1f us  %.2f TB/s of 16 B/point (%.3f of 8 TB/s)  %.1f TF (%.3f of 78.6)" % (
            P, tr, best, 16.0 * P**3 / best / 1e6, 16.0 * P**3 / best / 1e6 / 8.0, P * P**3 / best / 1e6, P * P**3 / best / 1e6 / 78.6))
        plan.destroy()
