#!/usr/bin/env python3
"""Time single ChebMult sweeps (cheb_apply) per direction on the GPU, with optional ablation bits.
usage: python tools/sweep_bench.py [P] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shape = (P, P, P)
x = torch.randn(shape, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
L = sp.lib()
names = {0: "full", 1: "no-gload", 2: "no-store", 3: "no-gload,no-store", 4: "no-mfma", 7: "only LDS park+sync", 8: "no-park", 6: "no-store,no-mfma", 5: "no-gload,no-mfma"}
for tr in range(3):
    plan = sp.ChebPlan(shape, tr)
    for ab in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        L.chebhip_debug_ablate(ab)
        for _ in range(3):
            plan.mult(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan.mult(x, y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("P=%d tr=%d %-22s %8.1f us  (%.2f TB/s alg, %.1f TF)" % (P, tr, names[ab], us, 16.0 * P**3 / us / 1e6, P * P**3 / us / 1e6))
    L.chebhip_debug_ablate(0)
    plan.destroy()
