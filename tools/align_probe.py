#!/usr/bin/env python3
"""How much do rows that are not multiples of 128 B cost a sweep?  One direction of the linear matvec
(cheb_apply_lap1d) on the interior layout 254^3 against layouts whose fastest dimension is padded to 256."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()

def timeit(fn, reps=200):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best

for shape in ((254, 254, 254), (254, 254, 256), (254, 256, 256)):
    n = shape[0] * shape[1] * shape[2]
    x = torch.randn(n, dtype=torch.float64, device="cuda"); w = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for tr in range(3):
        if shape[tr] != 254:
            continue
        plan = sp.Lap1dPlan(shape, tr)
        t_store = timeit(lambda: plan.apply(x, y, None, -1.0))
        t_acc = timeit(lambda: plan.apply(x, y, w, -1.0))
        print("shape %s tr=%d: STORE %.1f us  ACC %.1f us   (per 254^3-equivalent: %.1f / %.1f)" % (
            shape, tr, t_store, t_acc, t_store * 254**3 / n, t_acc * 254**3 / n))
        plan.destroy()
