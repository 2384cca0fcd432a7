#!/usr/bin/env python3
"""A/B of a call-time option on ONE Stokes handle (alternating timed loops): usage stokes_ab2.py <option> [P] [linear]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
optname = sys.argv[1]; P = int(sys.argv[2]) if len(sys.argv) > 2 else 128
op = sp.StokesOp((P, P, P))
if "linear" not in sys.argv:
    op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
x = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
def t(fn, reps=100):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
op.function(x, y)
for rnd in range(3):
    for v in (0, 1):
        sp.set_option(optname, v)
        print("%s=%d: MatMult %.1f us  Function %.1f us" % (optname, v, t(lambda: op.mult(x, y)), t(lambda: op.function(x, y))))
sp.set_option(optname, 0)
