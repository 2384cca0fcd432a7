#!/usr/bin/env python3
"""A/B of two Stokes handles in ONE process (alternating timed loops): usage stokes_ab.py <option> [P] [lin]
times StokesMatMult / StokesMatMultVV / StokesFunction (power law; `lin`: the linear rheology) with <option> = 0 and = 1 at handle
creation.  Only options that are read when the handle is created can be compared this way (the option is reset afterwards);
two handles with the same settings still differ by a few per cent (placement in memory)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
optname = sys.argv[1]; P = int(sys.argv[2]) if len(sys.argv) > 2 else 128
LIN = len(sys.argv) > 3 and sys.argv[3] == 'lin'
CALL_TIME = optname in ("stokes_z_separate", "stokes_pressure_sweeps")          # options read per call: one handle, the option toggled around the timed loops
ops = []
for v in (0, 1):
    sp.set_option(optname, 0 if CALL_TIME else v)
    op = sp.StokesOp((P, P, P)); op.set_rheology(*((0, 1.0, 1.0, 1.0, 1.0) if LIN else (1, 1.0, 3.0, 1e-4, 1.0)))
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
    ops.append(op)
sp.set_option(optname, 0)
x = torch.randn(ops[0].global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
def t(fn, reps=60):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for op in ops: op.function(x, y)
for rnd in range(3):
    for v, op in enumerate(ops):
        xv = x[:op.velocity_size]; yv = y[:op.velocity_size]
        if CALL_TIME: op = ops[0]; sp.set_option(optname, v)
        print("%s=%d: MatMult %.1f us  MatMultVV %.1f us  Function %.1f us" % (optname, v, t(lambda: op.mult(x, y)), t(lambda: op.mult_vv(xv, yv)), t(lambda: op.function(x, y))))
        if CALL_TIME: sp.set_option(optname, 0)
