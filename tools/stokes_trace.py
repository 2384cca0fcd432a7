#!/usr/bin/env python3
"""Run a few StokesMatMult / StokesFunction calls for a kernel trace (rocprofv3 --kernel-trace -- python3 tools/stokes_trace.py P power)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
power = (len(sys.argv) > 2 and sys.argv[2] == "1")
op = sp.StokesOp((P, P, P))
if power:
    op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
x = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
for _ in range(5):
    op.function(x, y)
for _ in range(20):
    op.mult(x, y)
torch.cuda.synchronize()
