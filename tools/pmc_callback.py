#!/usr/bin/env python3
"""A short loop of one operator callback for rocprofv3 counter / trace passes.
usage: pmc_callback.py <what> [P] [n]     what: stokes_lin | stokes_pl | ell_fn | ell_jac | chebmult | longline (P = line length, 257..1024)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sp = ge.load()
for a in [a for a in sys.argv[1:] if "=" in a]:
    k, v = a.split("="); sp.set_option(k, int(v))
argv = [a for a in sys.argv if "=" not in a]
what = argv[1]
P = int(argv[2]) if len(argv) > 2 else 128
n = int(argv[3]) if len(argv) > 3 else 6
dims = (P, P, P)
if what.startswith("stokes"):
    op = sp.StokesOp(dims)
    if what == "stokes_pl":
        op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
    x = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for i in range(n):
        op.function(x, y)
        op.mult(x, y)
elif what in ("ell_fn", "ell_jac"):
    op = sp.EllipticOp(dims)
    U = torch.rand(op.global_size, dtype=torch.float64, device="cuda") + 0.5
    b = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); R = torch.empty_like(U)
    for i in range(n):
        op.function(U, b, R, gamma=4.0, exponent=2.0)
        op.mult(b, R)
elif what == "longline":            # cheb_sweep_xl_kernel, both tilings: P x 8192 along dim 0, 8192 x P along dim 1
    for shape, tr in (((P, 8192), 0), ((8192, P), 1)):
        pl = sp.ChebPlan(shape, tr)
        x = torch.randn(shape, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
        for i in range(n):
            pl.mult(x, y)
elif what == "chebmult":
    for tr in range(3):
        pl = sp.ChebPlan(dims, tr)
        x = torch.randn(P ** 3, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
        for i in range(n):
            pl.mult(x, y)
torch.cuda.synchronize()
print("done", what, P, n)
