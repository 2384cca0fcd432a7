#!/usr/bin/env python3
"""Linear Stokes solve (-exact 2, README:43 inner settings) at a few sizes: outer FGMRES iterations, residual,
error vs the analytic solution.  usage: stokes_solve_probe.py [saddle_type]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import oracle_lib as orc
sp = ge.load()
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for dims in [(10, 9, 8), (12, 12, 12), (16, 16, 16), (20, 20, 20), (32, 32, 32), (64, 64, 64)]:
    d = len(dims)
    st = sp.StokesOp(dims)
    U, U2, dv = orc.stokes_exact(dims, 2)
    st.set_dirichlet(dv); st.set_force(U2)
    x = torch.zeros(st.global_size, dtype=torch.float64, device="cuda")
    F = torch.empty_like(x); dx = torch.empty_like(x)
    st.function(x, F); f0 = float(F.norm())
    pc = sp.StokesSaddlePc(st, kind); pc.setup()
    for rtol in (1e-6, 1e-10):
        ks = sp.Fgmres(st.global_size, restart=60, rtol=rtol, max_it=300)
        b = -F
        torch.cuda.synchronize(); t = time.time()
        ks.solve(st, b, dx, M=pc)
        torch.cuda.synchronize(); dt = time.time() - t
        xs = dx.cpu().numpy().reshape(-1, d + 1); Us = U.reshape(-1, d + 1)
        ev = np.abs(xs[:, :d] - Us[:, :d]).max(); ep = np.abs((xs[:, d] - xs[:, d].mean()) - (Us[:, d] - Us[:, d].mean())).max()
        print("%s rtol %.0e: its %3d reason %2d  |r|/|b| %.2e  err v %.2e p %.2e  %.2f s" % ("x".join(map(str, dims)), rtol, ks.iterations, ks.reason,
              ks.residual / f0, ev, ep, dt), flush=True)
        ks.destroy()
    pc.destroy(); st.destroy()
