#!/usr/bin/env python3
"""Power-law Stokes solve with continuation (README:52) at a given size: Newton / Krylov iteration counts per stage
for a few inner-solver settings.  usage: stokes_newton_probe.py P [eps] [cont]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from importlib import import_module
import __graft_entry__ as ge
import oracle_lib as orc
sp = ge.load(); solve = import_module(sp.__name__ + ".solve")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
cont = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dims = (P, P, P)
for (vel, schur, sweeps, kind) in [((4, 1e-5), (3, 1e-5), 0, 0), ((4, 1e-5), (3, 1e-5), 4, 0), ((8, 1e-5), (6, 1e-5), 4, 0), ((8, 1e-5), (6, 1e-5), 4, 1)]:
    st = sp.StokesOp(dims)
    U, U2, dv = orc.stokes_exact(dims, 2)
    st.set_dirichlet(dv); st.set_force(U2)
    x = torch.zeros(st.global_size, dtype=torch.float64, device="cuda")
    hist = []
    t = time.time()
    try:
        log = solve.stokes_solve(sp, st, x, rheology=(1, 1.0, 3.0, eps, 1.0), cont0=0, cont=cont, saddle_type=kind, snes_rtol=1e-7, ksp_rtol=1e-5,
                                 ksp_restart=60, ksp_max_it=120, vel=vel, schur=schur, pc_sweeps=sweeps, max_linear_fail=1000, snes_max_it=25,
                                 monitor=lambda e, r, it, fn, k, lam: hist.append((it, k, "%.1e" % fn, lam)))
        torch.cuda.synchronize()
        print("P=%d vel %s schur %s sweeps %d type %d: %.1f s" % (P, vel, schur, sweeps, kind, time.time() - t))
        for s in log:
            print("   stage exponent %.3f eps %.1e: newton %d, ksp %d, |F| %.2e" % s)
    except Exception as e:
        print("P=%d vel %s schur %s sweeps %d type %d FAILED: %s" % (P, vel, schur, sweeps, kind, str(e)[:200]))
    print("   history (it, ksp its, |F|, lambda):", hist[:40], flush=True)
    st.destroy()
