#!/usr/bin/env python3
"""End-to-end solves at the BASELINE sizes (informational):
  * elliptic -dim P,P,P -exact 0 -cos_scale 3 -gamma 4: Newton + FGMRES(30) + the finite-difference preconditioner
    (elliptic.C:177-185; the reference runs ILU(2) of the same matrix), error against the analytic solution;
  * stokes -dim P,P,P -exact 2: linear solve with the README's inner settings (README:43);
  * stokes -rheology 1 -exponent 3 -eps 1e-4 -cont 4 (README:52): Newton with continuation.
usage: solve_bench.py [elliptic|stokes|power] [P ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from importlib import import_module
import __graft_entry__ as ge
import oracle_lib as orc
sp = ge.load(); solve = import_module(sp.__name__ + ".solve")
what = sys.argv[1] if len(sys.argv) > 1 else "elliptic"
sizes = [int(a) for a in sys.argv[2:]] or [64, 128]

for P in sizes:
    dims = (P, P, P)
    if what == "elliptic":
        op = sp.EllipticOp(dims)
        u, u2, dv = orc.elliptic_exact(dims, 0, gamma=4.0, exponent=2.0, cos_scale=3.0)
        op.set_dirichlet(dv)
        b = torch.from_numpy(u2).cuda(); x = torch.zeros_like(b)
        for sweeps in (0, 2):
            x.zero_()
            pc = sp.FdPc(op, sweeps=sweeps)
            hist = []
            sp.timers(enable=True, reset=True)
            torch.cuda.synchronize(); t = time.time()
            # FormJacobian runs after every FormFunction: re-assemble P in the monitor (called once per Newton step)
            its, kits, fn = solve.newton_krylov(sp, op, b, x, 4.0, 2.0, snes_rtol=1e-10, ksp_rtol=1e-6, ksp_restart=30, ksp_max_it=300, M=pc,
                                                monitor=lambda i, f, k: (pc.update(), hist.append((i, k, "%.1e" % f))))
            torch.cuda.synchronize(); dt = time.time() - t
            err = np.abs(x.cpu().numpy() - u).max() / np.abs(u).max()
            print("elliptic %d^3 gamma 4, pc sweeps %d: newton %d, ksp %d, |F| %.2e, rel.err vs analytic %.2e, %.2f s  %s" % (P, sweeps, its, kits, fn, err, dt, hist), flush=True)
            print("   timers:", {k: "%.1f ms / %d" % v for k, v in sp.timers(enable=False).items()}, flush=True)
            pc.destroy()
        op.destroy()
    else:
        st = sp.StokesOp(dims)
        U, U2, dv = orc.stokes_exact(dims, 2)
        st.set_dirichlet(dv); st.set_force(U2)
        x = torch.zeros(st.global_size, dtype=torch.float64, device="cuda")
        rheo = (0, 1.0, 1.0, 1.0, 1.0) if what == "stokes" else (1, 1.0, 3.0, 1e-4, 1.0)
        cont = 1 if what == "stokes" else 4
        hist = []
        sp.timers(enable=True, reset=True)
        torch.cuda.synchronize(); t = time.time()
        log = solve.stokes_solve(sp, st, x, rheology=rheo, cont0=0, cont=cont, snes_rtol=1e-8, ksp_rtol=1e-5 if what == "power" else 1e-10,
                                 ksp_restart=60, ksp_max_it=200, max_linear_fail=50, snes_max_it=20,
                                 monitor=lambda e, r, it, fn, k, lam: hist.append((round(e, 3), it, k, "%.1e" % fn, lam)))
        torch.cuda.synchronize(); dt = time.time() - t
        d = 3
        xs = x.cpu().numpy().reshape(-1, d + 1); Us = U.reshape(-1, d + 1)
        ev = np.abs(xs[:, :d] - Us[:, :d]).max()
        lo, hi = st.viscosity_range()
        print("%s %d^3: %.2f s, stages %s" % (what, P, dt, [(round(s[0], 3), "%.0e" % s[1], s[2], s[3], "%.1e" % s[4]) for s in log]), flush=True)
        print("   history (exponent, it, ksp, |F|, lambda):", hist, flush=True)
        print("   velocity error vs Exact2 (linear only) %.2e; viscosity range [%.3e, %.3e]" % (ev, lo, hi), flush=True)
        print("   timers:", {k: "%.1f ms / %d" % v for k, v in sp.timers(enable=False).items()}, flush=True)
        st.destroy()
