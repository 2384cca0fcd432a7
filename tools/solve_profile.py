#!/usr/bin/env python3
"""Where the time of the config-5 solve goes (Stokes 128^3, power law, continuation 4, README inner settings; the solve of
bench.py's "solves"): wall time against the per-stage device timers of the library (chebhip_timers_*), plus the kernel
launch count.  usage: solve_profile.py [P] [lin]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
import __graft_entry__ as ge
import bench
sp = ge.load(); solve = import_module(sp.__name__ + ".solve")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
LIN = len(sys.argv) > 2 and sys.argv[2] == "lin"
c = np.cos(np.pi * np.arange(P) / (P - 1))
X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
u = np.sin(0.5 * np.pi * X) * np.cos(0.5 * np.pi * Y); v = -np.cos(0.5 * np.pi * X) * np.sin(0.5 * np.pi * Y)
val = np.stack([u, v, np.zeros_like(u), np.zeros_like(u)], axis=-1).reshape(-1, 4)
bd1 = (np.arange(P) == 0) | (np.arange(P) == P - 1)
bd = (bd1[:, None, None] | bd1[None, :, None] | bd1[None, None, :]).ravel()
U = val[~bd]; rhs = U.copy(); rhs[:, :2] *= (0.5 * np.pi) ** 2; rhs[:, 2:] = 0.0
dv = np.ascontiguousarray(val[bd][:, :3]).ravel()
rheo = (0, 1.0, 1.0, 1.0, 1.0) if LIN else (1, 1.0, 3.0, 1e-4, 1.0)
for rep in range(2):                                    # the second run is the one reported (handles, plans and rocBLAS warm)
    st = sp.StokesOp((P, P, P)); st.set_dirichlet(dv); st.set_force(rhs.ravel())
    x = torch.zeros(st.global_size, dtype=torch.float64, device="cuda")
    sp.timers(enable=True, reset=True)
    n0 = sp.lib().chebhip_launch_count() if hasattr(sp.lib(), "chebhip_launch_count") else 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stats = {}
    log = solve.stokes_solve(sp, st, x, rheology=rheo, cont0=0, cont=1 if LIN else 4, snes_rtol=1e-12 if LIN else 1e-8, ksp_rtol=1e-12 if LIN else 1e-5,
                             ksp_restart=60, ksp_max_it=200, max_linear_fail=3, snes_max_it=20, stats=stats)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t = sp.timers(enable=False)
    st.destroy()
print("wall %.3f s, %d Newton steps, %d outer Krylov its, stats %s" % (dt, sum(s[2] for s in log), sum(s[3] for s in log), stats))
tot = 0.0
for k, (ms, calls) in sorted(t.items(), key=lambda kv: -kv[1][0]):
    print("  %-28s %9.1f ms  %7d calls  %8.1f us/call" % (k, ms, calls, 1e3 * ms / max(calls, 1)))
print("(stages nest: a solve's time contains the applies inside it)")
