#!/usr/bin/env python3
"""Why does the SECOND of two identical 256^3 elliptic solves take 0.51 s after bench.py's extras() have run, when the first takes
0.21 s (and 0.18 s in a fresh process)?  Replays the bench's order -- headline operator, extras, then the solve four times -- with
the library's stage timers and wall clocks around create / solve / destroy of the Krylov handle.  usage: solve_order_probe.py [noextras]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from importlib import import_module
import __graft_entry__ as ge
import bench
sp = ge.load(); solve = import_module(sp.__name__ + ".solve")
P = 256
op0 = sp.EllipticOp((P, P, P)); U = torch.randn(op0.global_size, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
for _ in range(300): op0.mult(U, V)
torch.cuda.synchronize()
if "noextras" not in sys.argv:
    t0 = time.perf_counter(); ex = bench.extras(sp, torch); print("extras done in %.1f s" % (time.perf_counter() - t0), {k: round(v, 1) for k, v in ex.items()}, flush=True)
x1 = np.cos(np.pi * np.arange(1, P - 1) / (P - 1))
f = np.ones((P - 2,) * 3)
for k in range(3):
    g = (1.0 - x1 * x1) * (1.0 + 0.3 * np.cos(2.0 * (0.7 + 0.1 * k) * x1 + 0.9))
    f = f * g.reshape([-1 if j == k else 1 for j in range(3)])
op = sp.EllipticOp((P, P, P)); op.set_dirichlet(np.zeros(op.dirichlet_size))
us = torch.from_numpy(f.ravel()).cuda(); b = torch.empty_like(us)
op.function(us, None, b, 4.0, 2.0)
x = torch.zeros_like(us)
for rep in range(4):
    x.zero_()
    free0 = torch.cuda.mem_get_info()[0]
    pc = sp.FdPc(op, sweeps=0)
    sp.timers(enable=True, reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    its, kits, fn = solve.newton_krylov(sp, op, b, x, 4.0, 2.0, snes_rtol=1e-10, ksp_rtol=1e-6, ksp_restart=30, ksp_max_it=300, M=pc, monitor=lambda i, f_, k: pc.update())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t = sp.timers(enable=False)
    pc.destroy()
    print("run %d: %.3f s wall, %d Newton, %d Krylov its; free before %.1f GB" % (rep, dt, its, kits, free0 / 2**30))
    for k, (ms, calls) in sorted(t.items(), key=lambda kv: -kv[1][0])[:6]:
        print("      %-28s %9.1f ms  %7d calls  %8.1f us/call" % (k, ms, calls, 1e3 * ms / max(calls, 1)))
    # the handle's create / destroy alone
    torch.cuda.synchronize(); t0 = time.perf_counter(); ks = sp.Fgmres(op.global_size, restart=30); torch.cuda.synchronize(); t1 = time.perf_counter(); ks.destroy(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("      Fgmres create %.1f ms, destroy %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
