#!/usr/bin/env python3
"""Time the Stokes callbacks at the BASELINE configs 4 and 5 (sustained loop, HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
if os.environ.get("CHEBHIP_LIB_PATH"):      # diagnostic / A-B builds (tools/v4_overlap_ab.sh)
    sp.LIB_PATH = os.environ["CHEBHIP_LIB_PATH"]

def timeit(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for P, power in ((64, False), (128, True)):
    dims = (P, P, P)
    op = sp.StokesOp(dims)
    if power:
        op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
    x = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    v = torch.randn(op.velocity_size, dtype=torch.float64, device="cuda"); vo = torch.empty_like(v)
    p = torch.randn(op.pressure_size, dtype=torch.float64, device="cuda"); po = torch.empty_like(p)
    import numpy as np
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
    t_fn = timeit(lambda: op.function(x, y))
    t_mm = timeit(lambda: op.mult(x, y))
    t_vv = timeit(lambda: op.mult_vv(v, vo))
    t_pv = timeit(lambda: op.mult_pv(v, po))
    t_vp = timeit(lambda: op.mult_vp(p, vo))
    n = float(P) ** 3
    bm, bf = (680.0 if power else 600.0), 712.0
    print("stokes %d^3 %s: MatMult %.1f us (%.2f TB/s alg @%g B/node)  Function %.1f us (%.2f TB/s alg)  VV %.1f  PV %.1f  VP %.1f us" % (
        P, "power-law" if power else "linear", t_mm, bm * n / t_mm / 1e6, bm, t_fn, bf * n / t_fn / 1e6, t_vv, t_pv, t_vp))
    if not power:
        # config 4: one Schur apply with the built-in inner GMRES(30), bounded to 60 MatVV applies; counted per inner apply
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        op.mult_schur(p, po, restart=30, rtol=1e-5, max_it=60)
        torch.cuda.synchronize(); e0.record()
        op.mult_schur(p, po, restart=30, rtol=1e-5, max_it=60)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3
        its = op.inner_iterations
        print("stokes %d^3 Schur apply: %.0f us for VP + %d inner MatVV applies + PV = %.1f us per inner apply (MatVV alone %.1f us)" % (P, t, its, t / max(its, 1), t_vv))
    op.destroy()
