#!/usr/bin/env python3
"""Sustained-loop timing of the variable-coefficient elliptic callbacks (SURVEY 8d byte models):
FormFunction (160 B/point) and the Jacobian apply MatMult_Elliptic with eta, eta' from the last residual
(208 B/point), -gamma 4 -exponent 2 as in tests.sh:10.  usage: elliptic_bench.py [P ...] [option=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
if os.environ.get("CHEBHIP_LIB_PATH"):      # diagnostic builds (tools/f4_ablate.sh)
    sp.LIB_PATH = os.environ["CHEBHIP_LIB_PATH"]

def timeit(fn, reps):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best

for a in [a for a in sys.argv[1:] if "=" in a]:
    k, v = a.split("="); sp.set_option(k, int(v))
for P in [int(a) for a in sys.argv[1:] if "=" not in a] or [128, 256]:
    op = sp.EllipticOp((P, P, P))
    U = torch.rand(op.global_size, dtype=torch.float64, device="cuda") + 0.5     # positive state: eta = 1 + 4 u^2
    X = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    b = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    R = torch.empty_like(U); V = torch.empty_like(U)
    reps = 100 if P >= 256 else 300
    t_fn = timeit(lambda: op.function(U, b, R, gamma=4.0, exponent=2.0), reps)
    t_mm = timeit(lambda: op.mult(X, V), reps)
    n = float(P) ** 3
    print("elliptic %d^3 gamma=4: FormFunction %.1f us (%.2f TB/s of the 160 B/pt model = %.3f)  Jacobian MatMult %.1f us (%.2f TB/s of the 208 B/pt model = %.3f)" % (
        P, t_fn, 160 * n / t_fn / 1e6, 160 * n / t_fn / 1e6 / 8.0, t_mm, 208 * n / t_mm / 1e6, 208 * n / t_mm / 1e6 / 8.0))
    if P == 256:
        # PCIe-inclusive rate of the host-pointer entry point (plumbing path, never the metric): linear state
        import time, numpy as np
        lin = sp.EllipticOp((P, P, P))
        Uh = np.random.default_rng(1).standard_normal(lin.global_size)
        lin.mult_host(Uh)
        t0 = time.perf_counter()
        for _ in range(5):
            lin.mult_host(Uh)
        dt = (time.perf_counter() - t0) / 5
        print("elliptic %d^3 linear ell_op_mult_host (H2D + matvec + D2H, pageable host memory): %.2f ms = %.0f matvec/s" % (P, dt * 1e3, 1.0 / dt))
        lin.destroy()
    op.destroy()
