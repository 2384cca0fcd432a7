#!/usr/bin/env python3
"""ChebMult on lines of 257 .. 1024 points (and beyond): the library's own matrix-core kernel (sweep_xl.hip) against the
rocBLAS route (option long_lines_gemm).  Flop rates are quoted on the
even/odd arithmetic the own kernel does (P flop per point) AND on the dense 2 P the GEMM does, so the columns compare
wall time, not bookkeeping.  usage: longline_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()


def timed(plan, x, y, reps=20):
    for _ in range(5):
        plan.mult(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.mult(x, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ROUTES = (("own", {}), ("rocblas", {"long_lines_gemm": 1}))
print("%-18s %2s %10s %10s   %s" % ("shape", "tr", "own us", "rocblas us", "own: TF (P flop/pt), TB/s (16 B/pt), max |own - rocblas| / max|rocblas|"))
for shape in ((512, 512), (1024, 1024), (512, 4096), (4096, 512), (1024, 8192), (8192, 1024), (64, 512, 64), (300, 300, 300), (384, 384, 384),
              (512, 512, 64), (2048, 2048)):
    x = torch.randn(shape, dtype=torch.float64, device="cuda")
    for tr in range(len(shape)):
        if shape[tr] <= 256 or shape[tr] > 4096:
            continue
        us, ys = {}, {}
        for name, opts in ROUTES:
            for k, v in opts.items():
                sp.set_option(k, v)
            plan = sp.ChebPlan(shape, tr)
            y = torch.empty_like(x)
            us[name] = timed(plan, x, y)
            ys[name] = y
            plan.destroy()
            for k in opts:
                sp.set_option(k, 0)
        n = x.numel(); P = shape[tr]
        d = float((ys["own"] - ys["rocblas"]).abs().max() / ys["rocblas"].abs().max())
        print("%-18s %2d %10.1f %10.1f   %.2f TF  %.2f TB/s  %.1e" % ("x".join(map(str, shape)), tr, us["own"], us["rocblas"],
                                                                        1.0 * P * n / us["own"] / 1e6, 16.0 * n / us["own"] / 1e6, d), flush=True)
