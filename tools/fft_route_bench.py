#!/usr/bin/env python3
"""What the reference's own recipe costs on this GPU when ported the obvious way: ChebMult as DCT-I -> times k ->
DST-I -> scale (chebyshev.c:157-193) with the library FFT (torch.fft = rocFFT/hipFFT on real even / odd extensions
of length 2(P-1)) and elementwise passes.  A baseline for DESIGN.md section 2, not product code: it is checked
against cheb_apply and timed beside it."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()


sys.path.insert(0, os.path.join(ROOT, "tests"))
from fft_recipe import cheb_fft


for P in [int(a) for a in sys.argv[1:]] or [128, 256]:
    shape = (P, P, P)
    x = torch.randn(shape, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for tr in range(3):
        plan = sp.ChebPlan(shape, tr)
        plan.mult(x, y)
        ref = cheb_fft(x, tr)
        err = float((ref - y).norm() / y.norm())
        def t(fn, reps=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        t_fft = t(lambda: cheb_fft(x, tr)); t_mfma = t(lambda: plan.mult(x, y), 100)
        print("ChebMult %d^3 tr=%d: library-FFT recipe %8.1f us, dense MFMA kernel %6.1f us (%.1fx); the two differ by %.1e" % (P, tr, t_fft, t_mfma, t_fft / t_mfma, err))
        plan.destroy()
