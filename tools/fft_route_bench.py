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


def cheb_fft(x, dim):
    P = x.shape[dim]; n = P - 1
    shape = [1] * x.dim(); shape[dim] = -1
    k = torch.arange(0, n + 1, dtype=torch.float64, device=x.device).view(shape)
    xe = torch.cat([x, x.flip(dim).narrow(dim, 1, n - 1)], dim)                 # even extension, length 2n
    Y = torch.fft.rfft(xe, dim=dim).real                                        # REDFT00: Y_0..Y_n            (:157)
    W = Y * k                                                                   # k * Y_k                      (:171)
    Wi = W.narrow(dim, 1, n - 1)
    sgn = torch.where(torch.arange(0, n + 1, device=x.device) % 2 == 0, 1.0, -1.0).to(torch.float64).view(shape)
    y0 = (W * k).narrow(dim, 1, n - 1).sum(dim, keepdim=True) / n + 0.5 * n * Y.narrow(dim, n, 1)                       # (:172,176)
    yn = ((W * k) * (-sgn)).narrow(dim, 1, n - 1).sum(dim, keepdim=True) / n + 0.5 * (-1.0) ** (n + 1) * n * Y.narrow(dim, n, 1)   # (:173,177)
    z = torch.zeros_like(x.narrow(dim, 0, 1))
    oe = torch.cat([z, Wi, z, -Wi.flip(dim)], dim)                              # odd extension, length 2n
    Z = -torch.fft.rfft(oe, dim=dim).imag.narrow(dim, 1, n - 1)                 # RODFT00: 2 sum W_k sin(pi j k / n)  (:181)
    j = torch.arange(1, n, dtype=torch.float64, device=x.device).view(shape)
    yi = Z / (2.0 * n * torch.sin(math.pi * j / n))                             # (:190)
    return torch.cat([y0, yi, yn], dim)


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
