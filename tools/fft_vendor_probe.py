#!/usr/bin/env python3
"""What the vendor's tuned batched real FFT (rocFFT through torch.fft) costs for the transform lengths of the BASELINE
sizes, kernel time only: the real-even DFT of logical length 2(P-1) that REDFT00 on P points IS (chebyshev.c:127,157).
One such transform per line is a LOWER bound on what the FFT route needs per ChebMult (it needs two, plus the x k and
metric passes).  Compare with the dense MFMA kernel's whole ChebMult / Poisson launch.  usage: fft_vendor_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()


def t_us(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for P in (128, 256):
    M = 2 * (P - 1)
    nlines = P * P
    x = torch.randn(nlines, M, dtype=torch.float64, device="cuda")
    t_r = t_us(lambda: torch.fft.rfft(x, dim=1))
    X = torch.fft.rfft(x, dim=1)
    t_i = t_us(lambda: torch.fft.irfft(X, n=M, dim=1))
    # strided lines (the x / y directions of the grid): transform along dim 0 of (M, nlines)
    xs = torch.randn(M, nlines, dtype=torch.float64, device="cuda")
    t_s = t_us(lambda: torch.fft.rfft(xs, dim=0))
    pl = sp.ChebPlan((P, P, P), 2); a = torch.randn(P ** 3, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
    t_c = t_us(lambda: pl.mult(a, b))
    op = sp.EllipticOp((P, P, P)); U = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
    t_m = t_us(lambda: op.mult(U, V))
    print("P = %3d: %d lines; vendor real FFT of length %d = 2(P-1) [%s]: forward %.1f us, inverse %.1f us, forward on strided lines %.1f us"
          % (P, nlines, M, "x".join(map(str, [f for f in (2, 3, 5, 7, 17, 127) if M % f == 0])), t_r, t_i, t_s))
    print("         dense MFMA route: whole ChebMult (D x, all %d lines) %.1f us; whole Poisson matvec (3 launches of L = D D) %.1f us = %.1f us per direction"
          % (nlines, t_c, t_m, t_m / 3))
    print("         FFT route lower bound, forward + inverse transform alone (no x k / recurrence / metric passes, no strided access): %.1f us = %.2fx the dense "
          "ChebMult and %.2fx one dense Poisson direction (a fused spectral route needs forward + inverse per direction too)"
          % (t_r + t_i, (t_r + t_i) / t_c, (t_r + t_i) / (t_m / 3)))
    pl.destroy(); op.destroy()
