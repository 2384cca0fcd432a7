#!/usr/bin/env python3
"""ChebMult on lines of more than 256 points: library-GEMM route vs the dense VALU kernel (usage: longline_bench.py [no_rocblas])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
NO_RB = len(sys.argv) > 1 and sys.argv[1] == 'no_rocblas'
if NO_RB:
    sp.set_option('no_rocblas', 1)
for shape in ((512, 512), (1024, 1024), (2048, 2048), (64, 512, 64)):
    x = torch.randn(shape, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for tr in range(len(shape)):
        if shape[tr] <= 256:
            continue
        plan = sp.ChebPlan(shape, tr)
        for _ in range(5):
            plan.mult(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.mult(x, y)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        n = x.numel(); P = shape[tr]
        print("ChebMult %s tr=%d (%s): %9.1f us  %.2f TF of the full 2P flop/point  %.2f TB/s of 16 B/point" % (
            shape, tr, "VALU kernel" if NO_RB else "library GEMM", us, 2.0 * P * n / us / 1e6, 16.0 * n / us / 1e6))
        plan.destroy()
