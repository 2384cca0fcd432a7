#!/usr/bin/env python3
"""Time the fused Poisson matvec with parts of the kernel disabled (profiling only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
L = sp.lib()
names = {0: "full", 1: "no-gload", 2: "no-store", 3: "no-gload,no-store", 4: "no-mfma", 7: "no gload/store/mfma", 8: "no-park"}
for ab in (0, 1, 2, 3, 4, 7, 8):
    L.chebhip_debug_ablate(ab)
    for _ in range(3):
        op.mult(U, V)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        op.mult(U, V)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("P=%d matvec %-22s %8.1f us  (%.1f TF mfma-equivalent, %.2f TB/s alg)" % (P, names[ab], us, 6.0 * P * P**3 / us / 1e6, 112.0 * P**3 / us / 1e6))
L.chebhip_debug_ablate(0)
