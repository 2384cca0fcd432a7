#!/usr/bin/env python3
"""Sustained-loop timing of the Poisson matvec: usage quick_bench.py [P] [option=value ...]
(options: chebhip_set_option names, e.g. general_kernels=1; CHEBHIP_LIB_PATH selects a diagnostic build of tools/v4_ablate.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
if os.environ.get("CHEBHIP_LIB_PATH"):      # diagnostic builds (tools/v4_ablate.sh)
    sp.LIB_PATH = os.environ["CHEBHIP_LIB_PATH"]
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for a in sys.argv[2:]:
    k, v = a.split("="); sp.set_option(k, int(v))
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
for _ in range(300):
    op.mult(U, V)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        op.mult(U, V)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / 300)
print("P=%d %s: %.1f us/matvec  %.0f matvec/s  model frac (112 B/pt) %.3f  mfma-frac %.3f" % (
    P, " ".join(sys.argv[2:]) or "shipped settings", best, 1e6 / best, 112.0 * P**3 / best / 1e6 / 8.0, 3.0 * (P - 2) * (P - 2.0)**3 / best / 1e6 / 78.6))
