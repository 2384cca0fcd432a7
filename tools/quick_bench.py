#!/usr/bin/env python3
"""Sustained-loop timing of the Poisson matvec (and optional ablation bits): usage quick_bench.py [P] [ablate...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
if os.environ.get("CHEBHIP_LIB_PATH"):      # diagnostic builds (tools/v4_ablate.sh)
    sp.LIB_PATH = os.environ["CHEBHIP_LIB_PATH"]
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
abl = [int(a) for a in sys.argv[2:]] or [0]
variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]   # chebhip_debug_variant bits, A/B in one process
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
L = sp.lib()
for ab, var in [(a, v) for a in abl for v in variants]:
    L.chebhip_debug_variant(var)
    L.chebhip_debug_two_stage(1 if os.environ.get('TWO_STAGE') == '1' else 0)
    L.chebhip_debug_ablate(ab)
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
    print("P=%d ablate=%d variant=%s: %.1f us/matvec  %.0f matvec/s  hbm-frac(112B) %.3f  mfma-frac %.3f" % (
        P, ab, var, best, 1e6 / best, 112.0 * P**3 / best / 1e6 / 8.0, 6.0 * P * P**3 / best / 1e6 / 78.6))
L.chebhip_debug_ablate(0)
