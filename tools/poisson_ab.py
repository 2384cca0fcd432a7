#!/usr/bin/env python3
"""The constant-coefficient MatMult_Elliptic: one launch of d jobs + a sum (option poisson_launches = 1) against a launch per
direction (= 2) and the default by size (= 0: below 6 M unknowns two jobs + a last direction that takes both terms as it stores),
alternating timed loops on ONE handle (the option is read per call); results compared bitwise.
usage: poisson_ab.py [P ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
def t(fn, reps=100):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for P in [int(a) for a in sys.argv[1:]] or [32, 64, 96, 128, 160, 192, 256]:
    op = sp.EllipticOp((P, P, P))
    U = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); V0, V1, V2 = torch.empty_like(U), torch.empty_like(U), torch.empty_like(U)
    res = {}
    for rnd in range(2):
        for mode, V in ((1, V1), (2, V2), (0, V0)):
            sp.set_option("poisson_launches", mode)
            res[mode] = t(lambda: op.mult(U, V), 40 if P > 160 else 100)
    sp.set_option("poisson_launches", 0)
    print("P=%3d: one launch + sum %8.1f us   launch per direction %8.1f us   default %8.1f us   same bits: %s" % (
        P, res[1], res[2], res[0], bool(torch.equal(V1, V2) and torch.equal(V1, V0))), flush=True)
    op.destroy()
