#!/usr/bin/env python3
"""Host/GPU cost of the distributed driver's own steps (pack, sweeps, unpack, add) without any exchange:
DistPoissonOp at G = 1 on the full grid and on a grid the size of one rank's share at G = 8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load(); dsp = ge.load_dist()
for dims in ((256, 256, 256), (34, 256, 256)):
    op = dsp.DistPoissonOp(dims, backend=dsp.HipBackend(sp))
    U = op.random_input(1); V = torch.empty_like(U)
    for _ in range(50):
        op.mult(U, V)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        op.mult(U, V)
    t_host = (time.perf_counter() - t0) / 200          # enqueue time only
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        op.mult(U, V)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 200
    print("dims %s: host enqueue %.1f us per matvec, sustained %.1f us per matvec" % (dims, t_host * 1e6, t_all * 1e6))
