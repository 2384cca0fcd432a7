#!/usr/bin/env python3
"""Does a captured HIP graph shorten a StokesMatMult (9 small launches on two streams at 64^3)?
usage: graph_probe_stokes.py [P ...]   (stream launches vs graph replay, 1 and 10 callbacks per graph)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()

def timeit(fn, reps):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best

for P in [int(a) for a in sys.argv[1:]] or [64, 128]:
    op = sp.StokesOp((P, P, P))
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
    x = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    t_stream = timeit(lambda: op.mult(x, y), 200)
    res = []
    for n in (1, 10):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            op.mult(x, y)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    op.mult(x, y)
        y.zero_(); g.replay(); torch.cuda.synchronize(); chk = float(y.norm())
        res.append((n, timeit(lambda: g.replay(), 100) / n, chk))
    op.mult(x, y); torch.cuda.synchronize()
    print("stokes %d^3 MatMult: stream launches %.1f us; graph replay %s (|y| direct %.6e)" % (
        P, t_stream, ", ".join("%d per graph: %.1f us (|y| %.6e)" % r for r in res), float(y.norm())), flush=True)
    op.destroy()
