#!/usr/bin/env python3
"""Does a captured HIP graph shorten the gaps between the dependent launches of one matvec?
usage: graph_probe.py [P ...]   (prints stream-launch vs graph-replay time per matvec)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
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

for P in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    op = sp.EllipticOp((P, P, P))
    U = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
    t_stream = timeit(lambda: op.mult(U, V), 300)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        op.mult(U, V)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                op.mult(U, V)
    t_graph = timeit(lambda: g.replay(), 30) / 10
    print("P=%d: stream launches %.1f us/matvec, graph replay (10 matvecs per graph) %.1f us/matvec" % (P, t_stream, t_graph))
    op.destroy()
