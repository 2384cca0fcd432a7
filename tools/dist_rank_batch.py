#!/usr/bin/env python3
"""One rank's compute side of the slab-partitioned Poisson matvec (NULL transport) per vector, for nrhs vectors per exchange.
usage: dist_rank_batch.py [G] [nrhs ...] [option=value] [shadow=1]
shadow=1: the "peers'" slabs and result arrays are G - 1 arrays of their own (chebhip_comm_null_set_shadow) instead of the rank's own --
the pencil job's rows then come from (and go to) distinct memory, as among real ranks, not from the caches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load(); dsp = ge.load_dist()
SHADOW = any(a == "shadow=1" for a in sys.argv[1:])
for a in [a for a in sys.argv[1:] if "=" in a and not a.startswith("shadow=")]:
    k, v = a.split("="); sp.set_option(k, int(v))
argv = [a for a in sys.argv if "=" not in a]
G = int(argv[1]) if len(argv) > 1 else 8
comm = dsp.Comm(sp, null=(G, 0))
D = dsp.DistPoissonC((256, 256, 256), sp, comm=comm)
def t_us(fn, reps=100):
    t0 = time.perf_counter(); n = 0
    while n < 20 or time.perf_counter() - t0 < 0.03:
        fn(); n += 1
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
for nrhs in [int(a) for a in argv[2:]] or [1, 2, 4]:
    U = torch.randn((nrhs, D.local_size), dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
    if SHADOW:
        for k in (0, 1):
            comm.set_null_shadow(k, [None] + [torch.randn((nrhs, D.local_size), dtype=torch.float64, device="cuda") for _ in range(G - 1)])
    t = t_us((lambda: D.mult(U[0], V[0])) if nrhs == 1 else (lambda: D.mult_batch(U, V)))
    print("G = %d nrhs = %d %s: %.1f us per vector" % (G, nrhs, " ".join(a for a in sys.argv[1:] if "=" in a), t / nrhs))
