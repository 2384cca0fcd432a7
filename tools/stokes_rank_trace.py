#!/usr/bin/env python3
"""One rank's kernels of the slab-mode Stokes Jacobian apply at 128^3 (power law), NULL transport: a short loop for rocprofv3 --kernel-trace.
usage: stokes_rank_trace.py [G] [n] [option=value ...]   e.g. dist_packed_exchange=4 (pull form of the direct route)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load(); dsp = ge.load_dist()
for a in [a for a in sys.argv[1:] if "=" in a]:
    k, v = a.split("="); sp.set_option(k, int(v))
argv = [a for a in sys.argv if "=" not in a]
G = int(argv[1]) if len(argv) > 1 else 2
n = int(argv[2]) if len(argv) > 2 else 40
comm = dsp.Comm(sp, null=(G, 0))
D = dsp.DistStokesC((128, 128, 128), sp, comm=comm)
D.op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
D.op.set_dirichlet(np.zeros(D.dirichlet_size)); D.op.set_force(np.zeros(D.global_size))
x = torch.randn(D.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
D.function(x, y)
for _ in range(20):
    D.mult(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    D.mult(x, y)
e1.record(); torch.cuda.synchronize()
print("G = %d %s: StokesMatMult %.1f us per call" % (G, " ".join(a for a in sys.argv[1:] if "=" in a), e0.elapsed_time(e1) * 1e3 / n))
