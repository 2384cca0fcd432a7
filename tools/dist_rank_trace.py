#!/usr/bin/env python3
"""One rank's kernels of the slab-partitioned Poisson matvec with a transport that moves nothing (chebhip_comm_create_null):
a short loop for rocprofv3 --kernel-trace, or timed directly.  usage: dist_rank_trace.py [G] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load(); dsp = ge.load_dist()
if os.environ.get("CHEBHIP_LIB_PATH"):      # diagnostic / A-B builds (tools/v4_overlap_ab.sh)
    sp.LIB_PATH = os.path.join(ROOT, os.environ["CHEBHIP_LIB_PATH"])
for a in [a for a in sys.argv[1:] if "=" in a]:
    k, v = a.split("="); sp.set_option(k, int(v))
sys.argv = [a for a in sys.argv if "=" not in a]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
comm = dsp.Comm(sp, null=(G, 0))
D = dsp.DistPoissonC((256, 256, 256), sp, comm=comm)
U = torch.randn(D.local_size, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
for _ in range(10):
    D.mult(U, V)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    D.mult(U, V)
e1.record(); torch.cuda.synchronize()
print("G = %d: rank 0 compute side %.1f us per matvec (slab of %d values)" % (G, e0.elapsed_time(e1) * 1e3 / n, D.local_size))
