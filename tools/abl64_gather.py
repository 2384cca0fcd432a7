"""64^3 linear Stokes callbacks with and without their first launch (the gather of the global vector into component fields), in the pipelined
callback: what a sweep launch that read the global vector itself could save at most (diag build: chebhip_debug_stokes_ablate)."""
import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
sp.LIB_PATH = os.path.join(ROOT, "tools", "libchebhip_diag.so")
P = 64
op = sp.StokesOp((P, P, P)); op.set_rheology(0, 1.0, 1.0, 1.0, 1.0)
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
x = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
op.function(x, y)
L = sp.lib()
def t(fn, reps=200, warm=300):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for tag, fn in (("StokesMatMult", lambda: op.mult(x, y)), ("StokesFunction", lambda: op.function(x, y))):
    L.chebhip_debug_stokes_ablate(0); full = min(t(fn) for _ in range(3))
    L.chebhip_debug_stokes_ablate(1); nog = min(t(fn) for _ in range(3))
    L.chebhip_debug_stokes_ablate(0)
    print("%s 64^3 linear: %.1f us; without the gather launch %.1f us (marginal %.1f)" % (tag, full, nog, full - nog))
