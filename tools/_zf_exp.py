import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
op = sp.StokesOp((P, P, P)); op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
x = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
op.function(x, y)
v = x[:op.velocity_size].clone(); w = torch.empty_like(v)
def t(fn, reps=60):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rnd in range(3):
    for name, val in (("stokes_z_separate", 1), ("stokes_z_separate", 0), ("equal_shares", 1)):
        sp.set_option(name, val)
        print("%s=%d: MatVV %.1f us  MatMult %.1f us" % (name, val, t(lambda: op.mult_vv(v, w)), t(lambda: op.mult(x, y))))
        sp.set_option(name, 0)
