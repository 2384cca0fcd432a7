#!/usr/bin/env python3
"""A/B of the pieces of the config-5 solve in ONE process (alternating timed loops): usage solve_ab.py [P]
Stokes P^3 in a power-law state (-rheology 1 -exponent 3 -eps 1e-4, a smooth random state so that eta varies):
  MatVVPC solve          node-major with `fdm_passes` = 1 (1/eta and the modal scaling in passes of their own) / = 0 (inside the first and
                         last forward line transforms) / component-major (no (de)interleaving either)
  StokesPCApply0         `saddle_node_major` = 1 (the reference's layout inside the inner solves) / = 0 (component-major)
  MatPV, MatVV           node-major / component-major entry points
The options are read at handle creation or per call as include/chebhip.h says; every pair shares one operator handle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
op = sp.StokesOp((P, P, P)); op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
c = np.cos(np.pi * np.arange(1, P - 1) / (P - 1))
X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
bub = (1 - X * X) * (1 - Y * Y) * (1 - Z * Z)
st = np.stack([bub * np.sin(2 * Y + Z), bub * np.cos(X - 2 * Z), bub * np.sin(X + Y), 0 * bub], axis=-1).ravel()
x = torch.from_numpy(st).cuda(); y = torch.empty_like(x)
op.function(x, y)                                    # the power-law state: eta, eta', strain
print("viscosity range", op.viscosity_range())
gv, gp = op.velocity_size, op.pressure_size
v = torch.randn(gv, dtype=torch.float64, device="cuda"); w = torch.empty_like(v); q = torch.empty(gp, dtype=torch.float64, device="cuda")
r = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); z = torch.empty_like(r)


def t(fn, reps=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


pcs = {}
for passes in (1, 0):
    sp.set_option("fdm_passes", passes)
    pcs[passes] = sp.FdPc(op, sweeps=0); pcs[passes].update(); pcs[passes].apply(v, w)     # (the operand arrays are built on first use)
    sp.set_option("fdm_passes", 0)
sad = {}
for nm in (1, 0):
    sp.set_option("saddle_node_major", nm)
    sad[nm] = sp.StokesSaddlePc(op, 0); sad[nm].setup(); sad[nm].apply(r, z)
    sp.set_option("saddle_node_major", 0)
for rnd in range(3):
    sp.set_option("fdm_passes", 1); a = t(lambda: pcs[1].apply(v, w)); sp.set_option("fdm_passes", 0)
    b = t(lambda: pcs[0].apply(v, w)); cm = t(lambda: pcs[0].apply_cm(v, w))
    print("MatVVPC solve: passes %.1f us   in the transforms %.1f us   component-major %.1f us" % (a, b, cm))
    print("StokesPCApply0 (4 / 3 inner its): node-major %.1f us   component-major %.1f us" % (t(lambda: sad[1].apply(r, z), 10), t(lambda: sad[0].apply(r, z), 10)))
    print("MatVV %.1f / %.1f us   MatPV %.1f / %.1f us   MatVP %.1f / %.1f us  (node-major / component-major)" % (
        t(lambda: op.mult_vv(v, w)), t(lambda: op.mult_vv_cm(v, w)), t(lambda: op.mult_pv(v, q)), t(lambda: op.mult_pv_cm(v, q)),
        t(lambda: op.mult_vp(q, w)), t(lambda: op.mult_vp_cm(q, w))))
