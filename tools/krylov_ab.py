#!/usr/bin/env python3
"""A/B of the Gram-Schmidt step of chebhip_fgmres in ONE process: option `krylov_exact_norm` = 1 (three launches: dots, update + norm,
Givens + scale) against 0 (round 5: one reduction -- dots incl. w.w with the Givens step in the last-arriving block, update and
normalisation in one pass).  The option is read per solve.  Config 4: StokesMatMultSchur with a fixed 20 inner GMRES iterations at 64^3;
config 5: StokesPCApply0 (README inner limits) on a 128^3 power-law state; the 256^3 Poisson solve to 1e-8 (30 basis vectors of 131 MB)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()


def t(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def ab(tag, fn, reps=10, unit="us", per=1.0):
    for rnd in range(3):
        out = []
        for v in (1, 0):
            sp.set_option("krylov_exact_norm", v); out.append(t(fn, reps) / per)
        sp.set_option("krylov_exact_norm", 0)
        print("%-58s three launches %.1f %s   one reduction %.1f %s" % (tag, out[0], unit, out[1], unit), flush=True)


op = sp.StokesOp((64, 64, 64))
xp = torch.randn(op.pressure_size, dtype=torch.float64, device="cuda"); yp = torch.empty_like(xp)
op.mult_schur(xp, yp, restart=30, rtol=1e-300, max_it=20)
ab("Schur apply 64^3, 20 inner its: per inner iteration", lambda: op.mult_schur(xp, yp), 10, "us", 20.0)
xv = torch.randn(op.velocity_size, dtype=torch.float64, device="cuda"); yv = torch.empty_like(xv)
print("   (StokesMatMultVV 64^3: %.1f us)" % t(lambda: op.mult_vv(xv, yv), 100, 20))
op.destroy()

P = 128
op = sp.StokesOp((P, P, P)); op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
c = np.cos(np.pi * np.arange(1, P - 1) / (P - 1))
X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
bub = (1 - X * X) * (1 - Y * Y) * (1 - Z * Z)
stt = np.stack([bub * np.sin(2 * Y + Z), bub * np.cos(X - 2 * Z), bub * np.sin(X + Y), 0 * bub], axis=-1).ravel()
x = torch.from_numpy(stt).cuda(); y = torch.empty_like(x)
op.function(x, y)
r = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); z = torch.empty_like(r)
M = sp.StokesSaddlePc(op, 0); M.setup(); M.apply(r, z)
ab("StokesPCApply0 128^3 power law (4 / 3 inner its)", lambda: M.apply(r, z), 8)
outs = []
for v in (1, 0):
    sp.set_option("krylov_exact_norm", v); M.apply(r, z); torch.cuda.synchronize(); outs.append(z.clone()); print("   inner its", M.inner_iterations)
sp.set_option("krylov_exact_norm", 0)
print("   the two applies differ by %.2e (relative)" % float((outs[0] - outs[1]).norm() / outs[0].norm()))
M.destroy(); op.destroy(); del x, y, r, z

P = 256
op = sp.EllipticOp((P, P, P)); n = op.global_size
b = torch.randn(n, dtype=torch.float64, device="cuda"); xx = torch.empty_like(b)
pc = sp.FdPc(op, sweeps=0)
ks = sp.Fgmres(n, restart=30, rtol=1e-8, max_it=200)
def solve():
    ks.solve(op, b, xx, M=pc)
solve(); print("   256^3 Poisson solve: %d iterations, reason %d" % (ks.iterations, ks.reason))
ab("256^3 Poisson FGMRES(30) + FD preconditioner to 1e-8", solve, 3, "us")
sols = []
for v in (1, 0):
    sp.set_option("krylov_exact_norm", v); solve(); torch.cuda.synchronize(); sols.append((xx.clone(), ks.iterations, ks.residual))
sp.set_option("krylov_exact_norm", 0)
print("   iterations %d / %d, residual %.3e / %.3e, solutions differ by %.2e" % (sols[0][1], sols[1][1], sols[0][2], sols[1][2], float((sols[0][0] - sols[1][0]).norm() / sols[0][0].norm())))
