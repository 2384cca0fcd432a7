#!/usr/bin/env python3
"""Observed parity of the Stokes callbacks against the CPU oracle: rel_l2 of VV / PV / VP / MatMult (linear state) and of
StokesFunction + the Newton-linearised StokesMatMult (power law of README:52) at 64^3, 96^3, 128^3 and on 130-point lines.
The record behind the asserts of tests/test_gpu_stokes.py.  usage: stokes_parity_table.py [sizes...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import oracle_lib as orc
sp = ge.load()
SEED = 20240229
POWER = (1, 1.0, 3.0, 1e-4, 1.0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def run(fn, x, nout):
    y = torch.full((nout,), float("nan"), dtype=torch.float64, device="cuda")
    fn(dev(x), y)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def one(dims, nt=16):
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    rng = np.random.default_rng(SEED)
    v, p, x = rng.standard_normal(gv), rng.standard_normal(gp), rng.standard_normal(g)
    op = sp.StokesOp(dims)
    op.set_dirichlet(np.zeros(ndv)); op.set_force(np.zeros(g))
    r = {}
    r["VV"] = rel(run(op.mult_vv, v, gv), orc.stokes_mult_vv(dims, v, mode=orc.FAST, nthreads=nt))
    r["PV"] = rel(run(op.mult_pv, v, gp), orc.stokes_divergence(dims, v, mode=orc.FAST, nthreads=nt))
    r["VP"] = rel(run(op.mult_vp, p, gv), orc.stokes_mult_vp(dims, p, mode=orc.FAST, nthreads=nt))
    r["MatMult"] = rel(run(op.mult, x, g), orc.stokes_mult(dims, x, mode=orc.FAST, nthreads=nt))
    op.destroy()
    xs, dv, force, w = rng.standard_normal(g), rng.standard_normal(ndv), rng.standard_normal(g), rng.standard_normal(g)
    op = sp.StokesOp(dims)
    op.set_rheology(*POWER); op.set_dirichlet(dv); op.set_force(force)
    yf = run(op.function, xs, g)
    ym = run(op.mult, w, g)
    eta_h, deta_h = op.get_state(0), op.get_state(1)
    op.destroy()
    ref_f, eta, deta, strain = orc.stokes_function(dims, xs, dv, force, rheology=POWER, mode=orc.FAST, nthreads=nt)
    ref_m = orc.stokes_mult(dims, w, eta, deta, strain, mode=orc.FAST, nthreads=nt)
    r["Function(pl)"] = rel(yf, ref_f); r["MatMult(pl)"] = rel(ym, ref_m)
    r["eta"] = rel(eta_h, eta); r["deta"] = rel(deta_h, deta)
    return r


if __name__ == "__main__":
    shapes = [(64, 64, 64), (96, 96, 96), (128, 128, 128), (136, 132, 130)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    keys = None
    for dims in shapes:
        r = one(dims)
        if keys is None:
            keys = list(r)
            print("%-14s " % "dims" + " ".join("%-12s" % k for k in keys))
        print("%-14s " % "x".join(map(str, dims)) + " ".join("%-12.2e" % r[k] for k in keys), flush=True)
