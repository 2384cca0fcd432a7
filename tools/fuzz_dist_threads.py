#!/usr/bin/env python3
"""Randomised hardening run of the C++ slab drivers on thread ranks (LOCAL transport, one GPU): random rank counts and grid
shapes -- uneven splits, ranks that own a single plane, lines from 4 to 140 points -- for the linear Poisson host
(chebhip_dist_*), the Stokes host (chebhip_dist_stokes_*: power-law StokesFunction + StokesMatMult) and the general
elliptic host (chebhip_dist_ell_*), each against the SERIAL handle on the same GPU (the serial handle is what the parity
suite holds to the oracle).  Every case is printed before it runs, so a crash names its case; `dry` lists the cases of a
seed without touching the GPU.  usage: fuzz_dist_threads.py [seconds] [seed] [dry [ncases]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import test_gpu_dist_emul as em
import oracle_lib as orc
DRY = len(sys.argv) > 3 and sys.argv[3] == "dry"
NDRY = int(sys.argv[4]) if len(sys.argv) > 4 else 200
sp = ge.load(); dsp = ge.load_dist()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
worst = {}
def note(kind, e, what, bar):
    if e >= worst.get(kind, (-1, None))[0]:
        worst[kind] = (e, what)
    assert e < bar, (kind, e, what)

def dims_for(G, d, lo, hi, cap):
    while True:
        dm = [int(v) for v in rng.integers(lo, hi + 1, size=d)]
        if rng.random() < 0.3: dm[int(rng.integers(0, d))] = int(rng.choice([66, 72, 96, 130, 140]))
        dm[0] = max(dm[0], G + 2); dm[1] = max(dm[1], G + 2)
        if np.prod(dm) <= cap: return tuple(dm)

t0 = time.time(); n = 0
while (n < NDRY) if DRY else (time.time() - t0 < budget):
    kind = int(rng.integers(0, 3)); G = int(rng.integers(2, 7))
    if kind == 0:
        d = int(rng.integers(2, 4)); dims = dims_for(G, d, 5, 40, 300000)
        U = rng.standard_normal(int(np.prod([v - 2 for v in dims])))
        print("case %d: poisson G=%d dims=%s" % (n, G, dims), flush=True)
        if DRY: n += 1; continue
        V = em.poisson_ranks(dims, G, U)
        ser = sp.EllipticOp(dims); Ud = torch.from_numpy(U).cuda(); Vs = torch.empty_like(Ud); ser.mult(Ud, Vs); torch.cuda.synchronize(); ser.destroy()
        note("poisson", rel(V, Vs.cpu().numpy()), (G, dims), 1e-12)
    elif kind == 1:
        d = int(rng.integers(2, 4)); dims = dims_for(G, d, 4, 28 if d == 3 else 60, 60000)
        print("case %d: stokes G=%d dims=%s" % (n, G, dims), flush=True)
        if DRY: n += 1; continue
        x, dv, force, w = em.stokes_inputs(dims)
        yf, ym = em.stokes_ranks(dims, G, x, dv, force, w, em.POWER)
        ser = sp.StokesOp(dims); ser.set_rheology(*em.POWER); ser.set_dirichlet(dv); ser.set_force(force)
        xs, ws = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(); fs, ms = torch.empty_like(xs), torch.empty_like(xs)
        ser.function(xs, fs); ser.mult(ws, ms); torch.cuda.synchronize(); ser.destroy()
        note("stokes-fn", rel(yf, fs.cpu().numpy()), (G, dims), 1e-11); note("stokes-mult", rel(ym, ms.cpu().numpy()), (G, dims), 1e-11)
    else:
        d = int(rng.integers(2, 4)); dims = dims_for(G, d, 4, 30 if d == 3 else 70, 80000)
        _, g, nd = orc.sizes(dims)
        U = rng.random(g) + 0.5; b = rng.standard_normal(g); dirv = (rng.random(nd) + 0.5) if rng.random() < 0.7 else np.zeros(nd)      # positive: u ** 2.5 must exist
        X = rng.standard_normal(g)
        gam, ex = float(rng.random() * 3), float(rng.choice([2.0, 2.0, 3.0, 2.5]))
        print("case %d: elliptic G=%d dims=%s gamma=%.3f exponent=%.1f dirichlet %s" % (n, G, dims, gam, ex, "zero" if not dirv.any() else "nonzero"), flush=True)
        if DRY: n += 1; continue
        ser = sp.EllipticOp(dims)
        def body(r, comm):
            D = dsp.DistEllipticC(dims, sp, comm=comm)
            (n0, n1), (b0, b1) = D.serial_ranges()
            D.op.set_dirichlet(dirv[b0:b1])
            Ul, bl, Xl = (torch.from_numpy(a[n0:n1].copy()).cuda() for a in (U, b, X))
            R, V = torch.full_like(Ul, float("nan")), torch.full_like(Ul, float("nan"))
            D.function(Ul, bl, R, gamma=gam, exponent=ex); D.mult(Xl, V)
            torch.cuda.current_stream().synchronize()
            res = (n0, R.cpu().numpy(), V.cpu().numpy()); D.destroy(); return res
        def dist_side():
            parts = sorted(em.run_ranks(G, body), key=lambda t: t[0])
            return np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])
        R, V = dist_side()
        ser.set_dirichlet(dirv)
        Rs = ser.function_host(U, b, gam, ex); Vs = ser.mult_host(X)
        if not (rel(R, Rs) < 1e-11 and rel(V, Vs) < 1e-11):     # say which side is off, and whether it is again when repeated
            ref_r, eta, deta, gradu = orc.elliptic_function(dims, U, b, dirv, gamma=gam, exponent=ex, mode=orc.DIRECT)
            ref_v = orc.elliptic_mult(dims, X, eta, deta, gradu, mode=orc.DIRECT)
            print("MISMATCH: vs oracle: slabs fn %.2e jac %.2e, serial fn %.2e jac %.2e" % (rel(R, ref_r), rel(V, ref_v), rel(Rs, ref_r), rel(Vs, ref_v)))
            bad = np.flatnonzero(np.abs(R - ref_r) > 1e-9 * np.abs(ref_r).max()); bads = np.flatnonzero(np.abs(Rs - ref_r) > 1e-9 * np.abs(ref_r).max())
            print("  slabs: %d entries of fn off, first %s; serial: %d off, first %s" % (bad.size, bad[:8], bads.size, bads[:8]))
            for rep in range(3):
                R2, V2 = dist_side(); Rs2 = ser.function_host(U, b, gam, ex); Vs2 = ser.mult_host(X)
                print("  repeat %d: slabs fn %.2e jac %.2e, serial fn %.2e jac %.2e" % (rep, rel(R2, ref_r), rel(V2, ref_v), rel(Rs2, ref_r), rel(Vs2, ref_v)), flush=True)
        ser.destroy()
        note("ell-fn", rel(R, Rs), (G, dims, gam, ex), 1e-11); note("ell-jac", rel(V, Vs), (G, dims, gam, ex), 1e-11)
    n += 1
print("fuzz (thread ranks): %d cases in %.0f s, all within the bars; worst per kind:" % (n, time.time() - t0))
for k, (e, what) in sorted(worst.items()):
    print("   %-11s %.2e  %s" % (k, e, what))
