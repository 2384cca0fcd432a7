#!/usr/bin/env python3
"""Randomised parity sweep against the CPU oracle (one-off hardening run, not part of the test-suite):
random ranks, extents, directions and coefficient states for cheb_apply, the elliptic callbacks, the Stokes
callbacks and the slab-mode drivers at one rank.  usage: python -u fuzz_parity.py [seconds] [seed]   (-u: a progress line per 500 cases must reach the log while the run lasts)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import oracle_lib as orc
sp = ge.load(); dsp = ge.load_dist()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
worst = {}
def note(kind, e, what):
    if e > worst.get(kind, (0, None))[0]:
        worst[kind] = (e, what)
    assert e < 1e-10, (kind, e, what)

def rand_dims(rank, lo, hi, cap):
    while True:
        d = tuple(int(v) for v in rng.integers(lo, hi + 1, size=rank))
        if np.prod(d) <= cap:
            return d

t0 = time.time(); n = 0
while time.time() - t0 < budget:
    kind = rng.integers(0, 5)
    if kind == 0:        # ChebMult
        rank = int(rng.integers(1, 5)); dims = rand_dims(rank, 2, 70 if rank > 1 else 300, 400000); tr = int(rng.integers(0, rank))
        if rank in (2, 3) and rng.random() < 0.15:        # a long line (> 256 points) inside a small tensor
            dl = list(rand_dims(rank, 2, 40, 3000)); dl[tr] = int(rng.integers(257, 1100)); dims = tuple(dl)      # both tilings of sweep_xl.hip, ragged tiles, and beyond 1024
        if dims[tr] < 2: continue
        x = rng.standard_normal(dims)
        plan = sp.ChebPlan(dims, tr); y = torch.empty(x.size, dtype=torch.float64, device="cuda")
        plan.mult(dev(x.ravel()), y); plan.destroy()
        note("cheb", rel(y.cpu().numpy().reshape(dims), orc.cheb_mult(x, tr, orc.FAST)), (dims, tr))
    elif kind == 1:      # elliptic, linear + nonlinear
        rank = int(rng.integers(1, 4)); dims = rand_dims(rank, 3, 48 if rank > 1 else 200, 120000)
        if rank == 2 and rng.random() < 0.1:
            dims = (int(rng.integers(257, 400)), int(rng.integers(3, 12)))[::int(rng.choice([1, -1]))]
        elif rank == 3 and rng.random() < 0.06:                          # a long direction inside a 3-D operator (plain sweeps + pointwise passes)
            dl = [int(rng.integers(3, 20)) for _ in range(3)]; dl[int(rng.integers(0, 3))] = int(rng.integers(257, 420)); dims = tuple(dl)
        elif rank in (2, 3) and rng.random() < 0.35:
            # the shapes of cheb_fused4_kernel / the padded-W path: even extents of 66..256 points, mixed KS = 16 / 32
            hi = 256 if rank == 2 else 110
            dims = tuple(int(2 * rng.integers(33, hi // 2 + 1)) for _ in range(rank))
            if rng.random() < 0.3: dims = dims[:-1] + (int(rng.choice([66, 128, 130, 254, 256])),)
        op = sp.EllipticOp(dims)
        U = rng.standard_normal(op.global_size)
        note("ell-lin", rel(op.mult_host(U), orc.elliptic_mult(dims, U, mode=orc.FAST, nthreads=8)), dims)
        u = rng.random(op.global_size) + 0.5; b = rng.standard_normal(op.global_size); dv = rng.random(op.dirichlet_size) + 0.5
        gam, ex = float(rng.random() * 3), float(rng.choice([1.0, 2.0, 2.0, 3.0, 2.5]))
        if rng.random() < 0.4: dv = np.zeros(op.dirichlet_size)          # homogeneous rows: the interior-line FormFunction path (exponent 2)
        op.set_dirichlet(dv)
        r = op.function_host(u, b, gam, ex)
        ro, eta, deta, gradu = orc.elliptic_function(dims, u, b, dv, gam, ex, mode=orc.FAST, nthreads=8)
        note("ell-fn", rel(r, ro), (dims, gam, ex))
        note("ell-jac", rel(op.mult_host(U), orc.elliptic_mult(dims, U, eta, deta, gradu, mode=orc.FAST, nthreads=8)), (dims, gam, ex))
        if rng.random() < 0.3:                                          # the stored state, whatever path produced it
            note("ell-eta", rel(op.get_state(0), eta), dims)
            k = int(rng.integers(0, rank)); note("ell-gradu", rel(op.get_state(2 + k), gradu[k]), (dims, k))
        op.destroy()
    elif kind == 2:      # Stokes
        d = int(rng.integers(2, 4)); dims = rand_dims(d, 3, 40 if d == 2 else 22, 12000)
        if rng.random() < 0.25:                                         # long lines: the KS = 16 / 32 kernels, 6-component stress storage
            dims = tuple(int(2 * rng.integers(33, 66)) for _ in range(d)) if d == 2 else tuple(int(v) for v in rng.choice([66, 68, 72, 96, 128, 130], size=3))
            if d == 3 and np.prod(dims) > 1.3e6: dims = (dims[0], 66, dims[2])
        if d == 2 and rng.random() < 0.06:                              # one long direction (> 256 points)
            dims = (int(rng.integers(257, 330)), int(rng.integers(4, 20)))[::int(rng.choice([1, -1]))]
        op = sp.StokesOp(dims)
        power = (1, 1.0, float(rng.choice([1.0, 2.0, 3.0])), 10.0 ** -float(rng.integers(1, 5)), 1.0)
        if rng.random() < 0.35: power = (0, 1.0, 1.0, 1.0, 1.0)          # linear rheology: the uniform-viscosity Jacobian (no node loop)
        x = rng.standard_normal(op.global_size); dv = rng.standard_normal(op.dirichlet_size); f = rng.standard_normal(op.global_size)
        op.set_rheology(*power); op.set_dirichlet(dv); op.set_force(f)
        y = torch.empty(op.global_size, dtype=torch.float64, device="cuda")
        op.function(dev(x), y)
        yo, eta, deta, strain = orc.stokes_function(dims, x, dv, f, rheology=power, mode=orc.FAST, nthreads=8)
        note("st-fn", rel(y.cpu().numpy(), yo), (dims, power))
        op.mult(dev(x), y)
        note("st-mult", rel(y.cpu().numpy(), orc.stokes_mult(dims, x, eta, deta, strain, mode=orc.FAST, nthreads=8)), (dims, power))
        if rng.random() < 0.3:
            j = int(rng.integers(0, d)); note("st-strain", rel(op.get_state(2 + j), strain[j]), (dims, j))
        op.destroy()
    elif kind == 3:      # slab-mode elliptic at one rank vs serial handle
        rank = int(rng.integers(2, 4)); dims = rand_dims(rank, 3, 40, 60000)
        ser = sp.EllipticOp(dims); par = (dsp.DistEllipticC if rng.random() < 0.5 else dsp.DistEllipticOp)(dims, sp)
        u = dev(rng.random(ser.global_size) + 0.5); b = dev(rng.standard_normal(ser.global_size)); dv = rng.random(ser.dirichlet_size) + 0.5
        ser.set_dirichlet(dv); par.op.set_dirichlet(dv)
        r1, r2 = torch.empty_like(u), torch.empty_like(u)
        ser.function(u, b, r1, 1.0, 2.0); par.function(u, b, r2, 1.0, 2.0)
        note("slab-ell", rel(r2.cpu().numpy(), r1.cpu().numpy()), dims)
        ser.destroy(); par.destroy()
    else:                # slab-mode Stokes at one rank vs serial handle
        d = int(rng.integers(2, 4)); dims = rand_dims(d, 3, 36 if d == 2 else 20, 9000)
        ser = sp.StokesOp(dims); par = (dsp.DistStokesC if rng.random() < 0.5 else dsp.DistStokesOp)(dims, sp)
        x = dev(rng.standard_normal(ser.global_size)); dv = rng.standard_normal(ser.dirichlet_size)
        for o in (ser, par.op):
            o.set_rheology(1, 1.0, 3.0, 1e-3, 1.0); o.set_dirichlet(dv)
        y1, y2 = torch.empty_like(x), torch.empty_like(x)
        ser.function(x, y1); par.function(x, y2)
        note("slab-st", rel(y2.cpu().numpy(), y1.cpu().numpy()), dims)
        ser.destroy(); par.destroy()
    n += 1
    if n % 500 == 0:
        print("... %d cases, %.0f s" % (n, time.time() - t0), flush=True)
print("fuzz: %d cases in %.0f s, all within tolerance; worst per kind:" % (n, time.time() - t0))
for k, (e, what) in sorted(worst.items()):
    print("   %-9s %.2e  %s" % (k, e, what))
