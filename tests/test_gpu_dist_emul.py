"""The 8-rank legs of BASELINE configs 3 and 5 AT THEIR SIZES, rehearsed on one GPU: G host threads of one
process, each driving its own slab handle on its own stream through the C++ hosts behind the ABI (chebhip_dist_*,
chebhip_dist_stokes_*, chebhip_dist_ell_*), with the LOCAL transport of csrc/comm.hip -- event-ordered device copies
between the ranks' buffers -- in place of RCCL.  Everything but the wire is the code an 8-GPU run executes: the
254 = 6*32 + 2*31 split, the trimmed 254 x m1 x 254 pencil plans, k_pack / k_combine with uneven column blocks, the
3-field Stokes batches.  The bar is the serial answer (any G reproduces the G = 1 vector) and the CPU oracle.
Link performance is unmeasured on hardware (no multi-GPU box)."""
import threading

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
import oracle_lib as orc
from conftest import relerr

pytestmark = pytest.mark.gpu
SEED = 20240229
POWER = (1, 1.0, 3.0, 1e-4, 1.0)   # README:52


def run_ranks(G, body):
    """body(rank, comm) on G threads, each with its own stream; returns the list of results, re-raises the first error."""
    sp = ge.load(); dsp = ge.load_dist()
    lg = dsp.LocalGroup(sp, G)
    out, err = [None] * G, [None] * G

    def worker(r):
        comm = None
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                comm = lg.comm(r)
                out[r] = body(r, comm)
                st.synchronize()
        except BaseException as e:       # noqa: a failing rank must not leave the others waiting for the time limit
            err[r] = e
            lg.abort()
        finally:
            if comm is not None and err[r] is None:
                comm.destroy()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    lg.destroy()
    for e in err:
        if e is not None:
            raise e
    return out


def poisson_ranks(dims, G, U):
    sp = ge.load(); dsp = ge.load_dist()

    def body(r, comm):
        D = dsp.DistPoissonC(dims, sp, comm=comm)
        Ul = torch.from_numpy(U[D.slab_offset:D.slab_offset + D.local_size].copy()).cuda()
        Vl = torch.full_like(Ul, float("nan"))
        D.mult(Ul, Vl)
        D.mult(Ul, Vl)                         # a second call: buffers and events are reused across exchanges
        torch.cuda.current_stream().synchronize()
        res = (D.slab_offset, Vl.cpu().numpy())
        D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    return np.concatenate([p[1] for p in parts])


@pytest.mark.parametrize("G,dims", [(2, (12, 11, 10)), (3, (13, 9)), (4, (34, 31, 18)), (8, (66, 40, 12)), (5, (20, 7, 6, 5))], ids=str)
def test_poisson_thread_ranks_small(G, dims):
    rng = np.random.default_rng(SEED)
    U = rng.standard_normal(int(np.prod([v - 2 for v in dims])))
    V = poisson_ranks(dims, G, U)
    assert relerr(V, orc.elliptic_mult(dims, U, mode=orc.DIRECT)) < 1e-10


def test_poisson_256_over_8_ranks():
    """BASELINE config 3: -dim 256,256,256 slab-split over 8 ranks (254 = 6*32 + 2*31 planes, 254 x 32|31 x 254 pencils)."""
    sp = ge.load()
    dims, G = (256, 256, 256), 8
    rng = np.random.default_rng(SEED)
    U = rng.standard_normal(254 ** 3)
    V = poisson_ranks(dims, G, U)
    ser = sp.EllipticOp(dims)
    Ud = torch.from_numpy(U).cuda(); Vs = torch.empty_like(Ud)
    ser.mult(Ud, Vs); torch.cuda.synchronize()
    ser.destroy()
    assert relerr(V, Vs.cpu().numpy()) < 1e-13           # the G = 1 vector, to the rounding of the k = 0 line products
    assert relerr(V, orc.elliptic_mult(dims, U, mode=orc.FAST, nthreads=16)) < 1e-10


def stokes_ranks(dims, G, x, dv, force, w, rheology):
    sp = ge.load(); dsp = ge.load_dist()
    d = len(dims)

    def body(r, comm):
        D = dsp.DistStokesC(dims, sp, comm=comm)
        (n0, n1), (b0, b1) = D.serial_ranges()
        D.op.set_rheology(*rheology)
        D.op.set_dirichlet(dv[b0 * d:b1 * d]); D.op.set_force(force[n0 * (d + 1):n1 * (d + 1)])
        xl = torch.from_numpy(x[n0 * (d + 1):n1 * (d + 1)].copy()).cuda()
        wl = torch.from_numpy(w[n0 * (d + 1):n1 * (d + 1)].copy()).cuda()
        yf, ym = torch.full_like(xl, float("nan")), torch.full_like(xl, float("nan"))
        D.function(xl, yf)                     # StokesFunction, then StokesMatMult linearised about its state
        D.mult(wl, ym)
        torch.cuda.current_stream().synchronize()
        res = (n0, yf.cpu().numpy(), ym.cpu().numpy())
        D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    return np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])


def stokes_inputs(dims):
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    rng = np.random.default_rng(SEED)
    return rng.standard_normal(g), rng.standard_normal(ndv), rng.standard_normal(g), rng.standard_normal(g)


# (3, (4, 6)): the last rank owns only a boundary plane -- no unknowns, empty vectors, but it takes part in the exchanges
# (6, (11, 15, 25)): the same in 3-D, where the gather is the 16-byte kernel (it read node 0 of the -- NULL -- vector of such a
# rank for its boundary nodes: a GPU fault found by tools/fuzz_dist_threads.py)
@pytest.mark.parametrize("G,dims", [(2, (10, 9, 8)), (3, (13, 12)), (3, (4, 6)), (4, (18, 17, 9)), (8, (24, 16, 10)), (6, (11, 15, 25))], ids=str)
def test_stokes_thread_ranks_small(G, dims):
    x, dv, force, w = stokes_inputs(dims)
    yf, ym = stokes_ranks(dims, G, x, dv, force, w, POWER)
    ref_f, eta, deta, strain = orc.stokes_function(dims, x, dv, force, rheology=POWER, mode=orc.DIRECT)
    ref_m = orc.stokes_mult(dims, w, eta, deta, strain, mode=orc.DIRECT)
    assert relerr(yf, ref_f) < 1e-10 and relerr(ym, ref_m) < 1e-10


def test_stokes_128_power_law_over_8_ranks():
    """BASELINE config 5: -dim 128,128,128 -rheology 1 on 8 slabs of 16 planes: StokesFunction and the Newton-linearised
    StokesMatMult against the serial handle (and through it, tests/test_gpu_stokes.py, the oracle)."""
    sp = ge.load()
    dims, G = (128, 128, 128), 8
    x, dv, force, w = stokes_inputs(dims)
    yf, ym = stokes_ranks(dims, G, x, dv, force, w, POWER)
    ser = sp.StokesOp(dims)
    ser.set_rheology(*POWER); ser.set_dirichlet(dv); ser.set_force(force)
    xs, ws = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    fs, ms = torch.empty_like(xs), torch.empty_like(xs)
    ser.function(xs, fs); ser.mult(ws, ms); torch.cuda.synchronize()
    ser.destroy()
    assert relerr(yf, fs.cpu().numpy()) < 1e-12 and relerr(ym, ms.cpu().numpy()) < 1e-12
    ref_f, eta, deta, strain = orc.stokes_function(dims, x, dv, force, rheology=POWER, mode=orc.FAST, nthreads=16)
    assert relerr(yf, ref_f) < 1e-10
    assert relerr(ym, orc.stokes_mult(dims, w, eta, deta, strain, mode=orc.FAST, nthreads=16)) < 1e-10


@pytest.mark.parametrize("dims,G", [((20, 18, 10), 4), ((5, 29), 3), ((6, 7, 5), 4)], ids=["20x18x10-4", "5x29-3", "6x7x5-4"])
def test_elliptic_general_thread_ranks(dims, G):
    """FormFunction and the Jacobian apply with variable coefficients on slabs (chebhip_dist_ell_*) vs the oracle.  In the
    second and third case the last rank owns nothing but the boundary plane: no unknowns, yet the flux eta g_0 on that plane
    is its to form (found by tools/fuzz_dist_threads.py: the Jacobian apply sent g_0 without eta from such a rank)."""
    sp = ge.load(); dsp = ge.load_dist()
    n, g, nd = orc.sizes(dims)
    rng = np.random.default_rng(SEED)
    U = rng.random(g) + 0.5; b = rng.standard_normal(g); dirv = rng.standard_normal(nd); X = rng.standard_normal(g)

    def body(r, comm):
        D = dsp.DistEllipticC(dims, sp, comm=comm)
        (n0, n1), (b0, b1) = D.serial_ranges()
        D.op.set_dirichlet(dirv[b0:b1])
        Ul, bl, Xl = (torch.from_numpy(a[n0:n1].copy()).cuda() for a in (U, b, X))
        R, V = torch.full_like(Ul, float("nan")), torch.full_like(Ul, float("nan"))
        D.function(Ul, bl, R, gamma=4.0, exponent=2.0)
        D.mult(Xl, V)
        torch.cuda.current_stream().synchronize()
        res = (n0, R.cpu().numpy(), V.cpu().numpy())
        D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    R = np.concatenate([p[1] for p in parts]); V = np.concatenate([p[2] for p in parts])
    ref_r, eta, deta, gradu = orc.elliptic_function(dims, U, b, dirv, gamma=4.0, exponent=2.0, mode=orc.DIRECT)
    assert relerr(R, ref_r) < 1e-10
    assert relerr(V, orc.elliptic_mult(dims, X, eta, deta, gradu, mode=orc.DIRECT)) < 1e-10


def test_local_reduce_and_schur_over_ranks():
    """chebhip_comm_reduce on the LOCAL transport (rank-ordered sum, same bits on every rank) inside the built-in inner
    solve of StokesMatMultSchur on 3 slabs, against the serial handle."""
    sp = ge.load(); dsp = ge.load_dist()
    dims, G = (10, 9, 8), 3
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    p = np.random.default_rng(SEED).standard_normal(gp)

    def body(r, comm):
        D = dsp.DistStokesC(dims, sp, comm=comm)
        (n0, n1), _ = D.serial_ranges()
        pl = torch.from_numpy(p[n0:n1].copy()).cuda(); sl = torch.full_like(pl, float("nan"))
        D.mult_schur(pl, sl, restart=60, rtol=1e-12, max_it=5000)
        torch.cuda.current_stream().synchronize()
        res = (n0, sl.cpu().numpy(), D.op.inner_iterations)
        D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    assert len({q[2] for q in parts}) == 1               # every rank took the same convergence decisions
    ser = sp.StokesOp(dims)
    pd = torch.from_numpy(p).cuda(); sd = torch.empty_like(pd)
    ser.mult_schur(pd, sd, restart=60, rtol=1e-12, max_it=5000); torch.cuda.synchronize()
    ser.destroy()
    assert relerr(np.concatenate([q[1] for q in parts]), sd.cpu().numpy()) < 1e-8
