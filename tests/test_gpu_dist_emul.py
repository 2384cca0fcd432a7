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


@pytest.mark.parametrize("packed", [0, 4, 2, 1], ids=["direct_push", "direct_pull", "direct_push_after_pull_pack", "packed_exchange"])
@pytest.mark.parametrize("G,dims", [(2, (12, 11, 10)), (3, (13, 9)), (4, (34, 31, 18)), (8, (66, 40, 12)), (5, (20, 7, 6, 5)), (3, (21, 16, 11)), (8, (70, 68, 66)), (5, (130, 40, 34))], ids=str)
def test_poisson_thread_ranks_small(G, dims, packed):
    """Thread ranks on the LOCAL transport.  direct_push (default, round 6): no pack, no messages -- the launch that runs a rank's
    pencil direction reads the peers' slabs in place AND stores every output row into the result array of the rank that owns the
    plane, the final sum reads local memory (two rendezvous per matvec); direct_pull (option dist_packed_exchange = 4): the pencil
    results stay where they are computed and the final sum reads the peers'; = 2: the pencil is materialised (k_pull_pack) and its
    result pushed by a kernel of its own -- what a rank does whose geometry the gather kernels do not take, next to ranks that push
    from the sweep; packed_exchange (= 1): pack / segment exchange / combine, the sequence the RCCL transport runs.
    Even and odd trailing extents (16-byte and 8-byte runs), uneven splits, d = 2 .. 4."""
    sp = ge.load()
    rng = np.random.default_rng(SEED)
    U = rng.standard_normal(int(np.prod([v - 2 for v in dims])))
    sp.set_option("dist_packed_exchange", packed)
    try:
        V = poisson_ranks(dims, G, U)
    finally:
        sp.set_option("dist_packed_exchange", 0)
    assert relerr(V, orc.elliptic_mult(dims, U, mode=orc.DIRECT)) < 1e-10


def test_poisson_direct_pull_equals_packed_exchange_to_the_bit():
    """The LOCAL routes (push, pull, packed) run the same kernels on the same values in the same order (only where the values travel differs): with
    dist_exact_order = 1 both reproduce the serial handle's vector to the bit, at a size where 16-byte runs, uneven column blocks
    (68 = 4 x 9 + 4 x 8) and the two-job local launch all occur; several calls in a row (events and pointer tables are reused)."""
    sp = ge.load(); dsp = ge.load_dist()
    dims, G = (70, 70, 66), 8
    g = int(np.prod([v - 2 for v in dims]))
    U = np.random.default_rng(SEED + 9).standard_normal((3, g))

    def body(r, comm):
        D = dsp.DistPoissonC(dims, sp, comm=comm)
        o, n = D.slab_offset, D.local_size
        outs = []
        for packed in (0, 1, 4):
            sp_local = packed                                   # (the option is process-wide: every rank sets the same value between collectives)
            comm.allreduce_sum(0.0)                             # all ranks have finished the previous leg before the switch
            if r == 0:
                sp.set_option("dist_packed_exchange", sp_local)
            comm.allreduce_sum(0.0)
            Vs = []
            for k in range(3):
                Ul = torch.from_numpy(U[k, o:o + n].copy()).cuda(); Vl = torch.full_like(Ul, float("nan"))
                D.mult(Ul, Vl)
                Vs.append(Vl)
            torch.cuda.current_stream().synchronize()
            outs.append(torch.stack(Vs).cpu().numpy())
        D.destroy()
        return (o, outs[0], outs[1], outs[2])
    sp.set_option("dist_exact_order", 1)
    try:
        parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    finally:
        sp.set_option("dist_exact_order", 0); sp.set_option("dist_packed_exchange", 0)
    Vd = np.concatenate([p[1] for p in parts], axis=1); Vp = np.concatenate([p[2] for p in parts], axis=1); Vl = np.concatenate([p[3] for p in parts], axis=1)
    assert np.array_equal(Vd, Vp) and np.array_equal(Vd, Vl)              # push = packed = pull
    ser = sp.EllipticOp(dims)
    for k in range(3):
        Ud = torch.from_numpy(U[k]).cuda(); Vd_ = torch.empty_like(Ud)
        ser.mult(Ud, Vd_); torch.cuda.synchronize()
        assert np.array_equal(Vd[k], Vd_.cpu().numpy())
    ser.destroy()


@pytest.mark.parametrize("G,dims,nrhs", [(3, (13, 12, 10), 3), (2, (9, 8), 2), (5, (20, 18, 11), 4), (8, (70, 68, 66), 2)], ids=str)
def test_poisson_batch_thread_ranks(G, dims, nrhs):
    """chebhip_dist_mult_batch: nrhs vectors through ONE pack, ONE grouped exchange each way, one launch per direction on the stacked
    slabs / pencils -- every vector's result equals chebhip_dist_mult's on the same handle to the bit, and the oracle's serial matvec
    to 1e-10.  Uneven splits, odd extents (8-byte pack / combine), the 16-byte kernels (68 x 66 x 64 interior)."""
    sp = ge.load(); dsp = ge.load_dist()
    n, g, nd = orc.sizes(dims)
    rng = np.random.default_rng(SEED + 5)
    U = rng.standard_normal((nrhs, g))

    def body(r, comm):
        D = dsp.DistPoissonC(dims, sp, comm=comm)
        o, ln = D.slab_offset, D.local_size
        Ul = torch.from_numpy(np.ascontiguousarray(U[:, o:o + ln])).cuda()
        Vb = torch.full_like(Ul, float("nan")); V1 = torch.full_like(Ul, float("nan"))
        D.mult_batch(Ul, Vb)
        D.mult_batch(Ul, Vb)                                   # (work set reused)
        for q in range(nrhs):
            D.mult(Ul[q], V1[q])
        torch.cuda.current_stream().synchronize()
        res = (o, Vb.cpu().numpy(), V1.cpu().numpy())
        D.destroy()
        return res
    sp.set_option("dist_exact_order", 1)                         # (so that batch and single calls, which may take different launch routes, add in one order)
    try:
        parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    finally:
        sp.set_option("dist_exact_order", 0)
    Vb = np.concatenate([p[1] for p in parts], axis=1); V1 = np.concatenate([p[2] for p in parts], axis=1)
    assert np.array_equal(Vb, V1)
    for q in range(nrhs):
        assert relerr(Vb[q], orc.elliptic_mult(dims, U[q], mode=orc.FAST, nthreads=8)) < 1e-10


def test_failed_batch_work_set_is_not_kept():
    """a work set whose build fails (here: more than 2^31 values per rank, refused before any allocation) must
    not stay in the handle -- the next call with the same nrhs fails the same clean way instead of finding a half-built entry, and
    the handle keeps working for the sizes that fit."""
    import ctypes as C
    sp = ge.load(); dsp = ge.load_dist()
    dims = (256, 256, 256, 4)                                    # interior 254^3 x 2 = 32.8 M values: 64 of them exceed 2^31
    D = dsp.DistPoissonC(dims, sp)
    n = D.local_size
    U = torch.randn(2 * n, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
    L = sp.lib()
    for _ in range(2):                                           # the second call is the one that used to find the half-built entry
        rc = L.chebhip_dist_mult_batch(D._h, 64, C.c_void_p(U.data_ptr()), C.c_void_p(V.data_ptr()), sp._stream())
        assert rc != 0 and "2^31" in sp.lib().chebhip_last_error().decode()
    D.mult_batch(U.view(2, n), V.view(2, n))
    ref = torch.empty(n, dtype=torch.float64, device="cuda")
    D.mult(U[:n], ref)
    torch.cuda.synchronize()
    assert torch.equal(V[:n], ref)
    D.destroy()


def test_poisson_256_batch_of_4_over_8_ranks():
    """BASELINE config 3 at its size, four vectors per exchange over 8 thread ranks: each equals the serial handle's matvec (1e-13)."""
    sp = ge.load(); dsp = ge.load_dist()
    dims, G, nrhs = (256, 256, 256), 8, 4
    g = 254 ** 3
    U = torch.randn((nrhs, g), dtype=torch.float64, generator=torch.Generator().manual_seed(SEED + 6))

    def body(r, comm):
        D = dsp.DistPoissonC(dims, sp, comm=comm)
        o, ln = D.slab_offset, D.local_size
        Ul = U[:, o:o + ln].contiguous().cuda(); Vb = torch.empty_like(Ul)
        D.mult_batch(Ul, Vb)
        torch.cuda.current_stream().synchronize()
        res = (o, Vb.cpu())
        D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    Vb = torch.cat([p[1] for p in parts], dim=1)
    ser = sp.EllipticOp(dims)
    for q in range(nrhs):
        Ud = U[q].cuda(); Vd = torch.empty_like(Ud)
        ser.mult(Ud, Vd); torch.cuda.synchronize()
        assert relerr(Vb[q].numpy(), Vd.cpu().numpy()) < 1e-13
    ser.destroy()


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


@pytest.mark.parametrize("G", [3, 4])
def test_stokes_thread_ranks_read_the_peers_fields_in_place(G):
    """Round 6: on a direct transport the dimension-0 sweeps of the slab-mode Stokes callbacks read the peers' slab fields in place (gather
    loader, lines of more than 64 points) and the unpack reads the peers' pencil results in place.  70 x 68 x 66 over 3 (uneven: 24 / 23 / 23
    planes, 23 / 23 / 22 columns) and 4 ranks, power law: against the oracle, and against the pack / segment-exchange / unpack route
    (`dist_packed_exchange` = 1, read when the driver is created) to rounding."""
    sp = ge.load()
    dims = (70, 68, 66)
    x, dv, force, w = stokes_inputs(dims)
    yf, ym = stokes_ranks(dims, G, x, dv, force, w, POWER)
    sp.set_option("dist_packed_exchange", 1)
    try:
        yf_p, ym_p = stokes_ranks(dims, G, x, dv, force, w, POWER)
    finally:
        sp.set_option("dist_packed_exchange", 0)
    assert relerr(yf, yf_p) < 1e-13 and relerr(ym, ym_p) < 1e-13
    # ... and the pull form of the in-place route (= 4: pencil results stay where they are computed, the unpack reads the peers'; default:
    # the sweeps store each row into the owner's receive buffer, the unpack is local): the same values moved another way -- the same bits
    sp.set_option("dist_packed_exchange", 4)
    try:
        yf_l, ym_l = stokes_ranks(dims, G, x, dv, force, w, POWER)
    finally:
        sp.set_option("dist_packed_exchange", 0)
    assert np.array_equal(yf, yf_l) and np.array_equal(ym, ym_l)
    ref_f, eta, deta, strain = orc.stokes_function(dims, x, dv, force, rheology=POWER, mode=orc.FAST, nthreads=8)
    ref_m = orc.stokes_mult(dims, w, eta, deta, strain, mode=orc.FAST, nthreads=8)
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


# 70x68x66-4 / 72x40x34-3: lines of more than 64 points along dimension 0 -- on the LOCAL transport the pencil sweep reads the ranks' slab field in
# place (round 6; oracle in its FAST mode there)
@pytest.mark.parametrize("dims,G", [((20, 18, 10), 4), ((5, 29), 3), ((6, 7, 5), 4), ((70, 68, 66), 4), ((72, 40, 34), 3)], ids=["20x18x10-4", "5x29-3", "6x7x5-4", "70x68x66-4", "72x40x34-3"])
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
    mode = orc.DIRECT if n < 100000 else orc.FAST
    ref_r, eta, deta, gradu = orc.elliptic_function(dims, U, b, dirv, gamma=4.0, exponent=2.0, mode=mode)
    assert relerr(R, ref_r) < 1e-10
    assert relerr(V, orc.elliptic_mult(dims, X, eta, deta, gradu, mode=mode)) < 1e-10


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


# ---- the preconditioners on slabs (round 4): MatVVPC by fast diagonalisation with its dimension-0 transforms on pencils, the
# block preconditioners StokesPCApply0..3 with reductions over the ranks, and config 5's Newton / continuation over ranks ----
def _serial_state(sp, dims, x, dv, force, rheology):
    ser = sp.StokesOp(dims)
    ser.set_rheology(*rheology); ser.set_dirichlet(dv); ser.set_force(force)
    xs = torch.from_numpy(x).cuda(); fs = torch.empty_like(xs)
    ser.function(xs, fs); torch.cuda.synchronize()
    return ser


@pytest.mark.parametrize("G,dims", [(2, (10, 9, 8)), (3, (13, 12)), (3, (4, 6)), (4, (18, 17, 9)), (6, (11, 15, 25)), (5, (24, 20, 70)),
                                    (3, (24, 70)), (2, (40, 130)), (3, (12, 10, 100)), (4, (8, 5, 6))], ids=str)
def test_velocity_preconditioner_on_slabs(G, dims):
    """chebhip_dist_stokes_pc: z = P_1^-1 (r / eta) (MatVVPC's approximate solve, stokes.C:1160-1241 by fast diagonalisation) on the
    velocity unknowns of every slab, with a power-law viscosity, against the serial handle's.  (3, (4, 6)) and (6, (11, 15, 25)):
    a rank without unknowns still takes part in the exchanges; (5, (24, 20, 70)): 68-point lines, the 16-byte kernels.
    (3, (24, 70)), (2, (40, 130)): 2-D slabs whose last dimension takes the one-launch z solve (k_fdm_zsolve16: 66..128 interior
    points, an even number) -- the round-4 order of the forward transforms skipped dimension 0 there;
    (3, (12, 10, 100)): the same launch in 3-D."""
    sp = ge.load(); dsp = ge.load_dist()
    d = len(dims)
    x, dv, force, w = stokes_inputs(dims)
    rv = w.reshape(-1, d + 1)[:, :d].copy().ravel()

    def body(r, comm):
        D = dsp.DistStokesC(dims, sp, comm=comm)
        (n0, n1), (b0, b1) = D.serial_ranges()
        D.op.set_rheology(*POWER); D.op.set_dirichlet(dv[b0 * d:b1 * d]); D.op.set_force(force[n0 * (d + 1):n1 * (d + 1)])
        xl = torch.from_numpy(x[n0 * (d + 1):n1 * (d + 1)].copy()).cuda(); yl = torch.empty_like(xl)
        D.function(xl, yl)
        pc = D.pc(); pc.update()
        rl = torch.from_numpy(rv[n0 * d:n1 * d].copy()).cuda(); zl = torch.full_like(rl, float("nan"))
        pc.apply(rl, zl); pc.apply(rl, zl)
        torch.cuda.current_stream().synchronize()
        res = (n0, zl.cpu().numpy())
        D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    z = np.concatenate([p[1] for p in parts])
    ser = _serial_state(sp, dims, x, dv, force, POWER)
    pc = sp.FdPc(ser, sweeps=0); pc.update()
    rs = torch.from_numpy(rv).cuda(); zs = torch.empty_like(rs)
    pc.apply(rs, zs); torch.cuda.synchronize()
    pc.destroy(); ser.destroy()
    assert relerr(z, zs.cpu().numpy()) < 1e-12


@pytest.mark.parametrize("G,dims", [(3, (24, 70)), (2, (40, 130)), (4, (18, 17, 9)), (3, (5, 29)), (4, (5, 7, 6)), (4, (8, 5, 6)), (3, (12, 10, 100)), (2, (13, 12))], ids=str)
def test_elliptic_preconditioner_on_slabs(G, dims):
    """chebhip_dist_ell_pc (FormJacobian's matrix, elliptic.C:537-590, solved by fast diagonalisation on slabs): z = P_1^-1 (r / eta)
    with the variable eta a FormFunction (gamma = 4) leaves, against the serial handle's FdPc(sweeps = 0).  2-D and 3-D, last dimensions
    that take the one-launch z solve (68, 128 and 98 interior points), (3, (5, 29)) and (4, (5, 7, 6)): ranks without interior planes
    (SlabX::setup_interior with dims[0] - 2 < G) that still take part in the exchanges; (4, (8, 5, 6)): dims[1] - 2 < G, a rank
    without pencil columns."""
    sp = ge.load(); dsp = ge.load_dist()
    n, g, nd = orc.sizes(dims)
    rng = np.random.default_rng(SEED)
    U = rng.random(g) + 0.5; b = rng.standard_normal(g); dirv = rng.standard_normal(nd); r = rng.standard_normal(g)

    def body(rank, comm):
        D = dsp.DistEllipticC(dims, sp, comm=comm)
        (n0, n1), (b0, b1) = D.serial_ranges()
        D.op.set_dirichlet(dirv[b0:b1])
        Ul, bl, rl = (torch.from_numpy(a[n0:n1].copy()).cuda() for a in (U, b, r))
        R = torch.empty_like(Ul)
        D.function(Ul, bl, R, gamma=4.0, exponent=2.0)
        pc = D.pc(); pc.update()
        zl = torch.full_like(rl, float("nan"))
        pc.apply(rl, zl); pc.apply(rl, zl)
        torch.cuda.current_stream().synchronize()
        res = (n0, zl.cpu().numpy())
        D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    z = np.concatenate([p[1] for p in parts])
    ser = sp.EllipticOp(dims)
    ser.set_dirichlet(dirv)
    Us, bs = torch.from_numpy(U).cuda(), torch.from_numpy(b).cuda(); Rs = torch.empty_like(Us)
    ser.function(Us, bs, Rs, gamma=4.0, exponent=2.0)
    pc = sp.FdPc(ser, sweeps=0); pc.update()
    rs = torch.from_numpy(r).cuda(); zs = torch.empty_like(rs)
    pc.apply(rs, zs); torch.cuda.synchronize()
    pc.destroy(); ser.destroy()
    assert np.all(np.isfinite(z)) and relerr(z, zs.cpu().numpy()) < 1e-12


@pytest.mark.parametrize("G,dims,stype", [(2, (10, 9, 8), 0), (3, (13, 12), 1), (3, (4, 6), 0), (4, (18, 17, 9), 2), (6, (11, 15, 25), 3)], ids=str)
def test_saddle_preconditioner_on_slabs(G, dims, stype):
    """StokesPCApply<stype> (stokes.C:1714-1817) over slabs -- MatVVPC on slabs, the three inner Krylov solves and the removal of
    the constant pressure mode reduced over the ranks -- against the serial handle's apply, in a power-law state."""
    sp = ge.load(); dsp = ge.load_dist()
    d = len(dims)
    x, dv, force, w = stokes_inputs(dims)

    def body(r, comm):
        D = dsp.DistStokesC(dims, sp, comm=comm)
        (n0, n1), (b0, b1) = D.serial_ranges()
        D.op.set_rheology(*POWER); D.op.set_dirichlet(dv[b0 * d:b1 * d]); D.op.set_force(force[n0 * (d + 1):n1 * (d + 1)])
        xl = torch.from_numpy(x[n0 * (d + 1):n1 * (d + 1)].copy()).cuda(); yl = torch.empty_like(xl)
        D.function(xl, yl)
        M = D.saddle(stype); M.setup()
        wl = torch.from_numpy(w[n0 * (d + 1):n1 * (d + 1)].copy()).cuda(); zl = torch.full_like(wl, float("nan"))
        M.apply(wl, zl)
        torch.cuda.current_stream().synchronize()
        res = (n0, zl.cpu().numpy(), M.inner_iterations)
        M.destroy(); D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    assert len({q[2] for q in parts}) == 1               # every rank took the same convergence decisions
    z = np.concatenate([p[1] for p in parts])
    ser = _serial_state(sp, dims, x, dv, force, POWER)
    M = sp.StokesSaddlePc(ser, stype); M.setup()
    ws = torch.from_numpy(w).cuda(); zs = torch.empty_like(ws)
    M.apply(ws, zs); torch.cuda.synchronize()
    its = M.inner_iterations
    M.destroy(); ser.destroy()
    assert parts[0][2] == its
    assert relerr(z, zs.cpu().numpy()) < 1e-8


def _exact2(dims):
    """The reference's manufactured problem -exact 2 (stokes.C:1963-2012): fields, force (linear), Dirichlet values."""
    d = len(dims)
    c = [np.cos(np.pi * np.arange(P) / (P - 1)) for P in dims]
    X = np.meshgrid(*c, indexing="ij")
    u = np.sin(0.5 * np.pi * X[0]) * np.cos(0.5 * np.pi * X[1]); v = -np.cos(0.5 * np.pi * X[0]) * np.sin(0.5 * np.pi * X[1])
    comps = [u, v] + [np.zeros_like(u)] * (d - 2) + [np.zeros_like(u)]
    val = np.stack(comps, axis=-1).reshape(-1, d + 1)
    bd = np.zeros(dims, dtype=bool)
    for ax, P in enumerate(dims):
        sl = [slice(None)] * d
        sl[ax] = 0; bd[tuple(sl)] = True
        sl[ax] = P - 1; bd[tuple(sl)] = True
    bd = bd.ravel()
    U = val[~bd]
    rhs = U.copy(); rhs[:, :2] *= (0.5 * np.pi) ** 2; rhs[:, 2:] = 0.0
    return U.ravel(), rhs.ravel(), np.ascontiguousarray(val[bd][:, :d]).ravel()


def _continuation(dims, G, cont=3, rheology=(1, 1.0, 3.0, 1e-3, 1.0), **kw):
    """config 5's solve phase (stokes.C:213-235: continuation in exponent / regularisation, Newton, FGMRES + StokesPCApply0) on
    one GPU (G = 1) or over G thread ranks; returns (solution, log)."""
    import importlib
    sp = ge.load(); dsp = ge.load_dist()
    solve = importlib.import_module(sp.__name__ + ".solve")
    d = len(dims)
    U, F, dv = _exact2(dims)
    args = dict(rheology=rheology, cont0=0, cont=cont, snes_rtol=1e-8, ksp_rtol=1e-5, ksp_restart=60, ksp_max_it=200, max_linear_fail=3, snes_max_it=20)
    args.update(kw)
    if G == 1:
        st = sp.StokesOp(dims); st.set_dirichlet(dv); st.set_force(F)
        x = torch.zeros(st.global_size, dtype=torch.float64, device="cuda")
        log = solve.stokes_solve(sp, st, x, **args)
        torch.cuda.synchronize()
        st.destroy()
        return x.cpu().numpy(), log

    def body(r, comm):
        D = dsp.DistStokesC(dims, sp, comm=comm)
        (n0, n1), (b0, b1) = D.serial_ranges()
        D.op.set_dirichlet(dv[b0 * d:b1 * d]); D.op.set_force(F[n0 * (d + 1):n1 * (d + 1)])
        x = torch.zeros(D.global_size, dtype=torch.float64, device="cuda")
        log = solve.stokes_solve(sp, D.op, x, dist=D, **args)
        torch.cuda.current_stream().synchronize()
        res = (n0, x.cpu().numpy(), log)
        D.destroy()
        return res
    parts = sorted(run_ranks(G, body), key=lambda t: t[0])
    for q in parts[1:]:
        assert [s[2:4] for s in q[2]] == [s[2:4] for s in parts[0][2]]        # same Newton / Krylov counts on every rank
    return np.concatenate([q[1] for q in parts]), parts[0][2]


LINEAR = (0, 1.0, 1.0, 1.0, 1.0)
PL2 = (1, 1.0, 3.0, 1e-2, 1.0)


@pytest.mark.parametrize("G,dims,rheology", [(2, (16, 16, 16), PL2), (5, (16, 16, 16), PL2), (3, (24, 24), PL2), (8, (9, 16, 16), LINEAR)],
                         ids=lambda v: str(v) if not isinstance(v, tuple) or len(v) < 5 else ("linear" if v[0] == 0 else "power"))
def test_continuation_over_thread_ranks(G, dims, rheology):
    """Newton / continuation (./stokes -exact 2 -rheology 1 -exponent 3 -eps 1e-2 -cont 2, README:52) with the block preconditioner
    over G ranks reproduces the one-GPU solve: the same stages and Newton steps, Krylov iteration counts within a step or two
    (the reductions add in a different order), the same solution (1e-8).  (8, (9, 16, 16)): the last rank owns only the
    boundary plane -- no unknowns, yet it takes part in every collective (linear rheology: the power-law Newton iteration does
    not converge on a grid that coarse on one GPU either)."""
    kw = dict(cont=2 if rheology[0] else 1, rheology=rheology)
    if not rheology[0]:
        kw.update(ksp_rtol=1e-9, snes_rtol=1e-7, ksp_max_it=400)            # one linear solve (a second Newton step at most)
    xs, logs = _continuation(dims, 1, **kw)
    xd, logd = _continuation(dims, G, **kw)
    assert [s[2] for s in logd] == [s[2] for s in logs], (logd, logs)         # Newton steps per stage
    assert all(abs(a[3] - b[3]) <= max(2, 0.15 * b[3]) for a, b in zip(logd, logs)), (logd, logs)   # Krylov its per stage
    assert relerr(xd, xs) < (1e-8 if rheology[0] else 1e-6)                   # (linear: two solves to ksp_rtol 1e-9 of an ill-conditioned system)


def test_config5_continuation_128_over_8_ranks():
    """BASELINE config 5 end to end over ranks: -dim 128,128,128 -rheology 1 -exponent 3 -eps 1e-4 -cont 4 (README:52) on 8 slabs of
    16 planes -- Newton, continuation, FGMRES and StokesPCApply0 with MatVVPC on slabs -- against the one-GPU solve: the same
    stages and Newton steps; the outer Krylov counts within 15 % (the inner solves are truncated GMRES iterations, so rounding
    differences in their reductions make the two runs slightly different preconditioners); two roots of the same residual to the
    Newton tolerance 1e-8, i.e. the same solution to 1e-6."""
    dims, rheo = (128, 128, 128), (1, 1.0, 3.0, 1e-4, 1.0)
    xs, logs = _continuation(dims, 1, cont=4, rheology=rheo)
    xd, logd = _continuation(dims, 8, cont=4, rheology=rheo)
    summary = ([tuple(s[2:4]) for s in logd], [tuple(s[2:4]) for s in logs])
    assert [s[2] for s in logd] == [s[2] for s in logs], summary
    assert all(abs(a[3] - b[3]) <= max(3, 0.15 * b[3]) for a, b in zip(logd, logs)), summary
    assert relerr(xd, xs) < 1e-6, (relerr(xd, xs), summary)
    assert logd[-1][4] < 2.0 * logs[-1][4] + 1e-12, summary                   # final residual norms alike
    print("config 5 over 8 thread ranks: (Newton, Krylov) per stage %r; one GPU %r; rel. difference of the solutions %.2e" % (summary[0], summary[1], relerr(xd, xs)))


def test_null_transport_with_arrays_standing_for_the_peers():
    """chebhip_comm_null_set_shadow (bench.py dist_rank_compute: *_distinct_peer_arrays): one rank of 4 on the NULL transport with three
    arrays of its own standing for the peers' slabs and three for their result arrays.  Without them every "peer" is the rank itself and
    the rows it computes for the other owners land on top of each other in its own result array (timing only); with them each owner's
    rows go to that owner's array: the peers' arrays receive rows, and the rank's own column block is reproducible to the bit."""
    sp = ge.load(); dsp = ge.load_dist()
    dims, G = (70, 70, 66), 4
    comm = dsp.Comm(sp, null=(G, 0))
    D = dsp.DistPoissonC(dims, sp, comm=comm)
    n = D.local_size
    U = torch.randn(n, dtype=torch.float64, device="cuda"); V1 = torch.empty_like(U); V2 = torch.empty_like(U)
    shadow_u = [None] + [torch.randn(n, dtype=torch.float64, device="cuda") for _ in range(G - 1)]
    shadow_t = [None] + [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(G - 1)]
    comm.set_null_shadow(0, shadow_u); comm.set_null_shadow(1, shadow_t)
    D.mult(U, V1)
    D.mult(U, V2)
    torch.cuda.synchronize()
    M0, M1, R = (d - 2 for d in dims)
    w = M1 // G                                         # 68 = 4 x 17: the rank's own column block is columns 0 .. 17 of every plane
    a = V1.cpu().numpy().reshape(-1, M1, R); b = V2.cpu().numpy().reshape(-1, M1, R)
    assert np.isfinite(a[:, :w, :]).all() and np.array_equal(a[:, :w, :], b[:, :w, :])
    assert all(float(t.abs().max()) > 0 for t in shadow_t[1:])          # the peers' result arrays got the rows computed for them
    comm.set_null_shadow(0, [None] * G); comm.set_null_shadow(1, [None] * G)
    D.mult(U, V2); torch.cuda.synchronize()                             # back to the rank's own arrays: runs as before
    D.destroy(); comm.destroy()
