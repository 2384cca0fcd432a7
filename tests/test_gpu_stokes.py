"""GPU parity tests of the Stokes callbacks (stokes.C:499-758) through the C ABI, against the golden
vectors and the CPU oracle.  Tolerance: 1e-10 normwise on N(0,1) inputs (float64) for every block, the pressure
gradient included (its boundary extrapolation is a dot product here, a Neville table in the reference: observed
1.8e-14 ... 9.8e-14 at 64^3 ... 136x132x130, profiles/r03_stokes_parity.txt)."""
import os

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
import oracle_lib as orc
from conftest import relerr, HERE

pytestmark = pytest.mark.gpu
sp = ge.load()
SEED = 20240229
POWER = (1, 1.0, 3.0, 1e-4, 1.0)   # README:52


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def run(fn, x, nout):
    y = torch.full((nout,), float("nan"), dtype=torch.float64, device="cuda")
    fn(dev(x), y)
    torch.cuda.synchronize()
    return y.cpu().numpy()


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(HERE, "golden", "stokes_golden.npz")))


@pytest.mark.parametrize("dims", [(8, 7), (7, 6, 5)])
def test_stokes_golden(g, dims):
    tag = "x".join(str(v) for v in dims)
    d = len(dims)
    op = sp.StokesOp(dims)
    x = g["st_%s_x" % tag]
    X = x.reshape(-1, d + 1)
    vG, pG = np.ascontiguousarray(X[:, :d]).ravel(), np.ascontiguousarray(X[:, d])
    assert relerr(run(op.mult_vv, vG, op.velocity_size), g["st_%s_vv_lin" % tag]) < 1e-10
    assert relerr(run(op.mult_pv, vG, op.pressure_size), g["st_%s_pv" % tag]) < 1e-10
    assert relerr(run(op.mult_vp, pG, op.velocity_size), g["st_%s_vp" % tag]) < 1e-10
    assert relerr(run(op.mult, x, op.global_size), g["st_%s_mult_lin" % tag]) < 1e-10
    # StokesFunction with the power-law rheology, then the Jacobian apply with the state it leaves
    op.set_rheology(*POWER)
    op.set_dirichlet(g["st_%s_fn_dirichlet" % tag])
    op.set_force(g["st_%s_fn_force" % tag])
    y = run(op.function, g["st_%s_fn_x" % tag], op.global_size)
    assert relerr(op.get_state(0), g["st_%s_fn_eta" % tag]) < 1e-10
    assert relerr(op.get_state(1), g["st_%s_fn_deta" % tag]) < 1e-10
    for j in range(d):
        assert relerr(op.get_state(2 + j), g["st_%s_fn_strain" % tag][j]) < 1e-10
    assert relerr(y, g["st_%s_fn_y" % tag]) < 1e-10
    assert relerr(run(op.mult, x, op.global_size), g["st_%s_mult_nl" % tag]) < 1e-10
    op.destroy()


@pytest.mark.parametrize("dims", [(20, 17), (33, 18), (12, 11, 10), (24, 20, 18), (64, 64, 64), (260, 9), (8, 258, 7)],
                         ids=lambda s: "x".join(map(str, s)))
def test_stokes_mult_vs_oracle(dims):
    """StokesMatMult and its three blocks, linear state (BASELINE config 4 is -dim 64,64,64)."""
    d = len(dims)
    op = sp.StokesOp(dims)
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal(op.global_size)
    X = x.reshape(-1, d + 1)
    vG, pG = np.ascontiguousarray(X[:, :d]).ravel(), np.ascontiguousarray(X[:, d])
    nt = 16
    assert relerr(run(op.mult_vv, vG, op.velocity_size), orc.stokes_mult_vv(dims, vG, nthreads=nt)) < 1e-10
    assert relerr(run(op.mult_pv, vG, op.pressure_size), orc.stokes_divergence(dims, vG, nthreads=nt)) < 1e-10
    assert relerr(run(op.mult_vp, pG, op.velocity_size), orc.stokes_mult_vp(dims, pG, nthreads=nt)) < 1e-10
    assert relerr(run(op.mult, x, op.global_size), orc.stokes_mult(dims, x, nthreads=nt)) < 1e-10
    op.destroy()


@pytest.mark.parametrize("dims", [(18, 16), (14, 12, 10), (32, 32, 32), (10, 270)], ids=lambda s: "x".join(map(str, s)))
def test_stokes_function_power_law_vs_oracle(dims):
    """StokesFunction with -rheology 1 -exponent 3 -eps 1e-4 (config 5 parameters) and the Newton-linearised
    StokesMatMult that follows it."""
    d = len(dims)
    op = sp.StokesOp(dims)
    U, U2, dv = orc.stokes_exact(dims, 1)
    rng = np.random.default_rng(SEED)
    xs = U + 0.05 * rng.standard_normal(U.shape)
    op.set_rheology(*POWER)
    op.set_dirichlet(dv)
    op.set_force(U2)
    y = run(op.function, xs, op.global_size)
    yo, eta, deta, strain = orc.stokes_function(dims, xs, dv, U2, POWER, nthreads=16)
    assert relerr(op.get_state(0), eta) < 1e-10
    assert relerr(op.get_state(1), deta) < 1e-10
    assert relerr(y, yo) < 1e-10
    x = rng.standard_normal(op.global_size)
    assert relerr(run(op.mult, x, op.global_size), orc.stokes_mult(dims, x, eta, deta, strain, nthreads=16)) < 1e-10
    op.destroy()


@pytest.mark.parametrize("dims", [(16, 14), (12, 10, 9)])
def test_constant_pressure_null_space(dims):
    """stokes.C:190-212 MatNullSpaceTest on the GPU path: A [0; const] = 0."""
    d = len(dims)
    op = sp.StokesOp(dims)
    x = np.zeros((op.interior_nodes, d + 1))
    x[:, d] = 1.0
    y = run(op.mult, x.ravel(), op.global_size)
    assert np.abs(y).max() < 1e-9          # absolute, on an input of norm sqrt(I)
    op.destroy()


def test_exact2_residual():
    """stokes.C:190-212 with Exact2 (README:43): residual of the exact solution at 20^2 is ~1e-10."""
    dims = (20, 20)
    op = sp.StokesOp(dims)
    U, U2, dv = orc.stokes_exact(dims, 2)
    op.set_dirichlet(dv)
    op.set_force(U2)
    r = run(op.function, U, op.global_size)
    ro, *_ = orc.stokes_function(dims, U, dv, U2, nthreads=4)
    assert np.abs(r - ro).max() < 1e-9     # absolute difference of two residuals of size ~1e-10 .. 1e-6
    assert np.abs(r).max() < 1e-6
    op.destroy()


def test_stokes_linearity_config5_size():
    """Size-independent property at BASELINE config 5 size (128^3): StokesMatMult is linear."""
    dims = (128, 128, 128)
    op = sp.StokesOp(dims)
    torch.manual_seed(SEED)
    x = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    z = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    y1, y2, y3 = (torch.empty_like(x) for _ in range(3))
    op.mult(x, y1)
    op.mult(z, y2)
    op.mult(1.5 * x - 0.5 * z, y3)
    torch.cuda.synchronize()
    assert torch.isfinite(y3).all()
    assert (torch.linalg.norm(1.5 * y1 - 0.5 * y2 - y3) / torch.linalg.norm(y3)).item() < 1e-11
    op.destroy()


def _power_state_test(dims, nthreads):
    """StokesFunction with the power law of README:52, then StokesMatMult linearised about it (Newton), vs the oracle."""
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    rng = np.random.default_rng(SEED)
    x, dv, force, v = rng.standard_normal(g), rng.standard_normal(ndv), rng.standard_normal(g), rng.standard_normal(g)
    op = sp.StokesOp(dims)
    op.set_rheology(*POWER); op.set_dirichlet(dv); op.set_force(force)
    yf = run(op.function, x, g)
    ym = run(op.mult, v, g)
    op.destroy()
    ref_f, eta, deta, strain = orc.stokes_function(dims, x, dv, force, rheology=POWER, mode=orc.FAST, nthreads=nthreads)
    ref_m = orc.stokes_mult(dims, v, eta, deta, strain, mode=orc.FAST, nthreads=nthreads)
    assert relerr(yf, ref_f) < 1e-10 and relerr(ym, ref_m) < 1e-10


def test_stokes_power_law_config5_size_vs_oracle():
    """BASELINE config 5 at its size and rheology: -dim 128,128,128 -rheology 1 (KS = 16 sweeps, power-law node kernel)."""
    _power_state_test((128, 128, 128), 16)


def test_stokes_power_law_96_vs_oracle():
    _power_state_test((96, 80, 72), 16)


@pytest.mark.parametrize("dims", [(96, 96, 96), (128, 128, 128), (136, 132, 130), (200, 140)], ids=lambda s: "x".join(map(str, s)))
def test_stokes_blocks_large_vs_oracle(dims):
    """Linear VV / PV / VP / MatMult on lines of 65..128 points (the KS = 16 kernels) and of 130..200 points (KS = 32, the
    multi-job launches included) vs the oracle."""
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    rng = np.random.default_rng(SEED)
    v, p, x = rng.standard_normal(gv), rng.standard_normal(gp), rng.standard_normal(g)
    op = sp.StokesOp(dims)
    op.set_dirichlet(np.zeros(ndv)); op.set_force(np.zeros(g))
    assert relerr(run(op.mult_vv, v, gv), orc.stokes_mult_vv(dims, v, mode=orc.FAST, nthreads=16)) < 1e-10
    assert relerr(run(op.mult_pv, v, gp), orc.stokes_divergence(dims, v, mode=orc.FAST, nthreads=16)) < 1e-10
    assert relerr(run(op.mult_vp, p, gv), orc.stokes_mult_vp(dims, p, mode=orc.FAST, nthreads=16)) < 1e-10
    assert relerr(run(op.mult, x, g), orc.stokes_mult(dims, x, mode=orc.FAST, nthreads=16)) < 1e-10
    op.destroy()


@pytest.mark.parametrize("dims", [(120, 121, 68), (150, 97, 100), (122, 120, 128), (128, 120, 124), (128, 128, 128), (170, 85, 72)], ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("state", ["power", "eta_only"])
def test_z_direction_in_one_launch_equals_the_separate_passes(dims, state):
    """k_st_zfused16 (d = 3, contiguous lines of 68 .. 128 points): the z third of the gradient launch, the node loop and the z third of
    the divergence launch of StokesMatMult / StokesMatMultVV as ONE launch -- the same arithmetic in the same order, so the results
    equal those of the separate-pass route (option `stokes_z_separate`) to the bit.  Power-law state (eta' != 0: the S0 z term) and a
    variable viscosity with eta' = 0; tiles that end inside the grid (line counts that are not multiples of 16); grids of at least
    14 400 z-lines (smaller ones keep the separate passes)."""
    sp = ge.load()
    import torch
    rng = np.random.default_rng(SEED + 31)
    N = int(np.prod(dims))
    outs = []
    for sep in (0, 1):
        sp.set_option("stokes_z_separate", sep)
        sp.set_option("stokes_pressure_sweeps", 1)            # (like with like: the folded pressure route exists on the fused-z route only)
        try:
            st = sp.StokesOp(dims)
            fn = None
            if state == "power":
                st.set_rheology(1, 1.0, 3.0, 1e-2, 1.0)
                st.set_dirichlet(np.random.default_rng(1).standard_normal(st.dirichlet_size)); st.set_force(np.random.default_rng(6).standard_normal(st.global_size))
                x0 = torch.from_numpy(np.random.default_rng(2).standard_normal(st.global_size)).cuda()
                r0 = torch.full_like(x0, float("nan"))
                st.function(x0, r0); torch.cuda.synchronize()          # (StokesFunction takes the fused launch on two-stream grids: 128^3)
                fn = [r0.cpu().numpy()] + [st.get_state(w) for w in range(5)]
            else:
                st.set_state(0, np.exp(np.random.default_rng(3).uniform(np.log(0.5), np.log(10.0), N)))
            x = torch.from_numpy(np.random.default_rng(4).standard_normal(st.global_size)).cuda()
            v = torch.from_numpy(np.random.default_rng(5).standard_normal(st.velocity_size)).cuda()
            y = torch.full_like(x, float("nan")); w = torch.full_like(v, float("nan"))
            st.mult(x, y); st.mult_vv(v, w); torch.cuda.synchronize()
            outs.append((y.clone(), w.clone(), fn))
            st.destroy()
        finally:
            sp.set_option("stokes_z_separate", 0); sp.set_option("stokes_pressure_sweeps", 0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (
        float((outs[0][0] - outs[1][0]).norm() / outs[1][0].norm()), float((outs[0][1] - outs[1][1]).norm() / outs[1][1].norm()))
    if outs[0][2] is not None:                                # residual, eta, eta', the three strain blocks
        for k, (p_, q_) in enumerate(zip(outs[0][2], outs[1][2])):
            assert np.array_equal(p_, q_), k


@pytest.mark.parametrize("dims", [(128, 128, 128), (120, 121, 68), (150, 97, 100), (120, 120, 72)], ids=lambda d: "x".join(map(str, d)))
def test_pressure_inside_the_stress_equals_the_pressure_sweeps(dims):
    """Round 5: on the fused-z route StokesMatMult / StokesFunction deliver -div(tau - p_ext I) from the three divergence sweeps -- the
    face-interior values of p extrapolated along each line by k_st_pfaces (StokesPressureReduceOrder, stokes.C:1029-1080), p subtracted
    from the diagonal stress inside k_st_zfused16 -- instead of running the three pressure-gradient sweeps and adding grad p in the
    scatter (option `stokes_pressure_sweeps` = 1).  Same operator, sums in another order: <= 1e-13; short z lines (dead loader slots),
    partial tiles, odd extents; non-zero Dirichlet values and force.  (Against the oracle: the 128^3 tests above run the folded route.)"""
    sp = ge.load()
    import torch
    st = sp.StokesOp(dims)
    st.set_rheology(1, 1.0, 3.0, 1e-2, 1.0)
    st.set_dirichlet(np.random.default_rng(1).standard_normal(st.dirichlet_size)); st.set_force(np.random.default_rng(6).standard_normal(st.global_size))
    x0 = torch.from_numpy(np.random.default_rng(2).standard_normal(st.global_size)).cuda()
    x = torch.from_numpy(np.random.default_rng(4).standard_normal(st.global_size)).cuda()
    res = []
    try:
        for sw in (1, 0):
            sp.set_option("stokes_pressure_sweeps", sw)
            r0 = torch.full_like(x0, float("nan")); y = torch.full_like(x, float("nan"))
            st.function(x0, r0); st.mult(x, y); torch.cuda.synchronize()
            res.append((r0.cpu().numpy(), y.cpu().numpy(), [st.get_state(w) for w in range(5)]))
    finally:
        sp.set_option("stokes_pressure_sweeps", 0)
    st.destroy()
    assert relerr(res[1][0], res[0][0]) < 1e-13 and relerr(res[1][1], res[0][1]) < 1e-13
    for a, b in zip(res[0][2], res[1][2]):                    # the state StokesFunction leaves (eta, eta', strain) is untouched by the route
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dims", [(120, 121, 68), (150, 97, 100)], ids=lambda d: "x".join(map(str, d)))
def test_pressure_inside_the_stress_at_odd_shapes_vs_oracle(dims):
    """the folded pressure route met the oracle at 96^3 and 128^3 only; its odd shapes (short z lines with dead loader
    slots, partial tiles, an odd middle extent) were checked against the pressure-sweeps route, i.e. transitively.  Here directly: power-law
    StokesFunction and StokesMatMult (linearised about its state) against the oracle, non-zero Dirichlet values and force."""
    sp = ge.load()
    import torch
    power = (1, 1.0, 3.0, 1e-2, 1.0)
    st = sp.StokesOp(dims)
    rng = np.random.default_rng(SEED + 21)
    x0 = rng.standard_normal(st.global_size); x = rng.standard_normal(st.global_size)
    dv = rng.standard_normal(st.dirichlet_size); force = rng.standard_normal(st.global_size)
    st.set_rheology(*power); st.set_dirichlet(dv); st.set_force(force)
    r0 = torch.full((st.global_size,), float("nan"), dtype=torch.float64, device="cuda"); y = torch.full_like(r0, float("nan"))
    st.function(torch.from_numpy(x0).cuda(), r0); st.mult(torch.from_numpy(x).cuda(), y); torch.cuda.synchronize()
    st.destroy()
    ref_f, eta, deta, strain = orc.stokes_function(dims, x0, dv, force, rheology=power, mode=orc.FAST, nthreads=8)
    ref_m = orc.stokes_mult(dims, x, eta, deta, strain, mode=orc.FAST, nthreads=8)
    assert relerr(r0.cpu().numpy(), ref_f) < 1e-10 and relerr(y.cpu().numpy(), ref_m) < 1e-10


def test_folded_pressure_route_with_an_8_byte_aligned_result_vector():
    """The scatter of the folded pressure route is the 16-byte pair kernel; a result vector at an odd 8-byte offset must fall back to the
    pressure-gradient sweeps (not to a scatter that would add a stale grad p): same answer as with an aligned vector."""
    sp = ge.load()
    import torch
    dims = (120, 122, 68)
    st = sp.StokesOp(dims)
    st.set_rheology(1, 1.0, 3.0, 1e-2, 1.0)
    st.set_dirichlet(np.random.default_rng(1).standard_normal(st.dirichlet_size)); st.set_force(np.random.default_rng(6).standard_normal(st.global_size))
    g = st.global_size
    x0 = torch.from_numpy(np.random.default_rng(2).standard_normal(g)).cuda()
    x = torch.from_numpy(np.random.default_rng(4).standard_normal(g)).cuda()
    buf = torch.full((g + 1,), float("nan"), dtype=torch.float64, device="cuda")
    ya, yu = torch.empty_like(x), buf[1:]
    assert yu.data_ptr() % 16 == 8
    r = torch.empty_like(x0)
    st.function(x0, r)
    st.mult(x, ya); st.mult(x, yu); torch.cuda.synchronize()
    assert relerr(yu.cpu().numpy(), ya.cpu().numpy()) < 1e-13
    ra = r.clone()
    st.function(x0, yu); torch.cuda.synchronize()
    assert relerr(yu.cpu().numpy(), ra.cpu().numpy()) < 1e-13
    st.destroy()
