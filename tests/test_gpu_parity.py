"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the
C ABI, against the committed golden vectors and the CPU oracle on the same seeded inputs.

Tolerances (float64, stated per BASELINE.md / SURVEY 7.4):
  white-noise inputs : ||y - y_ref||_2 / ||y_ref||_2 <= 1e-10   (observed ~1e-13)
  smooth fields      : each path is compared with the analytic answer, and the HIP error
                       must not exceed a small multiple of the oracle's own error."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
import oracle_lib as orc
from conftest import relerr

pytestmark = pytest.mark.gpu
sp = ge.load()
TOL = 1e-10
SEED = 20240229


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def gpu_cheb(x, tr):
    plan = sp.ChebPlan(x.shape, tr)
    xd = dev(x)
    yd = torch.full_like(xd, float("nan"))
    plan.mult(xd, yd)
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), torch.from_numpy(np.ascontiguousarray(x)))  # input untouched
    plan.destroy()
    return yd.cpu().numpy()


def _cases(g):
    out = []
    for k in g:
        if k.startswith("cheb_") and "_tr" in k:
            tag, kind, tr = k[5:].rsplit("_", 2)
            out.append((tag, kind, int(tr[2:])))
    return sorted(out)


@pytest.mark.parametrize("n0", [1790, 1800], ids=["just_below_the_32bit_offset_limit", "just_above_it"])
def test_cheb_arrays_around_the_buffer_offset_limit(n0):
    """cheb_sweep_vec4_kernel addresses its arrays with 32-bit buffer offsets and takes arrays of less than
    0x38000000 bytes (939.5 MB; csrc/sweep_vec.hip prepare_v); larger ones go to the general kernel of sweep.hip -- silently, so this
    is the test that the hand-over is right on BOTH sides of the limit (a 512^3 scalar field is 1 GiB on a 288-GB part):
    (1790, 256, 256) = 938.5 MB runs the fast kernel with offsets up to the limit, (1800, 256, 256) = 943.7 MB does not qualify.
    256-point lines along dimension 1 (strided) and 2 (contiguous) of a seeded N(0,1) array, compared with the oracle on sub-blocks
    of whole lines taken at the start, in the middle and at the very end of the array (lines are independent: chebyshev.c:162-193)."""
    dims = (n0, 256, 256)
    nbytes = 8 * n0 * 256 * 256
    assert (nbytes < 0x38000000) == (n0 == 1790)
    g = torch.Generator(device="cuda").manual_seed(SEED + n0)
    xd = torch.randn(dims, dtype=torch.float64, device="cuda", generator=g)
    yd = torch.empty_like(xd)
    blocks = [(0, 3), (n0 // 2 - 1, n0 // 2 + 2), (n0 - 3, n0)]
    for tr in (1, 2):
        plan = sp.ChebPlan(dims, tr)
        yd.fill_(float("nan"))
        plan.mult(xd.view(-1), yd.view(-1))
        torch.cuda.synchronize()
        plan.destroy()
        assert bool(torch.isfinite(yd).all())
        for lo, hi in blocks:
            x = xd[lo:hi].cpu().numpy()
            ref = orc.cheb_mult(x, tr, mode=orc.FAST, nthreads=4)
            assert relerr(yd[lo:hi].cpu().numpy(), ref) < TOL, (tr, lo)
    del xd, yd
    torch.cuda.empty_cache()


def test_native_library_loaded():
    L = sp.lib()
    assert L.chebhip_arch() == b"gfx950"
    before = L.chebhip_launch_count()
    gpu_cheb(np.ones((4, 4)), 0)
    assert L.chebhip_launch_count() == before + 1


def test_cheb_golden(cheb_golden):
    n = 0
    for tag, kind, tr in _cases(cheb_golden):
        x = cheb_golden["cheb_%s_%s_in" % (tag, kind)]
        ref = cheb_golden["cheb_%s_%s_tr%d" % (tag, kind, tr)]
        y = gpu_cheb(x, tr)
        assert np.isfinite(y).all(), (tag, kind, tr)
        assert relerr(y, ref) < TOL, (tag, kind, tr, relerr(y, ref))
        n += 1
    assert n > 40


def test_cheb_host_pointer_path(cheb_golden):
    x = cheb_golden["cheb_8x7x5_rand_in"]
    for tr in range(3):
        plan = sp.ChebPlan(x.shape, tr)
        y = plan.mult_host(x)
        assert relerr(y, cheb_golden["cheb_8x7x5_rand_tr%d" % tr]) < TOL


SHAPES = [
    (2,), (3,), (4,), (16,), (17,), (31,), (100,), (255,), (256,),
    (2, 2), (3, 5), (32, 32), (33, 17), (7, 256), (256, 7), (129, 130),
    (5, 6, 7), (16, 16, 16), (33, 32, 31), (31, 33, 35), (64, 64, 64), (2, 200, 3), (130, 20, 10),
    (8, 7, 5, 3), (20, 18, 16, 3), (3, 4, 5, 6, 7), (64, 64, 3), (40, 48, 56, 3),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_cheb_random_vs_oracle(shape):
    rng = np.random.default_rng(SEED + sum(shape))
    x = rng.standard_normal(shape)
    for tr in range(len(shape)):
        if shape[tr] < 2:
            continue
        y = gpu_cheb(x, tr)
        ref = orc.cheb_mult(x, tr, orc.FAST, nthreads=8)
        assert relerr(y, ref) < TOL, (shape, tr, relerr(y, ref))


@pytest.mark.parametrize("shape,tr", [((128, 128, 128), 0), ((128, 128, 128), 1), ((128, 128, 128), 2),
                                      ((256, 64, 40), 0), ((40, 256, 64), 1), ((40, 64, 256), 2)])
def test_cheb_large_vs_oracle(shape, tr):
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal(shape)
    y = gpu_cheb(x, tr)
    ref = orc.cheb_mult(x, tr, orc.FAST, nthreads=16)
    assert relerr(y, ref) < TOL


@pytest.mark.parametrize("shape,tr", [((300, 40), 0), ((12, 513), 1), ((6, 400, 5), 1), ((257, 33), 0), ((19, 258), 1), ((1024, 21), 0),
                                      ((3, 1024), 1), ((5, 700, 17), 1), ((640, 4, 9), 0), ((2, 3, 999), 2), ((1025, 18), 0), ((7, 1100), 1)])
@pytest.mark.parametrize("gemm", [0, 1], ids=["own", "rocblas"])
def test_cheb_long_lines(shape, tr, gemm):
    """Lines of 257 .. 1024 points: cheb_sweep_xl_kernel (the matrix halves stream past a tile of lines held in LDS), both
    tilings, ragged tiles, odd and even extents; beyond 1024 and with option long_lines_gemm: rocBLAS, or the VALU kernel
    where the layout is no GEMM."""
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal(shape)
    sp.set_option("long_lines_gemm", gemm)
    try:
        y = gpu_cheb(x, tr)
    finally:
        sp.set_option("long_lines_gemm", 0)
    ref = orc.cheb_mult(x, tr, orc.FAST, nthreads=8)
    assert relerr(y, ref) < TOL, relerr(y, ref)


@pytest.mark.parametrize("shape,tr", [((512, 16400), 0), ((16400, 512), 1), ((1000, 8200), 0), ((4104, 2, 1000), 2)], ids=str)
def test_cheb_long_lines_large(shape, tr):
    """Arrays large enough for the long-line kernel's tiles of 32 lines with two m-tiles per wave (at least 512 workgroups),
    with a ragged last tile; 512 points = one row block of 16 m-tiles, 1000 points = two (the last m-tile partly real)."""
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal(shape)
    y = gpu_cheb(x, tr)
    ref = orc.cheb_mult(x, tr, orc.FAST, nthreads=16)
    assert relerr(y, ref) < TOL, relerr(y, ref)


@pytest.mark.parametrize("dims", [(33, 32, 31), (64, 64, 64)])
def test_cheb_exp_known_answer(dims):
    """cheb.c:73-112 on the GPU: d/dx_d (e^x+e^y+e^z) = e^{x_d}; HIP error <= 4x oracle error + eps."""
    grids = np.meshgrid(*[np.cos(np.arange(p) * np.pi / (p - 1)) for p in dims], indexing="ij")
    u = sum(np.exp(g) for g in grids)
    for tr in range(3):
        truth = np.exp(grids[tr])
        e_gpu = np.abs(gpu_cheb(u, tr) - truth).max()
        e_orc = np.abs(orc.cheb_mult(u, tr, orc.FAST, 8) - truth).max()
        assert e_gpu < 4 * e_orc + 1e-13, (tr, e_gpu, e_orc)
        assert e_gpu < 1e-10


def test_cheb_1d_known_answer():
    """cheb.c:66-71,95-103 (ChebD1Mult = rank-1 plan)."""
    for m1 in (5, 16, 24):
        x = np.cos(np.arange(m1) * np.pi / (m1 - 1))
        y = gpu_cheb(np.exp(x), 0)
        ref = orc.cheb_mult(np.exp(x), 0, orc.DIRECT)
        assert relerr(y, ref) < 1e-12


def test_linearity_full_size():
    """Size-independent property at the BASELINE size: D(a x + b z) = a D x + b D z, 256^3, every axis."""
    torch.manual_seed(SEED)
    shape = (256, 256, 256)
    x = torch.randn(shape, dtype=torch.float64, device="cuda")
    z = torch.randn(shape, dtype=torch.float64, device="cuda")
    y1, y2, y3 = (torch.empty_like(x) for _ in range(3))
    for tr in range(3):
        plan = sp.ChebPlan(shape, tr)
        plan.mult(x, y1)
        plan.mult(z, y2)
        plan.mult(2.0 * x - 3.0 * z, y3)
        torch.cuda.synchronize()
        num = torch.linalg.norm((2.0 * y1 - 3.0 * y2 - y3).ravel()).item()
        den = torch.linalg.norm(y3.ravel()).item()
        assert num / den < 1e-12
        # constants differentiate to zero, x_tr to one
        plan.mult(torch.ones_like(x), y1)
        torch.cuda.synchronize()
        assert y1.abs().max().item() < 1e-9
        plan.destroy()


def test_polynomial_exactness_full_size():
    """d/dx of x^5 sampled on the 256-point CGL grid is 5x^4 to rounding (degree < P)."""
    P = 256
    xg = np.cos(np.arange(P) * np.pi / (P - 1))
    for tr in range(3):
        shp = [1, 1, 1]
        shp[tr] = P
        f = np.broadcast_to((xg ** 5).reshape(shp), (P, 64, 64) if tr == 0 else ((64, P, 64) if tr == 1 else (64, 64, P)))
        y = gpu_cheb(np.ascontiguousarray(f), tr)
        truth = np.broadcast_to((5 * xg ** 4).reshape(shp), f.shape)
        assert np.abs(y - truth).max() < 2e-8   # ||D||~P^2 amplifies eps


# ----------------------------------------------------------------------------------------------
# operator level
# ----------------------------------------------------------------------------------------------
def test_elliptic_golden(ell_golden):
    g = ell_golden
    for dims in [(8, 6), (32, 32), (9, 8, 7)]:
        tag = "x".join(str(s) for s in dims)
        op = sp.EllipticOp(dims)
        U = g["ell_%s_U" % tag]
        Ud = dev(U)
        Vd = torch.full_like(Ud, float("nan"))
        op.mult(Ud, Vd)
        torch.cuda.synchronize()
        assert relerr(Vd.cpu().numpy(), g["ell_%s_mult_lin" % tag]) < TOL
        assert relerr(op.mult_host(U), g["ell_%s_mult_lin" % tag]) < TOL
        # FormFunction on the -exact 2 field with gamma = 4, exponent = 2
        op.set_dirichlet(g["ell_%s_exact2_dirichlet" % tag])
        rhs = op.function_host(g["ell_%s_exact2_u" % tag], g["ell_%s_exact2_b" % tag], 4.0, 2.0)
        scale = np.abs(g["ell_%s_exact2_b" % tag]).max()
        assert np.abs(rhs - g["ell_%s_fn_rhs" % tag]).max() < 1e-9 * scale
        assert relerr(op.get_state(0), g["ell_%s_fn_eta" % tag]) < 1e-14
        assert relerr(op.get_state(1), g["ell_%s_fn_deta" % tag]) < 1e-14
        for k in range(len(dims)):
            assert relerr(op.get_state(2 + k), g["ell_%s_fn_gradu" % tag][k]) < 1e-10
        # Jacobian apply with the nonlinear state left behind by FormFunction
        assert relerr(op.mult_host(U), g["ell_%s_mult_nl" % tag]) < TOL
        op.destroy()


@pytest.mark.parametrize("dims", [(32, 32), (5, 4), (3, 3), (40,), (64, 64, 64), (20, 18, 16), (6, 5, 4, 3), (33, 70, 9),
                                  (12, 12, 12, 12, 12), (300, 20), (12, 260, 6), (514,),
                                  # padded-accumulator layout of the constant-coefficient path (interior lines > 64, even)
                                  (72, 68, 70), (100, 90), (200, 68, 98), (68, 132, 76), (71, 68, 70), (256, 130)],
                         ids=lambda s: "x".join(map(str, s)))
def test_elliptic_mult_vs_oracle(dims):
    """MatMult_Elliptic, linear Poisson state (config 1 is -dim 32,32)."""
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED)
    U = rng.standard_normal(op.global_size)
    V = op.mult_host(U)
    ref = orc.elliptic_mult(dims, U, mode=orc.FAST, nthreads=8)
    assert relerr(V, ref) < TOL
    op.destroy()


def test_elliptic_mult_128_vs_oracle():
    """BASELINE config 2: 3-D Poisson -dim 128,128,128."""
    dims = (128, 128, 128)
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED)
    U = rng.standard_normal(op.global_size)
    V = op.mult_host(U)
    ref = orc.elliptic_mult(dims, U, mode=orc.FAST, nthreads=16)
    assert relerr(V, ref) < TOL
    op.destroy()


def test_elliptic_mult_256_vs_oracle():
    """BASELINE config 3 on one GPU: 3-D Poisson -dim 256,256,256, the bench workload, against the oracle."""
    dims = (256, 256, 256)
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED)
    U = rng.standard_normal(op.global_size)
    V = op.mult_host(U)
    ref = orc.elliptic_mult(dims, U, mode=orc.FAST, nthreads=16)
    assert relerr(V, ref) < TOL
    op.destroy()


def test_elliptic_misaligned_vectors_take_the_general_kernels():
    """MatShell vectors that are only 8-byte aligned (a sub-vector of a bigger allocation): the 16-byte kernels with
    per-array geometry do not apply; the general kernels must give the same answers (linear matvec, FormFunction, Jacobian)."""
    import torch
    dims = (68, 70, 72)
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED + 11)
    n = op.global_size

    def off8(a=None):
        buf = torch.empty(n + 1, dtype=torch.float64, device="cuda")
        v = buf[1:]
        assert v.data_ptr() % 16 == 8
        if a is not None:
            v.copy_(torch.from_numpy(a))
        return v
    Uh = rng.standard_normal(n)
    U, V = off8(Uh), off8()
    op.mult(U, V); torch.cuda.synchronize()
    assert relerr(V.cpu().numpy(), orc.elliptic_mult(dims, Uh, mode=orc.FAST, nthreads=16)) < TOL
    uh = rng.random(n) + 0.5; bh = rng.standard_normal(n); dv = rng.random(op.dirichlet_size) + 0.5
    op.set_dirichlet(dv)
    u, b, r = off8(uh), off8(bh), off8()
    op.function(u, b, r, 1.5, 2.0); torch.cuda.synchronize()
    rhs_o, eta, deta, gradu = orc.elliptic_function(dims, uh, bh, dv, 1.5, 2.0, mode=orc.FAST, nthreads=16)
    assert relerr(r.cpu().numpy(), rhs_o) < TOL
    op.mult(U, V); torch.cuda.synchronize()
    assert relerr(V.cpu().numpy(), orc.elliptic_mult(dims, Uh, eta, deta, gradu, mode=orc.FAST, nthreads=16)) < TOL
    op.destroy()


def test_elliptic_nonlinear_256_vs_oracle():
    """BASELINE config 3 with variable coefficients (-gamma 4 -exponent 2, tests.sh:10) at full size: FormFunction and the
    Jacobian apply of cheb_fused4_kernel at 256^3 against the oracle (16 threads)."""
    dims = (256, 256, 256)
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED + 3)
    u = rng.random(op.global_size) + 0.5
    b = rng.standard_normal(op.global_size)
    dv = rng.random(op.dirichlet_size) + 0.5
    op.set_dirichlet(dv)
    rhs = op.function_host(u, b, 4.0, 2.0)
    rhs_o, eta, deta, gradu = orc.elliptic_function(dims, u, b, dv, 4.0, 2.0, mode=orc.FAST, nthreads=16)
    assert relerr(rhs, rhs_o) < TOL
    U = rng.standard_normal(op.global_size)
    V = op.mult_host(U)
    ref = orc.elliptic_mult(dims, U, eta, deta, gradu, mode=orc.FAST, nthreads=16)
    assert relerr(V, ref) < TOL
    op.destroy()


@pytest.mark.parametrize("dims", [(24, 20), (12, 11, 10)])
def test_elliptic_nonlinear_vs_oracle(dims):
    """FormFunction + Jacobian apply with gamma != 0 (elliptic.C:481-533, 297-339)."""
    op = sp.EllipticOp(dims)
    u, u2, dv = orc.elliptic_exact(dims, 0, gamma=4.0, exponent=2.0, cos_scale=3.0)
    op.set_dirichlet(dv)
    rhs = op.function_host(u, u2, 4.0, 2.0)
    rhs_o, eta, deta, gradu = orc.elliptic_function(dims, u, u2, dv, 4.0, 2.0, mode=orc.FAST)
    assert np.abs(rhs - rhs_o).max() < 1e-9 * np.abs(u2).max()
    rng = np.random.default_rng(SEED)
    U = rng.standard_normal(op.global_size)
    V = op.mult_host(U)
    ref = orc.elliptic_mult(dims, U, eta, deta, gradu, mode=orc.FAST)
    assert relerr(V, ref) < TOL
    op.destroy()


@pytest.mark.parametrize("dims,exponent", [((130, 66), 2.0), ((20, 129, 18), 3.0), ((64, 64, 64), 2.0), ((33, 40), 2.5), ((48, 31, 16), 0.5),
                                           ((264, 24), 2.0)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else "e%g" % v)
def test_elliptic_nonlinear_random_state(dims, exponent):
    """FormFunction and the Jacobian apply on a positive random state, at sizes that run the KS = 16 / 32
    kernels, odd extents (the self-mirrored mid point), integer exponents (products) and real ones (pow)."""
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED)
    u = rng.random(op.global_size) + 0.5
    b = rng.standard_normal(op.global_size)
    dv = rng.random(op.dirichlet_size) + 0.5
    op.set_dirichlet(dv)
    rhs = op.function_host(u, b, 1.5, exponent)
    rhs_o, eta, deta, gradu = orc.elliptic_function(dims, u, b, dv, 1.5, exponent, mode=orc.FAST)
    assert relerr(rhs, rhs_o) < TOL
    assert relerr(op.get_state(0), eta) < 1e-14 and relerr(op.get_state(1), deta) < 1e-14
    U = rng.standard_normal(op.global_size)
    V = op.mult_host(U)
    ref = orc.elliptic_mult(dims, U, eta, deta, gradu, mode=orc.FAST)
    assert relerr(V, ref) < TOL
    op.destroy()


@pytest.mark.parametrize("dims,exponent", [((136, 200), 2.0), ((68, 70, 72), 2.0), ((132, 68, 130), 3.0), ((256, 66, 68), 2.5),
                                           ((66, 128, 254), 2.0)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else "e%g" % v)
def test_elliptic_nonlinear_straight_line_kernel(dims, exponent):
    """The shapes cheb_fused4_kernel serves (d = 2, 3, even extents of 66..256 points, KS = 16 and 32 mixed): FormFunction
    with and without b, the stored gradient (c->gradu, elliptic.C:497-499), the Jacobian apply -- against the oracle."""
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED + 1)
    u = rng.random(op.global_size) + 0.5
    b = rng.standard_normal(op.global_size)
    dv = rng.random(op.dirichlet_size) + 0.5
    op.set_dirichlet(dv)
    rhs0 = op.function_host(u, None, 1.5, exponent)
    rhs = op.function_host(u, b, 1.5, exponent)
    rhs_o, eta, deta, gradu = orc.elliptic_function(dims, u, b, dv, 1.5, exponent, mode=orc.FAST, nthreads=16)
    assert relerr(rhs, rhs_o) < TOL
    assert relerr(rhs0, rhs_o + b) < TOL
    for k in range(len(dims)):
        assert relerr(op.get_state(2 + k), gradu[k]) < TOL
    for seed in (0, 1):
        U = np.random.default_rng(SEED + 7 + seed).standard_normal(op.global_size)
        V = op.mult_host(U)
        ref = orc.elliptic_mult(dims, U, eta, deta, gradu, mode=orc.FAST, nthreads=16)
        assert relerr(V, ref) < TOL
    op.destroy()


@pytest.mark.parametrize("dims", [(136, 200), (68, 70, 72), (66, 128, 254), (256, 66, 68)], ids=lambda v: "x".join(map(str, v)))
def test_elliptic_formfunction_interior_line_path(dims):
    """FormFunction with homogeneous Dirichlet rows and the default exponent 2 (elliptic.C:141, 468-476): three launches on
    the interior lines, no gather pass, eta / deta formed from w0 only when somebody reads them.  Residual with and
    without b, the stored state (eta, deta, grad u on EVERY node, zero lines inside the boundary included), the Jacobian
    apply from that state (coefficient pairs straight from w0), the preconditioner matrix from it, and the hand-over
    between this path and the general one (non-zero Dirichlet rows) in both directions -- against the oracle."""
    d = len(dims)
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED + 11)
    u = rng.random(op.global_size) + 0.5
    b = rng.standard_normal(op.global_size)
    zero = np.zeros(op.dirichlet_size)

    def check(dv, gamma):
        rhs0 = op.function_host(u, None, gamma, 2.0)
        rhs = op.function_host(u, b, gamma, 2.0)
        rhs_o, eta, deta, gradu = orc.elliptic_function(dims, u, b, dv, gamma, 2.0, mode=orc.FAST, nthreads=16)
        assert relerr(rhs, rhs_o) < TOL and relerr(rhs0, rhs_o + b) < TOL
        U = np.random.default_rng(SEED + 12).standard_normal(op.global_size)
        V = op.mult_host(U)                                   # the Jacobian apply BEFORE anything has asked for eta / deta
        assert relerr(V, orc.elliptic_mult(dims, U, eta, deta, gradu, mode=orc.FAST, nthreads=16)) < TOL
        for k in range(d):
            g = op.get_state(2 + k)
            assert relerr(g, gradu[k]) < TOL
            assert np.abs(g - gradu[k]).max() <= 1e-10 * np.abs(gradu[k]).max()          # lines inside the boundary too
        assert relerr(op.get_state(0), eta) < 1e-14 and relerr(op.get_state(1), deta) < 1e-14
        return eta, deta, gradu
    check(zero, 1.5)                                          # never set: homogeneous
    dv = rng.random(op.dirichlet_size) + 0.5
    op.set_dirichlet(dv)
    check(dv, 1.5)                                            # general path: gather pass, non-zero lines inside the boundary
    op.set_dirichlet(zero)
    eta, deta, gradu = check(zero, 0.75)                      # back: the boundary lines must read zero again
    if d == 2:                                                # the preconditioner matrix right after a FormFunction on this path
        op.function_host(u, b, 0.75, 2.0)
        pc = sp.FdPc(op, sweeps=0)
        P = orc.fd_matrix(dims, eta, deta, gradu)
        x = rng.standard_normal(op.global_size)
        y = torch.empty(op.global_size, dtype=torch.float64, device="cuda")
        pc.mult(torch.from_numpy(x).cuda(), y)
        assert relerr(y.cpu().numpy(), P @ x) < 1e-12
        pc.destroy()
    op.destroy()


def test_elliptic_exact_residual():
    """elliptic.C:193-209 with -exact 2: the residual of the polynomial exact solution is ~0."""
    dims = (12, 11, 10)
    op = sp.EllipticOp(dims)
    u, u2, dv = orc.elliptic_exact(dims, 2)
    op.set_dirichlet(dv)
    rhs = op.function_host(u, u2, 0.0, 2.0)
    assert np.abs(rhs).max() < 1e-9 * np.abs(u2).max()
    op.destroy()


def test_elliptic_symmetry_full_size():
    """Property at BASELINE config 3 size (256^3): linearity and definiteness of the Poisson matvec:
    <U, A U> > 0 and A(aU + bZ) = aAU + bAZ."""
    dims = (256, 256, 256)
    op = sp.EllipticOp(dims)
    torch.manual_seed(SEED)
    U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    Z = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    AU, AZ, AC = (torch.empty_like(U) for _ in range(3))
    op.mult(U, AU)
    op.mult(Z, AZ)
    op.mult(0.5 * U + 2.0 * Z, AC)
    torch.cuda.synchronize()
    assert torch.isfinite(AC).all()
    num = torch.linalg.norm(0.5 * AU + 2.0 * AZ - AC).item()
    assert num / torch.linalg.norm(AC).item() < 1e-12
    assert torch.dot(U, AU).item() > 0
    op.destroy()


# ----------------------------------------------------------------------------------------------
# slab / pencil building block and the distributed driver at G = 1
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,axis", [((30, 18, 14), 0), ((30, 18, 14), 1), ((30, 18, 14), 2), ((7, 62), 1),
                                        ((126, 40, 33), 0), ((5, 254, 20), 1), ((6, 10, 254), 2), ((4, 5, 6, 3), 2)])
def test_lap1d_vs_oracle(shape, axis):
    """cheb_apply_lap1d: y = acc + alpha * D D x on interior-layout tensors (zero end points)."""
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal(shape)
    acc = rng.standard_normal(shape)
    pad = [(0, 0)] * len(shape)
    pad[axis] = (1, 1)
    t = orc.cheb_mult(orc.cheb_mult(np.pad(x, pad), axis, orc.FAST, 8), axis, orc.FAST, 8)
    sl = [slice(None)] * len(shape)
    sl[axis] = slice(1, -1)
    t = t[tuple(sl)]
    plan = sp.Lap1dPlan(shape, axis)
    xd, ad = dev(x).reshape(-1), dev(acc).reshape(-1)
    yd = torch.full_like(xd, float("nan"))
    plan.apply(xd, yd, None, 1.0)
    torch.cuda.synchronize()
    assert relerr(yd.cpu().numpy(), t) < TOL
    plan.apply(xd, yd, ad, -1.0)
    torch.cuda.synchronize()
    assert relerr(yd.cpu().numpy(), acc - t) < TOL
    plan.apply(xd, ad, ad, -1.0)          # acc aliasing the output
    torch.cuda.synchronize()
    assert relerr(ad.cpu().numpy(), acc - t) < TOL
    plan.destroy()


@pytest.mark.parametrize("geom", [(5, 14, 6, (0, 5, 10, 14)), (32, 254, 254, tuple(int(v) for v in np.cumsum([0] + [32] * 6 + [31] * 2))),
                                  (3, 7, 1, (0, 0, 7)), (4, 9, 3, (0, 9))], ids=lambda g: "%dx%dx%d-G%d" % (g[0], g[1], g[2], len(g[3]) - 1))
def test_slab_pack_unpack(geom):
    """cheb_slab_pack / cheb_slab_unpack_add against slicing (the exchange-buffer layout of dist.py)."""
    m0, M1, R, c1 = geom
    rng = np.random.default_rng(SEED)
    a = rng.standard_normal((m0, M1, R)); acc = rng.standard_normal(m0 * M1 * R)
    ref = np.concatenate([a[:, c1[s]:c1[s + 1], :].ravel() for s in range(len(c1) - 1)])
    ad = torch.from_numpy(a.reshape(-1)).cuda(); buf = torch.full_like(ad, float("nan"))
    sp.slab_pack(ad, buf, m0, M1, R, c1)
    assert np.array_equal(buf.cpu().numpy(), ref)
    out = torch.full_like(ad, float("nan"))
    sp.slab_unpack_add(buf, torch.from_numpy(acc).cuda(), out, m0, M1, R, c1)
    assert np.array_equal(out.cpu().numpy(), acc + a.reshape(-1))
    sp.slab_unpack_add(buf, None, out, m0, M1, R, c1)
    assert np.array_equal(out.cpu().numpy(), a.reshape(-1))


@pytest.mark.parametrize("dims", [(20, 18, 16), (34, 33), (64, 64, 64)])
def test_dist_driver_single_rank(dims):
    """DistPoissonOp with G = 1 (no process group) goes through pack / pencil / unpack and must equal
    the single-GPU operator to rounding and the oracle to the parity bar."""
    dsp = ge.load_dist()
    dop = dsp.DistPoissonOp(dims, dsp.HipBackend(sp))
    U = dop.random_input(SEED)
    V = torch.empty_like(U)
    dop.mult(U, V)
    torch.cuda.synchronize()
    op = sp.EllipticOp(dims)
    V1 = torch.empty_like(U)
    op.mult(U, V1)
    torch.cuda.synchronize()
    assert relerr(V.cpu().numpy(), V1.cpu().numpy()) < 1e-13
    ref = orc.elliptic_mult(dims, U.cpu().numpy(), mode=orc.FAST, nthreads=8)
    assert relerr(V.cpu().numpy(), ref) < TOL
    op.destroy()
