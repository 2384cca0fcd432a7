"""GPU tests of the Krylov driver and StokesMatMultSchur (8a, a16) against dense algebra on the
oracle's operators.  The dense matrices are built column by column from oracle applies at small sizes."""
import numpy as np
import pytest

from conftest import HERE  # noqa: F401
import oracle_lib as orc

pytestmark = pytest.mark.gpu
SEED = 20240229


@pytest.fixture(scope="module")
def sp():
    import __graft_entry__ as ge
    return ge.load()


def dense(apply, n, m=None):
    m = n if m is None else m
    A = np.empty((m, n))
    e = np.zeros(n)
    for j in range(n):
        e[j] = 1.0
        A[:, j] = apply(e)
        e[j] = 0.0
    return A


@pytest.mark.parametrize("dims", [(12, 10), (8, 7, 6)], ids=lambda d: "x".join(map(str, d)))
def test_fgmres_poisson_vs_dense_solve(sp, dims):
    """KSPSolve's job on the linear Poisson operator: x = A^{-1} b to the requested tolerance."""
    import torch
    op = sp.EllipticOp(dims)
    n = op.global_size
    A = dense(lambda e: orc.elliptic_mult(dims, e, mode=orc.DIRECT), n)
    rng = np.random.default_rng(SEED)
    b = rng.standard_normal(n)
    x_ref = np.linalg.solve(A, b)
    ks = sp.Fgmres(n, restart=30, rtol=1e-12, max_it=2000)
    bd = torch.from_numpy(b).cuda(); xd = torch.empty_like(bd)
    ks.solve(op, bd, xd)
    torch.cuda.synchronize()
    assert ks.reason == 2 and 0 < ks.iterations <= 2000
    x = xd.cpu().numpy()
    assert np.linalg.norm(b - A @ x) <= 2e-12 * np.linalg.norm(b)
    assert np.linalg.norm(x - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
    # zero right-hand side: converged immediately, zero iterations
    ks.solve(op, torch.zeros_like(bd), xd)
    assert ks.iterations == 0 and float(xd.abs().max()) == 0.0
    # a non-zero initial guess is used, not overwritten
    xd.copy_(torch.from_numpy(x_ref).cuda())
    ks.solve(op, bd, xd, x_nonzero=True)
    assert ks.iterations <= 2
    ks.destroy(); op.destroy()


def test_fgmres_flexible_preconditioner(sp):
    """Right preconditioning through the Z basis: any non-singular M must give the same solution.
    M = the operator itself (solves A A y = b, x = A y)."""
    import torch
    dims = (9, 8)
    op = sp.EllipticOp(dims); pc = sp.EllipticOp(dims)
    n = op.global_size
    A = dense(lambda e: orc.elliptic_mult(dims, e, mode=orc.DIRECT), n)
    rng = np.random.default_rng(SEED + 1)
    b = rng.standard_normal(n)
    ks = sp.Fgmres(n, restart=n, rtol=1e-11, max_it=10 * n)
    bd = torch.from_numpy(b).cuda(); xd = torch.empty_like(bd)
    ks.solve(op, bd, xd, M=pc)
    x = xd.cpu().numpy()
    assert ks.reason == 2
    assert np.linalg.norm(b - A @ x) <= 1e-9 * np.linalg.norm(b)
    ks.destroy(); op.destroy(); pc.destroy()


def test_fgmres_max_it_and_restart(sp):
    import torch
    dims = (14, 14)
    op = sp.EllipticOp(dims)
    n = op.global_size
    bd = torch.randn(n, dtype=torch.float64, device="cuda"); xd = torch.empty_like(bd)
    ks = sp.Fgmres(n, restart=5, rtol=1e-14, max_it=12)
    ks.solve(op, bd, xd)
    assert ks.reason == -3 and ks.iterations == 12
    r = bd - op.mult(xd, torch.empty_like(bd))
    assert float(r.norm()) < float(bd.norm())          # restarted GMRES never increases the residual
    ks.destroy(); op.destroy()


@pytest.mark.parametrize("max_it,restart", [(4, 30), (5, 5), (1, 30), (7, 3)])
def test_fgmres_truncated_solve_is_the_krylov_minimiser(sp, max_it, restart):
    """A solve cut off by its iteration limit (the inner velocity solves of the Stokes preconditioners: -vel_ksp_max_it 4,
    README:43) returns the minimiser of |b - A x| over the Krylov space it built -- also when the limit falls inside a cycle,
    where the last basis vector is never normalised, and with a zero initial guess that is never written (x arrives as NaN).
    Reference: least squares over the same space in numpy, cycle by cycle."""
    import torch
    dims = (9, 8)
    op = sp.EllipticOp(dims)
    n = op.global_size
    A = dense(lambda e: orc.elliptic_mult(dims, e, mode=orc.DIRECT), n)
    b = np.random.default_rng(SEED + 7).standard_normal(n)
    x = np.zeros(n); left = max_it
    while left > 0:                                       # restarted GMRES: each cycle minimises over K_k(A, r)
        k = min(restart, left); r = b - A @ x
        K = np.empty((n, k)); v = r.copy()
        for j in range(k):
            K[:, j] = v / np.linalg.norm(v); v = A @ K[:, j]
        Q, _ = np.linalg.qr(K)
        y = np.linalg.lstsq(A @ Q, r, rcond=None)[0]
        x = x + Q @ y; left -= k
    ks = sp.Fgmres(n, restart=restart, rtol=1e-300, max_it=max_it)
    bd = torch.from_numpy(b).cuda(); xd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    ks.solve(op, bd, xd)
    assert ks.reason == -3 and ks.iterations == max_it
    xg = xd.cpu().numpy()
    assert np.linalg.norm(xg - x) <= 1e-9 * np.linalg.norm(x), np.linalg.norm(xg - x) / np.linalg.norm(x)
    assert abs(ks.residual - np.linalg.norm(b - A @ xg)) <= 1e-9 * np.linalg.norm(b)
    ks.destroy(); op.destroy()


def test_fgmres_zero_rhs_clears_x(sp):
    """b = 0 with a zero initial guess: the solve ends before any update, and x (never written on the way) must come back 0."""
    import torch
    op = sp.EllipticOp((9, 8))
    n = op.global_size
    ks = sp.Fgmres(n, restart=10, rtol=1e-8, max_it=50)
    bd = torch.zeros(n, dtype=torch.float64, device="cuda"); xd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    ks.solve(op, bd, xd)
    assert ks.iterations == 0 and ks.reason > 0
    assert float(xd.abs().max()) == 0.0
    ks.destroy(); op.destroy()


@pytest.mark.parametrize("dims", [(7, 6), (6, 5, 5)], ids=lambda d: "x".join(map(str, d)))
def test_stokes_schur_vs_dense(sp, dims):
    """StokesMatMultSchur (stokes.C:523-535): y = -PV VV^{-1} VP x with the inner solve driven to 1e-12."""
    import torch
    op = sp.StokesOp(dims)
    gv, gp = op.velocity_size, op.pressure_size
    VV = dense(lambda e: orc.stokes_mult_vv(dims, e), gv)
    VP = dense(lambda e: orc.stokes_mult_vp(dims, e), gp, gv)
    PV = dense(lambda e: orc.stokes_divergence(dims, e), gv, gp)
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal(gp)
    ref = -PV @ np.linalg.solve(VV, VP @ x)
    xd = torch.from_numpy(x).cuda(); yd = torch.empty_like(xd)
    op.mult_schur(xd, yd, restart=60, rtol=1e-12, max_it=5000)
    y = yd.cpu().numpy()
    assert op.inner_iterations > 0
    assert np.linalg.norm(y - ref) <= 1e-7 * np.linalg.norm(ref)
    # KSP defaults (rtol 1e-5): fewer inner applies, answer within the looser tolerance
    its_tight = op.inner_iterations
    op.mult_schur(xd, yd, restart=30, rtol=1e-5)
    assert op.inner_iterations <= its_tight
    assert np.linalg.norm(yd.cpu().numpy() - ref) <= 1e-2 * np.linalg.norm(ref)
    op.destroy()
