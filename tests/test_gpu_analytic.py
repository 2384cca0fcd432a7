"""The HIP path against the ANALYTIC fixtures of tests/golden/analytic_golden.npz (see test_analytic.py for the CPU
twin and make_analytic.py for how they are made): the full Chebyshev basis through cheb_apply on both kernel
tilings, and the manufactured elliptic / Stokes fields through the operator callbacks."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
import oracle_lib as orc
from conftest import relerr
from test_analytic import G, ell_cases, st_cases, stokes_vectors, interior_mask

pytestmark = pytest.mark.gpu
sp = ge.load()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def gpu_cheb(x, tr):
    plan = sp.ChebPlan(x.shape, tr)
    yd = torch.full(x.shape, float("nan"), dtype=torch.float64, device="cuda")
    plan.mult(dev(x), yd)
    torch.cuda.synchronize()
    plan.destroy()
    return yd.cpu().numpy()


@pytest.mark.parametrize("P", [32, 64, 128, 256])
def test_hip_full_basis(P):
    """cheb_apply of T_k is T_k' for every k < P: strided lines (COLFAST tiling) and contiguous lines (JFAST)."""
    T, dT = G["basis_%d_T" % P], G["basis_%d_dT" % P]
    y0 = gpu_cheb(T, 0)
    y1 = gpu_cheb(np.ascontiguousarray(T.T), 1)
    # the dense product is accurate to a few ulp of the row sums of |D| |x|: observed 1e-15 normwise at P = 256,
    # two orders below the FFT recipe of the reference (1.3e-13, test_analytic.py)
    assert relerr(y0, dT) < 1e-13 and relerr(y1, dT.T) < 1e-13
    n = P - 1
    assert np.abs(y0 - dT).max() < 64 * n * n * 2.3e-16
    # a 3-D tensor whose lines along the middle axis are the basis functions (the layout of the operator sweeps)
    x3 = np.ascontiguousarray(np.broadcast_to(T[None, :, :, None], (3, P, P, 2)).transpose(0, 1, 3, 2).reshape(3, P, 2 * P))
    e3 = np.ascontiguousarray(np.broadcast_to(dT[None, :, :, None], (3, P, P, 2)).transpose(0, 1, 3, 2).reshape(3, P, 2 * P))
    assert relerr(gpu_cheb(x3, 1), e3) < 1e-13


@pytest.mark.parametrize("case", ell_cases(), ids=lambda c: "%s-exact%d" % ("x".join(map(str, c[0])), c[1]))
def test_hip_elliptic_exact_residual(case):
    """elliptic.C:193-209 on the device: FormFunction at the analytic field with the analytic forcing."""
    dims, exact, gamma, expo, cs = case
    tag = "ell_%s_e%d" % ("x".join(map(str, dims)), exact)
    U, F = G[tag + "_u"], G[tag + "_f"]
    m = interior_mask(dims)
    op = sp.EllipticOp(dims)
    op.set_dirichlet(U[~m].copy())
    r = op.function_host(U[m].copy(), F[m].copy(), gamma, expo)
    ro = orc.elliptic_function(dims, U[m].copy(), F[m].copy(), U[~m].copy(), gamma, expo, mode=orc.FAST)[0]
    op.destroy()
    if exact in (1, 2):        # polynomial fields: exact to rounding (dim >= degree + 2)
        assert np.abs(r).max() <= 1e-9 * np.abs(F).max()
    else:                      # cosine field: the discretisation error itself, identical on both paths
        assert np.abs(r).max() <= (1e-3 if min(dims) >= 24 else 5e-2) * np.abs(F).max()
        assert np.abs(r - ro).max() <= 1e-9 * np.abs(F).max()


@pytest.mark.parametrize("case", st_cases(), ids=lambda c: "%s-Exact%d" % ("x".join(map(str, c[0])), c[1]))
def test_hip_stokes_exact_residual(case):
    """stokes.C:190-212 on the device: StokesFunction at the analytic solution with the analytic forcing."""
    dims, exact = case
    Ug, Fg, dvals = stokes_vectors(dims, exact)
    st = sp.StokesOp(dims)
    st.set_dirichlet(dvals); st.set_force(Fg)
    y = torch.full((st.global_size,), float("nan"), dtype=torch.float64, device="cuda")
    st.function(dev(Ug), y)
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    yo = orc.stokes_function(dims, Ug, dvals, Fg, mode=orc.FAST)[0]
    st.destroy()
    assert np.abs(y).max() <= (5e-3 if min(dims) >= 12 else 5e-2) * np.abs(Fg).max()
    assert np.abs(y - yo).max() <= 1e-9 * np.abs(Fg).max()
