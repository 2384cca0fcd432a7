"""CPU tests pinning the Stokes part of the oracle (stokes.C:499-758, 1029-1080, 1920-1944;
util.C:129-144) to the numpy/scipy golden vectors and to structural properties the reference
checks itself (constant pressure in the null space, stokes.C:190-212)."""
import numpy as np
import pytest

import oracle_lib as orc
from conftest import relerr, HERE
import os

POWER = (1, 1.0, 3.0, 1e-4, 1.0)   # README:52: -rheology 1 -exponent 3 -eps 1e-4


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(HERE, "golden", "stokes_golden.npz")))


@pytest.mark.parametrize("dims", [(8, 7), (7, 6, 5)])
@pytest.mark.parametrize("mode", [orc.DIRECT, orc.FAST])
def test_stokes_golden(g, dims, mode):
    tag = "x".join(str(v) for v in dims)
    d = len(dims)
    x = g["st_%s_x" % tag]
    X = x.reshape(-1, d + 1)
    vG, pG = np.ascontiguousarray(X[:, :d]).ravel(), np.ascontiguousarray(X[:, d])
    assert relerr(orc.stokes_mult_vv(dims, vG, mode=mode), g["st_%s_vv_lin" % tag]) < 1e-12
    assert relerr(orc.stokes_divergence(dims, vG, mode=mode), g["st_%s_pv" % tag]) < 1e-12
    assert relerr(orc.stokes_mult_vp(dims, pG, mode=mode), g["st_%s_vp" % tag]) < 1e-11
    assert relerr(orc.stokes_mult(dims, x, mode=mode), g["st_%s_mult_lin" % tag]) < 1e-11
    y, eta, deta, strain = orc.stokes_function(dims, g["st_%s_fn_x" % tag], g["st_%s_fn_dirichlet" % tag],
                                               g["st_%s_fn_force" % tag], POWER, mode=mode)
    assert relerr(eta, g["st_%s_fn_eta" % tag]) < 1e-12
    assert relerr(deta, g["st_%s_fn_deta" % tag]) < 1e-12
    assert relerr(strain, g["st_%s_fn_strain" % tag]) < 1e-12
    assert relerr(y, g["st_%s_fn_y" % tag]) < 1e-11
    assert relerr(orc.stokes_mult(dims, x, eta, deta, strain, mode=mode), g["st_%s_mult_nl" % tag]) < 1e-11
    U, U2, dv = orc.stokes_exact(dims, 1)
    assert relerr(U, g["st_%s_exact1_U" % tag]) < 1e-15
    r, *_ = orc.stokes_function(dims, U, dv, U2, mode=mode)
    assert np.abs(r - g["st_%s_exact1_residual" % tag]).max() < 1e-10


@pytest.mark.parametrize("dims", [(8, 7), (7, 6, 5)])
def test_pressure_reduce_golden(g, dims):
    tag = "x".join(str(v) for v in dims)
    d = len(dims)
    N, I = orc.sizes(dims)[:2]
    pG = np.ascontiguousarray(g["st_%s_x" % tag].reshape(-1, d + 1)[:, d])
    mask = np.ones(dims, bool)
    for ax, p in enumerate(dims):
        sl = [slice(None)] * d
        sl[ax] = [0, p - 1]
        mask[tuple(sl)] = False
    pL = np.zeros(dims)
    pL[mask] = pG
    assert relerr(orc.stokes_pressure_reduce(dims, pL), g["st_%s_preduce" % tag]) < 1e-13


def test_pressure_reduce_is_polynomial_extension():
    """A polynomial of degree <= P-3 per direction is reproduced exactly on the boundary."""
    dims = (9, 8, 7)
    grids = np.meshgrid(*[np.cos(np.arange(p) * np.pi / (p - 1)) for p in dims], indexing="ij")
    f = (1 + grids[0] + grids[0] ** 3) * (2 - grids[1] ** 2) * (1 + 0.5 * grids[2] ** 4)
    pL = f.copy()
    for ax, p in enumerate(dims):
        sl = [slice(None)] * 3
        sl[ax] = [0, p - 1]
        pL[tuple(sl)] = 0.0
    out = orc.stokes_pressure_reduce(dims, pL).reshape(dims)
    assert np.abs(out - f).max() < 1e-11


@pytest.mark.parametrize("dims", [(10, 9), (8, 7, 6)])
def test_constant_pressure_in_null_space(dims):
    """stokes.C:190-212 MatNullSpaceTest: A [0; const] = 0 (grad of the extended constant is 0)."""
    d = len(dims)
    I = orc.sizes(dims)[1]
    x = np.zeros((I, d + 1))
    x[:, d] = 1.0
    y = orc.stokes_mult(dims, x.ravel(), mode=orc.FAST)
    assert np.abs(y).max() < 1e-10


def test_exact2_residual_converges():
    """stokes.C:190-212 with Exact2 (README:43): residual of the exact solution falls spectrally."""
    errs = []
    for n in (8, 12, 16, 20):
        dims = (n, n)
        U, U2, dv = orc.stokes_exact(dims, 2)
        r, *_ = orc.stokes_function(dims, U, dv, U2, mode=orc.FAST)
        errs.append(np.abs(r).max())
    assert errs[1] < 1e-2 * errs[0] and errs[3] < 1e-7 * errs[0]
