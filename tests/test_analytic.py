"""The oracle against ANALYTIC answers held by tests/golden/analytic_golden.npz (50-digit closed forms from
tests/golden/make_analytic.py, independent of oracle/ and of the product code): the full Chebyshev basis for
P in {32, 64, 128, 256}, the manufactured fields of elliptic.C:619-655 and stokes.C:1963-2012.

These are the known answers the reference's own tests use (cheb.c:66-112, elliptic.C:193-209, stokes.C:190-212),
extended from one smooth function per axis to a basis of the whole space, which pins every entry of the
operator.  The GPU twin of this file is tests/test_gpu_analytic.py."""
import os
import numpy as np
import pytest
import oracle_lib as orc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "analytic_golden.npz"))
TOL = 1e-10          # north_star: residuals within 1e-10 relative of the reference (normwise)


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def interior_mask(dims):
    m = np.ones(dims, dtype=bool)
    for ax, p in enumerate(dims):
        sl = [slice(None)] * len(dims)
        sl[ax] = 0; m[tuple(sl)] = False
        sl[ax] = p - 1; m[tuple(sl)] = False
    return m


def ell_cases():
    out = []
    for s in G["ell_cases"]:
        dm, ex, ga, e, cs = str(s).split("|")
        out.append((tuple(int(v) for v in dm.split("x")), int(ex), float(ga), float(e), float(cs)))
    return out


def st_cases():
    return [(tuple(int(v) for v in str(s).split("|")[0].split("x")), int(str(s).split("|")[1])) for s in G["stokes_cases"]]


def stokes_vectors(dims, exact):
    """Interleaved global vectors (stokes.C:499-519) and compact Dirichlet values cut from the fixture."""
    tag = "st_%s_e%d" % ("x".join(map(str, dims)), exact)
    V, Pp, F = G[tag + "_v"], G[tag + "_p"], G[tag + "_f"]
    m = interior_mask(dims)
    Ug = np.concatenate([V[m], Pp[m][:, None]], axis=1).ravel()
    Fg = np.concatenate([F[m], np.zeros((m.sum(), 1))], axis=1).ravel()
    return Ug, Fg, V[~m].ravel().copy()


@pytest.mark.parametrize("P", [32, 64, 128, 256])
@pytest.mark.parametrize("mode", [orc.DIRECT, orc.FAST])
def test_oracle_full_basis(P, mode):
    """ChebMult of T_k at the nodes is T_k' for EVERY k < P (columns of a P x P tensor, both axes)."""
    if mode == orc.DIRECT and P > 128:
        pytest.skip("O(P^2) long-double sums per line: covered by P <= 128")
    T, dT = G["basis_%d_T" % P], G["basis_%d_dT" % P]
    y0 = orc.cheb_mult(T, 0, mode=mode)                            # lines = columns (strided)
    y1 = orc.cheb_mult(np.ascontiguousarray(T.T), 1, mode=mode)    # lines = rows (contiguous)
    assert relerr(y0, dT) < TOL and relerr(y1, dT.T) < TOL
    # per basis function: error bounded by the conditioning of differentiation, c n^2 eps max|T_k|;
    # the FFT recipe of chebyshev.c:157-193 measures 440 n^2 eps at P = 256, 36 at P = 128
    n = P - 1
    assert np.abs(y0 - dT).max() < 2000 * n * n * 2.3e-16


def test_truth_path_matches_analytic_basis():
    """The long-double direct summation (the oracle's ground truth) reproduces the closed forms to rounding."""
    for P in (32, 64, 128):
        T, dT = G["basis_%d_T" % P], G["basis_%d_dT" % P]
        assert relerr(orc.cheb_mult_truth(T, 0), dT) < 1e-15


@pytest.mark.parametrize("case", ell_cases(), ids=lambda c: "%s-exact%d" % ("x".join(map(str, c[0])), c[1]))
def test_elliptic_exact_fields_and_residual(case):
    dims, exact, gamma, expo, cs = case
    tag = "ell_%s_e%d" % ("x".join(map(str, dims)), exact)
    U, F = G[tag + "_u"], G[tag + "_f"]
    m = interior_mask(dims)
    u, u2, dv = orc.elliptic_exact(dims, exact, gamma=gamma, exponent=expo, cos_scale=cs)
    assert np.abs(u - U[m]).max() < 1e-14 * max(1.0, np.abs(U).max())
    assert np.abs(dv - U[~m]).max() < 1e-14 * max(1.0, np.abs(U).max())
    if exact == 1 and len(dims) > 2:
        # the reference's -exact 1 forcing carries a factor 2 per extra dimension (elliptic.C:638-640: z *= 2 (1-x^2)
        # for EVERY k != j); the oracle restates the reference, the fixture holds the true -laplace(u)
        assert relerr(u2, 2.0 ** (len(dims) - 2) * F[m]) < 1e-13
        return
    assert relerr(u2, F[m]) < 1e-12
    # elliptic.C:193-209: FormFunction(u_exact) with b = forcing, printed as a norm.  Exact to rounding for the
    # polynomial fields (-exact 1, 2), spectrally small for the cosine field
    r = orc.elliptic_function(dims, U[m].copy(), F[m].copy(), U[~m].copy(), gamma, expo, mode=orc.FAST)[0]
    bound = 1e-9 if exact in (1, 2) else (1e-3 if min(dims) >= 24 else 5e-2)   # cos(1.5 pi x)^3 needs ~30 points per dim
    assert np.abs(r).max() <= bound * np.abs(F).max()


@pytest.mark.parametrize("case", st_cases(), ids=lambda c: "%s-Exact%d" % ("x".join(map(str, c[0])), c[1]))
def test_stokes_exact_fields_and_residual(case):
    dims, exact = case
    tag = "st_%s_e%d" % ("x".join(map(str, dims)), exact)
    assert np.abs(G[tag + "_div"]).max() < 1e-15                       # the manufactured velocity is solenoidal
    Ug, Fg, dvals = stokes_vectors(dims, exact)
    U, U2, dv = orc.stokes_exact(dims, exact)
    assert np.abs(U - Ug).max() < 1e-13 * max(1.0, np.abs(Ug).max())
    assert np.abs(dv - dvals).max() < 1e-14
    assert relerr(U2, Fg) < 1e-13
    # stokes.C:190-212: residual of the exact solution.  The pressure block is exact only up to the boundary
    # extrapolation of StokesPressureReduceOrder, hence spectrally small rather than rounding-small
    y = orc.stokes_function(dims, Ug, dvals, Fg, mode=orc.FAST)[0]
    assert np.abs(y).max() <= (5e-3 if min(dims) >= 12 else 5e-2) * np.abs(Fg).max()
