#!/usr/bin/env python3
"""Generate the committed golden vectors in tests/golden/.

Independent of oracle/: the transforms come from scipy.fft (pocketfft) type-1
DCT/DST, which implement the same unnormalised definitions as FFTW's
REDFT00/RODFT00 (FFTW manual, "1d Real-even DFTs"/"1d Real-odd DFTs"); the
steps around them restate chebyshev.c:142-199 (ChebMult), elliptic.C:297-339
(MatMult_Elliptic) and elliptic.C:481-533 (FormFunction) in numpy.  The
reference itself (FFTW3 + PETSc 3.0) cannot be built in this image, so these
vectors pin the oracle and the HIP path to an independent evaluation of the
reference's formulas, not to a run of the reference binary.

Usage: python tests/golden/make_golden.py   (rewrites tests/golden/*.npz)
"""
import os
import numpy as np
import scipy.fft as sf

PI = 3.14159265358979323846  # chebyshev.h:10
HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20240229


def cheb_mult(x, tr):
    """chebyshev.c:142-199 on axis tr of a C-ordered array."""
    x = np.moveaxis(np.asarray(x, dtype=np.float64), tr, -1)
    P = x.shape[-1]
    n = P - 1
    N = float(n)
    work = sf.dct(x, type=1, axis=-1)                       # :157 REDFT00
    y = np.zeros_like(work)
    I = np.arange(1, n, dtype=np.float64)
    work[..., 1:n] *= I                                     # :171
    y0 = np.zeros(work.shape[:-1])
    yn = np.zeros(work.shape[:-1])
    s = 1.0
    for i in range(1, n):                                   # :168-175 (same order)
        y0 += float(i) * work[..., i]
        yn += s * float(i) * work[..., i]
        s = -s
    y[..., 0] = 0.5 * work[..., n] * N + y0 / n             # :176
    y[..., n] = yn / N + 0.5 * s * N * work[..., n]         # :177
    if n > 1:
        z = sf.dst(work[..., 1:n], type=1, axis=-1)         # :181 RODFT00 on n-1 points
        pin = PI / N
        y[..., 1:n] = z / (2 * n * np.sqrt(1.0 - np.cos(I * pin) ** 2))  # :190
    return np.ascontiguousarray(np.moveaxis(y, -1, tr))


def interior_mask(dims):
    m = np.ones(dims, dtype=bool)
    for ax, p in enumerate(dims):
        sl = [slice(None)] * len(dims)
        sl[ax] = 0
        m[tuple(sl)] = False
        sl[ax] = p - 1
        m[tuple(sl)] = False
    return m


def elliptic_mult(dims, U, eta, deta, gradu0):
    """elliptic.C:297-339."""
    d = len(dims)
    mask = interior_mask(dims)
    w0 = np.zeros(dims)
    w0[mask] = U                                            # :305-308 (dirichlet0 = 0)
    g = [cheb_mult(w0, k) for k in range(d)]                # :309-311
    f = [eta * g[k] + deta * w0 * gradu0[k] for k in range(d)]  # :319-323
    acc = np.zeros(dims)                                    # :330
    for k in range(d):
        acc += -1.0 * cheb_mult(f[k], k)                    # :331-334
    return acc[mask].copy()                                 # :336


def elliptic_function(dims, U, b, dirichlet, gamma, exponent):
    """elliptic.C:481-533."""
    d = len(dims)
    mask = interior_mask(dims)
    w0 = np.zeros(dims)
    w0[mask] = U
    w0[~mask] = dirichlet                                   # compact, row-major boundary order
    gradu = [cheb_mult(w0, k) for k in range(d)]
    eta = 1.0 + gamma * np.power(w0, exponent)              # :508
    deta = exponent * gamma * np.power(w0, exponent - 1.0)  # :509
    acc = np.zeros(dims)
    for k in range(d):
        acc += -1.0 * cheb_mult(eta * gradu[k], k)
    rhs = acc[mask] + -1.0 * b                              # :530
    return rhs, eta, deta, gradu


def exact2(dims):
    """elliptic.C:644-655 (-exact 2): u = prod x_j^(4+j), u2 = -laplacian."""
    d = len(dims)
    grids = np.meshgrid(*[np.cos(np.arange(p) * np.pi / (p - 1)) for p in dims], indexing="ij")
    v = np.ones(dims)
    w = np.zeros(dims)
    for j in range(d):
        v *= grids[j] ** (4 + j)
        z = np.ones(dims)
        for k in range(d):
            z *= (4 + k) * (3 + k) * grids[k] ** (2 + k) if k == j else grids[k] ** (4 + k)
        w -= z
    return v, w


def exp_field(dims):
    """cheb.c:66-93: u = sum_j exp(x_j)."""
    grids = np.meshgrid(*[np.cos(np.arange(p) * PI / (p - 1)) if p > 1 else np.zeros(1) for p in dims],
                        indexing="ij")
    return sum(np.exp(g) for g in grids)


def main():
    rng = np.random.default_rng(SEED)
    out = {}
    shapes = [(5,), (8,), (32,), (33,), (2,), (3,), (8, 7), (32, 32), (8, 7, 5), (17, 16, 15),
              (8, 7, 5, 3), (4, 6, 5, 3)]
    for shp in shapes:
        tag = "x".join(str(s) for s in shp)
        xr = rng.standard_normal(shp)
        out["cheb_%s_rand_in" % tag] = xr
        xe = exp_field(shp)
        out["cheb_%s_exp_in" % tag] = xe
        for tr in range(len(shp)):
            if shp[tr] < 2:
                continue
            out["cheb_%s_rand_tr%d" % (tag, tr)] = cheb_mult(xr, tr)
            out["cheb_%s_exp_tr%d" % (tag, tr)] = cheb_mult(xe, tr)
    xr = rng.standard_normal((33, 32, 31))
    out["cheb_33x32x31_rand_in"] = xr
    out["cheb_33x32x31_rand_tr1"] = cheb_mult(xr, 1)
    np.savez_compressed(os.path.join(HERE, "cheb_golden.npz"), **out)

    ell = {}
    for dims in [(8, 6), (32, 32), (9, 8, 7)]:
        tag = "x".join(str(s) for s in dims)
        d = len(dims)
        mask = interior_mask(dims)
        G = int(mask.sum())
        U = rng.standard_normal(G)
        ones, zeros = np.ones(dims), np.zeros(dims)
        ell["ell_%s_U" % tag] = U
        ell["ell_%s_mult_lin" % tag] = elliptic_mult(dims, U, ones, zeros, [zeros] * d)
        # nonlinear state from the -exact 2 field with gamma=4, exponent=2
        v, w = exact2(dims)
        rhs, eta, deta, gradu = elliptic_function(dims, v[mask], w[mask], v[~mask], 4.0, 2.0)
        ell["ell_%s_exact2_u" % tag] = v[mask]
        ell["ell_%s_exact2_b" % tag] = w[mask]
        ell["ell_%s_exact2_dirichlet" % tag] = v[~mask]
        ell["ell_%s_fn_rhs" % tag] = rhs
        ell["ell_%s_fn_eta" % tag] = eta.ravel()
        ell["ell_%s_fn_deta" % tag] = deta.ravel()
        ell["ell_%s_fn_gradu" % tag] = np.stack([g.ravel() for g in gradu])
        ell["ell_%s_mult_nl" % tag] = elliptic_mult(dims, U, eta, deta, gradu)
    np.savez_compressed(os.path.join(HERE, "elliptic_golden.npz"), **ell)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
