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


# ---------------------------------------------------------------------------------------------
# Stokes (stokes.C), -boundary 0 (all-Dirichlet velocity), numpy restatement
# ---------------------------------------------------------------------------------------------
def poly_interp(x, f, x0, x1):
    """util.C:129-144 (Neville): value at x0 and x1 of the polynomial through (x, f)."""
    n = len(x)
    a = np.array(f, dtype=np.float64)
    b = a.copy()
    for di in range(1, n):
        for i in range(n - di):
            a[i] = ((x0 - x[i + di]) * a[i] + (x[i] - x0) * a[i + 1]) / (x[i] - x[i + di])
            b[i] = ((x1 - x[i + di]) * b[i] + (x[i] - x1) * b[i + 1]) / (x[i] - x[i + di])
    return a[0], b[0]


def pressure_reduce(dims, pL):
    """stokes.C:1029-1080, same sweep order (z inside the i-loop, then y, then x)."""
    d = len(dims)
    m, n = dims[0], dims[1]
    p = 1 if d == 2 else dims[2]
    P = pL.reshape(m, n, p).copy()
    cx = np.cos(np.arange(m) * PI / (m - 1))
    cy = np.cos(np.arange(n) * PI / (n - 1))
    cz = np.cos(np.arange(p) * PI / (p - 1)) if p > 1 else np.zeros(1)
    for i in range(1, m):
        if p > 1:
            for j in range(1, n):
                P[i, j, 0], P[i, j, p - 1] = poly_interp(cz[1:p - 1], P[i, j, 1:p - 1], cz[0], cz[p - 1])
        for k in range(p):
            P[i, 0, k], P[i, n - 1, k] = poly_interp(cy[1:n - 1], P[i, 1:n - 1, k], cy[0], cy[n - 1])
    for j in range(n):
        for k in range(p):
            P[0, j, k], P[m - 1, j, k] = poly_interp(cx[1:m - 1], P[1:m - 1, j, k], cx[0], cx[m - 1])
    return P.reshape(pL.shape)


def stokes_vv(dims, eta, deta, Strain, vG):
    """stokes.C:623-676.  Local velocity arrays have shape dims + (d,)."""
    d = len(dims)
    mask = interior_mask(dims)
    xL = np.zeros(dims + (d,))
    xL[mask] = vG.reshape(-1, d)
    V = [cheb_mult(xL, i) for i in range(d)]
    strain = np.empty(dims + (d, d))
    for j in range(d):
        for k in range(d):
            strain[..., j, k] = 0.5 * (V[j][..., k] + V[k][..., j])
    z = np.zeros(dims)
    for j in range(d):
        for k in range(d):
            z += strain[..., j, k] * Strain[j][..., k]
    F = [np.empty(dims + (d,)) for _ in range(d)]
    for j in range(d):
        for k in range(d):
            F[j][..., k] = eta * strain[..., j, k] + deta * Strain[j][..., k] * z
    yL = np.zeros(dims + (d,))
    for i in range(d):
        yL += -1.0 * cheb_mult(F[i], i)
    return yL[mask].reshape(-1)


def stokes_div(dims, vG, dirichlet=None):
    """stokes.C:570-595."""
    d = len(dims)
    mask = interior_mask(dims)
    xL = np.zeros(dims + (d,))
    xL[mask] = vG.reshape(-1, d)
    if dirichlet is not None:
        xL[~mask] = dirichlet.reshape(-1, d)
    acc = np.zeros(dims)
    for i in range(d):
        acc += 1.0 * cheb_mult(np.ascontiguousarray(xL[..., i]), i)
    return acc[mask].copy()


def stokes_vp(dims, pG):
    """stokes.C:599-619."""
    d = len(dims)
    mask = interior_mask(dims)
    pL = np.zeros(dims)
    pL[mask] = pG
    pL = pressure_reduce(dims, pL)
    vL = np.zeros(dims + (d,))
    for i in range(d):
        vL[..., i] = cheb_mult(pL, i)
    return vL[mask].reshape(-1), pL


def stokes_mult(dims, eta, deta, Strain, xG):
    """stokes.C:499-519; full global vector = [v_0..v_{d-1}, p] per interior node."""
    d = len(dims)
    X = xG.reshape(-1, d + 1)
    vG0, pG0 = np.ascontiguousarray(X[:, :d]).reshape(-1), np.ascontiguousarray(X[:, d])
    vG1 = stokes_vv(dims, eta, deta, Strain, vG0)
    pG1 = stokes_div(dims, vG0)
    vG1 = vG1 + 1.0 * stokes_vp(dims, pG0)[0]
    return np.concatenate([vG1.reshape(-1, d), pG1[:, None]], axis=1).reshape(-1)


def rheology(kind, gamma, hardness=1.0, exponent=1.0, eps=1.0, gamma0=1.0):
    """stokes.C:1920-1944."""
    if kind == 0:
        return np.ones_like(gamma), np.zeros_like(gamma)
    n = exponent
    p = (1.0 - n) / (2.0 * n)
    eta = hardness * np.power(eps + gamma / gamma0, p)
    deta = hardness * p / gamma0 * np.power(eps + gamma / gamma0, p - 1.0)
    return eta, deta


def stokes_function(dims, xG, dirichlet, force, rh):
    """stokes.C:680-758."""
    d = len(dims)
    mask = interior_mask(dims)
    X = xG.reshape(-1, d + 1)
    vG0, pG0 = np.ascontiguousarray(X[:, :d]).reshape(-1), np.ascontiguousarray(X[:, d])
    xL = np.zeros(dims + (d,))
    xL[mask] = vG0.reshape(-1, d)
    xL[~mask] = dirichlet.reshape(-1, d)
    G = [cheb_mult(xL, i) for i in range(d)]
    s = np.empty(dims + (d, d))
    gamma = np.zeros(dims)
    for j in range(d):
        for k in range(d):
            s[..., j, k] = 0.5 * (G[j][..., k] + G[k][..., j])
            gamma += 0.5 * s[..., j, k] ** 2
    eta, deta = rheology(*rh, gamma=gamma) if False else rheology(rh[0], gamma, *rh[1:])
    yL = np.zeros(dims + (d,))
    for i in range(d):
        yL += -1.0 * cheb_mult(np.ascontiguousarray(eta[..., None] * s[..., i, :]), i)
    vG1 = yL[mask].reshape(-1)
    pG1 = stokes_div(dims, vG0, dirichlet)
    vG1 = vG1 + 1.0 * stokes_vp(dims, pG0)[0]
    y = np.concatenate([vG1.reshape(-1, d), pG1[:, None]], axis=1).reshape(-1)
    y = y + -1.0 * force
    strain = [np.ascontiguousarray(s[..., j, :]) for j in range(d)]
    return y, eta, deta, strain


def stokes_exact(dims, exact):
    """stokes.C:1963-2012 via StokesCreateExactSolution (:942-1003); 3-D Exact2 pressure pinned to 0."""
    d = len(dims)
    grids = np.meshgrid(*[np.cos(np.arange(p) * PI / (p - 1)) for p in dims], indexing="ij")
    x, y = grids[0], grids[1]
    u = np.sin(0.5 * PI * x) * np.cos(0.5 * PI * y)
    v = -np.cos(0.5 * PI * x) * np.sin(0.5 * PI * y)
    p = 0.25 * (np.cos(PI * x) + np.cos(PI * y)) + 10 * (x + y) if exact == 1 else np.zeros(dims)
    val = np.zeros(dims + (d + 1,))
    rhs = np.zeros(dims + (d + 1,))
    val[..., 0], val[..., 1], val[..., d] = u, v, p
    rhs[..., 0] = (0.5 * PI) ** 2 * u
    rhs[..., 1] = (0.5 * PI) ** 2 * v
    if exact == 1:
        rhs[..., 0] += -0.25 * PI * np.sin(PI * x) + 10
        rhs[..., 1] += -0.25 * PI * np.sin(PI * y) + 10
    mask = interior_mask(dims)
    return val[mask].reshape(-1), rhs[mask].reshape(-1), val[~mask][:, :d].reshape(-1)


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
    st = {}
    for dims in [(8, 7), (7, 6, 5)]:
        tag = "x".join(str(v) for v in dims)
        d = len(dims)
        mask = interior_mask(dims)
        I = int(mask.sum())
        xG = rng.standard_normal(I * (d + 1))
        ones, zeros = np.ones(dims), np.zeros(dims)
        zS = [np.zeros(dims + (d,)) for _ in range(d)]
        st["st_%s_x" % tag] = xG
        st["st_%s_mult_lin" % tag] = stokes_mult(dims, ones, zeros, zS, xG)
        X = xG.reshape(-1, d + 1)
        vG, pG = np.ascontiguousarray(X[:, :d]).reshape(-1), np.ascontiguousarray(X[:, d])
        st["st_%s_vv_lin" % tag] = stokes_vv(dims, ones, zeros, zS, vG)
        st["st_%s_pv" % tag] = stokes_div(dims, vG)
        vp, pL = stokes_vp(dims, pG)
        st["st_%s_vp" % tag] = vp
        st["st_%s_preduce" % tag] = pL.ravel()
        # power-law state (README:52: -exponent 3 -eps 1e-4) from the Exact1 field plus noise
        U, U2, dv = stokes_exact(dims, 1)
        Un = U + 0.05 * rng.standard_normal(U.shape)
        rh = (1, 1.0, 3.0, 1e-4, 1.0)
        y, eta, deta, strain = stokes_function(dims, Un, dv, U2, rh)
        st["st_%s_fn_x" % tag] = Un
        st["st_%s_fn_force" % tag] = U2
        st["st_%s_fn_dirichlet" % tag] = dv
        st["st_%s_fn_y" % tag] = y
        st["st_%s_fn_eta" % tag] = eta.ravel()
        st["st_%s_fn_deta" % tag] = deta.ravel()
        st["st_%s_fn_strain" % tag] = np.stack([a.ravel() for a in strain])
        st["st_%s_mult_nl" % tag] = stokes_mult(dims, eta, deta, strain, xG)
        ylin, *_ = stokes_function(dims, U, dv, U2, (0,))
        st["st_%s_exact1_U" % tag] = U
        st["st_%s_exact1_residual" % tag] = ylin
    np.savez_compressed(os.path.join(HERE, "stokes_golden.npz"), **st)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
