#!/usr/bin/env python3
"""Generate tests/golden/analytic_golden.npz: ANALYTIC answers in 50-digit arithmetic (mpmath / sympy).

What the reference's own tests hold for this path are analytic known answers, not numbers
(SURVEY 8c): cheb.c:66-112 (d/dx of a known function per axis), elliptic.C:193-209 with the
manufactured fields of elliptic.C:619-655 (-exact 0/1/2), stokes.C:190-212 with stokes.C:1963-2012
(Exact1 / Exact2).  This script evaluates such answers independently of oracle/ and of the product
code -- closed forms and symbolic differentiation only, no transform, no differentiation matrix:

1. Full Chebyshev basis.  For P in {32, 64, 128, 256} and EVERY k < P: T_k at the Gauss-Lobatto
   nodes x_i = cos(i pi/n), n = P-1 (chebyshev.c:186-190, elliptic.C:279), and its exact derivative
       T_k'(x_i) = k sin(k theta_i) / sin(theta_i),  T_k'(+1) = k^2,  T_k'(-1) = (-1)^(k+1) k^2.
   The interpolant of T_k (k <= n) is T_k itself, so ChebMult must return T_k' to rounding; the P
   columns form a basis, so together they pin all P^2 entries of the operator ChebMult applies.
2. Elliptic manufactured fields: u and f = -div((1 + gamma u^e) grad u) by symbolic differentiation
   of u (NOT the hand-derived expressions of elliptic.C:619-655, which the oracle restates).
3. Stokes Exact1 / Exact2: velocity, pressure and f_k = -sum_j d_j s_jk + d_k p with
   s = (grad v + grad v^T)/2 (the operator of stokes.C:623-676, 599-619 with eta = 1), div v.

Arrays are stored on the FULL local grid, row-major, last dimension fastest (the reference's local
layout); tests cut interior / boundary pieces themselves.

Usage: python tests/golden/make_analytic.py      (rewrites tests/golden/analytic_golden.npz, ~1.5 MB)
"""
import itertools
import os
import numpy as np
import mpmath as mp
import sympy as sy

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 50


def basis(P):
    n = P - 1
    C = np.empty((P, P)); E = np.empty((P, P))
    for i in range(P):
        th = mp.pi * i / n
        s = mp.sin(th)
        for k in range(P):
            C[i, k] = float(mp.cos(k * th))
            if i == 0:
                E[i, k] = float(k * k)
            elif i == n:
                E[i, k] = float((-1) ** (k + 1) * k * k)
            else:
                E[i, k] = float(k * mp.sin(k * th) / s)
    return C, E


def nodes(P):
    return [mp.cos(mp.pi * i / (P - 1)) for i in range(P)]


def on_grid(dims, exprs, syms):
    """Evaluate sympy expressions on the tensor grid of Gauss-Lobatto nodes, 50 digits, rounded once."""
    fns = [sy.lambdify(syms, e, modules="mpmath") for e in exprs]
    xs = [nodes(p) for p in dims]
    out = [np.empty(dims) for _ in exprs]
    for idx in itertools.product(*[range(p) for p in dims]):
        pt = [xs[a][i] for a, i in enumerate(idx)]
        for o, f in zip(out, fns):
            o[idx] = float(f(*pt))
    return out


def elliptic_case(dims, exact, gamma=0, exponent=2, cos_scale=1):
    d = len(dims)
    X = sy.symbols("x0:%d" % d)
    if exact == 0:
        s = sy.Rational(1, 2) * sy.nsimplify(cos_scale)
        u = sy.prod([sy.cos(s * sy.pi * x) for x in X])                       # elliptic.C:621-622
    elif exact == 1:
        u = sy.prod([(1 - x) * (1 + x) for x in X])                           # :635
    else:
        u = sy.prod([x ** (4 + j) for j, x in enumerate(X)])                  # :646
    eta = 1 + sy.nsimplify(gamma) * u ** sy.nsimplify(exponent)               # :508
    f = -sum(sy.diff(eta * sy.diff(u, x), x) for x in X)
    return on_grid(dims, [u, f], X)


def stokes_case(dims, exact):
    d = len(dims)
    X = sy.symbols("x0:%d" % d)
    h = sy.pi / 2
    v = [sy.sin(h * X[0]) * sy.cos(h * X[1]), -sy.cos(h * X[0]) * sy.sin(h * X[1])] + [sy.Integer(0)] * (d - 2)   # stokes.C:1970-71
    p = (sy.Rational(1, 4) * (sy.cos(sy.pi * X[0]) + sy.cos(sy.pi * X[1])) + 10 * (X[0] + X[1])) if exact == 1 else sy.Integer(0)
    strain = [[(sy.diff(v[k], X[j]) + sy.diff(v[j], X[k])) / 2 for k in range(d)] for j in range(d)]
    f = [-sum(sy.diff(strain[j][k], X[j]) for j in range(d)) + sy.diff(p, X[k]) for k in range(d)]
    div = sum(sy.diff(v[k], X[k]) for k in range(d))
    return on_grid(dims, v + [p] + f + [div], X)


def main():
    out = {}
    for P in (32, 64, 128, 256):
        C, E = basis(P)
        out["basis_%d_T" % P] = C
        out["basis_%d_dT" % P] = E
    ell = [((16, 14), 1, 0, 2, 1), ((8, 7, 6), 1, 0, 2, 1), ((10, 9), 2, 0, 2, 1), ((12, 11, 10), 2, 0, 2, 1),
           ((36, 32), 0, 4, 2, 3), ((22, 20, 18), 0, 4, 2, 3), ((24, 24), 0, 0.5, 3, 2)]
    out["ell_cases"] = np.array(["%s|%d|%g|%g|%g" % ("x".join(map(str, dm)), ex, ga, e, cs) for dm, ex, ga, e, cs in ell])
    for dm, ex, ga, e, cs in ell:
        u, f = elliptic_case(dm, ex, ga, e, cs)
        tag = "ell_%s_e%d" % ("x".join(map(str, dm)), ex)
        out[tag + "_u"] = u; out[tag + "_f"] = f
    st = [((14, 12), 1), ((14, 12), 2), ((9, 8, 7), 1), ((9, 8, 7), 2)]
    out["stokes_cases"] = np.array(["%s|%d" % ("x".join(map(str, dm)), ex) for dm, ex in st])
    for dm, ex in st:
        d = len(dm)
        arrs = stokes_case(dm, ex)
        tag = "st_%s_e%d" % ("x".join(map(str, dm)), ex)
        out[tag + "_v"] = np.stack(arrs[:d], axis=-1)            # interleaved components, as workV (stokes.C:284-290)
        out[tag + "_p"] = arrs[d]
        out[tag + "_f"] = np.stack(arrs[d + 1:2 * d + 1], axis=-1)
        out[tag + "_div"] = arrs[2 * d + 1]
    np.savez_compressed(os.path.join(HERE, "analytic_golden.npz"), **out)
    print("wrote analytic_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
