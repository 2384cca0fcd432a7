#!/usr/bin/env python3
"""Generate tests/golden/analytic_powerlaw.npz: the power-law Stokes residual at ANALYTIC fields, 50-digit arithmetic.

The reference never checks its power-law rheology (stokes.C:1930-1944) or the eta' terms of its Jacobians directly
(its own known-answer runs use the linear rheology, stokes.C:190-212 with README:43).  This script derives what
StokesFunction (stokes.C:680-758) must return for a smooth velocity / pressure field from the continuous equations
alone -- symbolic differentiation, no transform, no differentiation matrix, nothing from oracle/ or the product:

    s      = (grad v + grad v^T) / 2                      stokes.C:711-716
    gamma  = 1/2 sum_jk s_jk^2                             stokes.C:717
    eta    = B (eps + gamma/gamma0)^p,  p = (1-n)/(2n)     stokes.C:1933-1939
    eta'   = d eta / d gamma                               stokes.C:1940-1942   (symbolic derivative of the line above)
    f_k    = -sum_j d_j (eta s_jk) + d_k pr                stokes.C:737-747 (stress divergence, pressure gradient)
    f_p    = div v                                         stokes.C:746

With `force` = (f_k, f_p) and the Dirichlet values of v, the discrete residual at the nodal values of (v, pr) is the
truncation error of the collocation scheme: it decays spectrally with the resolution, and eta, eta', s at the nodes
must match the discrete state to the same accuracy.  Arrays are on the FULL local grid, row-major.

Usage: python tests/golden/make_analytic_powerlaw.py        (rewrites tests/golden/analytic_powerlaw.npz)
"""
import itertools
import os
import numpy as np
import mpmath as mp
import sympy as sy

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 50

# rheology parameters of the fixture: (hardness B, exponent n, regularisation eps, gamma0)
RHEO = (sy.Rational(13, 10), sy.Integer(3), sy.Rational(1, 2), sy.Rational(7, 10))      # eps = 1/2 keeps the branch point of eta(gamma) away from the fields
CASES = [(12, 10), (20, 18), (28, 26), (8, 7, 6), (14, 12, 10), (20, 18, 16)]


def nodes(P):
    return [mp.cos(mp.pi * i / (P - 1)) for i in range(P)]


def on_grid(dims, exprs, syms):
    fns = [sy.lambdify(syms, e, modules="mpmath") for e in exprs]
    xs = [nodes(p) for p in dims]
    out = [np.empty(dims) for _ in exprs]
    for idx in itertools.product(*[range(p) for p in dims]):
        pt = [xs[a][i] for a, i in enumerate(idx)]
        for o, f in zip(out, fns):
            o[idx] = float(f(*pt))
    return out


def fields(d):
    X = sy.symbols("x0:%d" % d)
    h = sy.pi / 2
    a = sy.Rational(1, 4)          # a background shear plus a quarter-amplitude smooth field: gamma stays within [0.7, 1.6], so that
    if d == 2:                     # eta(gamma(x)) is resolved to rounding on 28 x 26 / to 2e-7 on 20 x 18 x 16 (eta varies by 17 %)
        v = [2 * X[1] + a * (sy.sin(h * X[0]) * sy.cos(h * X[1]) + X[1] ** 2 / 3),
             -X[0] / 2 + a * (-sy.cos(h * X[0]) * sy.sin(h * X[1]) + X[0] * X[1] / 2)]
        pr = sy.cos(sy.pi * X[0]) / 4 + sy.sin(X[1]) + X[0] * X[1]
    else:
        v = [2 * X[1] + a * (sy.sin(h * X[0]) * sy.cos(h * X[1]) * sy.cos(X[2]) + X[1] * X[2] / 3),
             X[2] + a * (-sy.cos(h * X[0]) * sy.sin(h * X[1]) + X[2] ** 2 / 4),
             -X[0] / 2 + a * (sy.sin(X[0] + X[1] / 2) * sy.cos(h * X[2]) + X[0] * X[1] / 5)]
        pr = sy.cos(sy.pi * X[0]) / 4 + sy.sin(X[1]) * sy.cos(X[2]) + X[0] * X[2]
    B, n, eps, g0 = RHEO
    s = [[(sy.diff(v[k], X[j]) + sy.diff(v[j], X[k])) / 2 for k in range(d)] for j in range(d)]
    gamma = sum(s[j][k] ** 2 for j in range(d) for k in range(d)) / 2
    g = sy.Symbol("g", positive=True)
    p = (1 - n) / (2 * n)
    eta_g = B * (eps + g / g0) ** p
    eta = eta_g.subs(g, gamma)
    deta = sy.diff(eta_g, g).subs(g, gamma)
    f = [-sum(sy.diff(eta * s[j][k], X[j]) for j in range(d)) + sy.diff(pr, X[k]) for k in range(d)]
    div = sum(sy.diff(v[k], X[k]) for k in range(d))
    strain = [s[j][k] for j in range(d) for k in range(d)]
    return X, v, pr, f, div, eta, deta, strain


def main():
    out = {"rheology": np.array([float(r) for r in RHEO]),
           "cases": np.array(["x".join(map(str, dm)) for dm in CASES])}
    for dm in CASES:
        d = len(dm)
        X, v, pr, f, div, eta, deta, strain = fields(d)
        arrs = on_grid(dm, v + [pr] + f + [div, eta, deta] + strain, X)
        tag = "pl_" + "x".join(map(str, dm))
        out[tag + "_v"] = np.stack(arrs[:d], axis=-1)                       # interleaved components (stokes.C:284-290)
        out[tag + "_p"] = arrs[d]
        out[tag + "_f"] = np.stack(arrs[d + 1:2 * d + 1], axis=-1)
        out[tag + "_div"] = arrs[2 * d + 1]
        out[tag + "_eta"] = arrs[2 * d + 2]
        out[tag + "_deta"] = arrs[2 * d + 3]
        out[tag + "_strain"] = np.stack(arrs[2 * d + 4:], axis=-1).reshape(dm + (d, d))      # [node][j][k]
        print(tag, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "analytic_powerlaw.npz"), **out)
    print("wrote analytic_powerlaw.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
