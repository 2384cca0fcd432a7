"""End-to-end solves through the C ABI, modelled on the reference's own test (tests.sh:10-17):
./elliptic -dim n,n -exact 0 -cos_scale 3 -gamma 4 -ksp_rtol 1e-12 -snes_rtol 1e-12 | grep 'Norm of error'
The analytic forcing of the separable-cosine solution (elliptic.C:619-632) is solved for by Newton +
matrix-free FGMRES on the device; the error against the analytic solution must fall spectrally with n, and
the discrete solution must agree with a dense Newton iteration on the oracle's operators."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
import oracle_lib as orc

pytestmark = pytest.mark.gpu
sp = ge.load()
GAMMA, EXPO, COS = 4.0, 2.0, 3.0


def gpu_solve(dims):
    from importlib import import_module
    solve = import_module(sp.__name__ + ".solve")
    op = sp.EllipticOp(dims)
    u, u2, dv = orc.elliptic_exact(dims, 0, gamma=GAMMA, exponent=EXPO, cos_scale=COS)
    op.set_dirichlet(dv)
    b = torch.from_numpy(u2).cuda()
    x = torch.zeros_like(b)                                             # VecSet(x, 0), elliptic.C:212
    its, kits, fn = solve.newton_krylov(sp, op, b, x, GAMMA, EXPO, snes_rtol=1e-12, ksp_rtol=1e-12,
                                        ksp_restart=min(256, op.global_size), ksp_max_it=20000)
    xs = x.cpu().numpy()
    op.destroy()
    return xs, u, its, kits, fn


def oracle_newton(dims, steps=40):
    """Dense Newton with the backtracking line search of solve.newton_krylov on the oracle's FormFunction /
    MatMult_Elliptic (small sizes only)."""
    u, u2, dv = orc.elliptic_exact(dims, 0, gamma=GAMMA, exponent=EXPO, cos_scale=COS)
    n = u.size
    x = np.zeros(n)
    F, eta, deta, gradu = orc.elliptic_function(dims, x, u2, dv, GAMMA, EXPO, mode=orc.DIRECT)
    for _ in range(steps):
        fn = np.linalg.norm(F)
        if fn < 1e-13 * np.linalg.norm(u2):
            break
        J = np.empty((n, n)); e = np.zeros(n)
        for j in range(n):
            e[j] = 1.0; J[:, j] = orc.elliptic_mult(dims, e, eta, deta, gradu, mode=orc.DIRECT); e[j] = 0.0
        dx = -np.linalg.solve(J, F); lam = 1.0
        while True:
            F, eta, deta, gradu = orc.elliptic_function(dims, x + lam * dx, u2, dv, GAMMA, EXPO, mode=orc.DIRECT)
            if np.linalg.norm(F) <= (1.0 - 1e-4 * lam) * fn or lam <= 1e-6:
                break
            lam *= 0.5
        x = x + lam * dx
    return x


def test_spectral_convergence_tests_sh():
    errs = {}
    for n in (8, 12, 16, 20, 24, 28):
        xs, u, its, kits, fn = gpu_solve((n, n))
        assert its <= 20 and np.isfinite(fn)
        errs[n] = np.abs(xs - u).max() / np.abs(u).max()
    # spectral decay once the cosine is resolved; a dense Newton iteration on the oracle's operators gives
    # 4.7e-1, 5.0e-2, 1.3e-3, 1.3e-5, 6.1e-8 for n = 12 .. 28
    assert errs[12] < 1.0 and errs[16] < 0.1 and errs[20] < 4e-3 and errs[24] < 4e-5 and errs[28] < 2e-7
    print("norm of error (tests.sh:10, cos_scale 3, gamma 4):", {k: "%.2e" % v for k, v in errs.items()})


@pytest.mark.parametrize("dims", [(10, 10), (7, 6, 5)], ids=lambda d: "x".join(map(str, d)))
def test_newton_matches_oracle_newton(dims):
    xs, u, its, kits, fn = gpu_solve(dims)
    xo = oracle_newton(dims)
    assert np.linalg.norm(xs - xo) <= 1e-9 * np.linalg.norm(xo)
