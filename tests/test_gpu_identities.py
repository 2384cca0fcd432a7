"""The HIP path held to the identities of tests/test_oracle_identities.py (which pin the results the reference holds no
fixture for, from the reference's own definitions alone -- no oracle involved here except where stated):

(a) ell_op_mult in the state ell_op_function(u) leaves is the derivative of ell_op_function at u (elliptic.C:319-323
    against :507-513), for gamma = 4 and exponents 2 / 2.5 / 3, on the general kernels (small, odd extents), on
    cheb_fused4_kernel (even extents of 66..256 points) and on its interior-line FormFunction path;
(b) stokes_op_mult in the state of a power-law stokes_op_function is the derivative of that residual (stokes.C:647-662
    against :710-725, :1930-1944), stokes_op_mult_vv with eta' != 0 included, on the general node loops and on the
    six-component storage of lines of more than 64 points;
(c) the power-law stokes_op_function at the ANALYTIC fields of tests/golden/analytic_powerlaw.npz (symbolic forcing,
    50 digits): residual = truncation error, decaying spectrally; eta, eta', strain equal their closed forms.
"""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from conftest import relerr
import test_oracle_identities as ident

pytestmark = pytest.mark.gpu
sp = ge.load()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def fd_errors(F, Jv, u, v, eps_pair):
    errs = []
    for e in eps_pair:
        dF = (F(u + e * v) - F(u - e * v)) / (2 * e)
        errs.append((torch.linalg.norm(dF - Jv) / torch.linalg.norm(Jv)).item())
    return errs


ELL_CASES = [((14, 12), 4.0, 2.0, True), ((14, 12), 4.0, 2.5, True), ((9, 8, 7), 4.0, 2.0, True), ((9, 8, 7), 4.0, 2.5, True),
             ((72, 68, 66), 4.0, 2.0, False),        # interior-line FormFunction + cheb_fused4_kernel Jacobian (homogeneous rows, exponent 2)
             ((72, 68, 66), 4.0, 2.0, True),         # the same shape through the gather pass (inhomogeneous rows)
             ((72, 68, 66), 4.0, 2.5, True), ((132, 70), 1.5, 3.0, True)]


@pytest.mark.parametrize("dims,gamma,exponent,inhomogeneous", ELL_CASES)
def test_elliptic_jacobian_is_the_derivative_of_the_residual(dims, gamma, exponent, inhomogeneous):
    g = torch.Generator(device="cuda").manual_seed(41)
    op = sp.EllipticOp(dims)
    if inhomogeneous:
        op.set_dirichlet(0.5 + np.random.default_rng(5).random(op.dirichlet_size))
    G = op.global_size
    u = 0.5 + torch.rand(G, dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn(G, dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn(G, dtype=torch.float64, device="cuda", generator=g)

    def F(w):
        r = torch.empty_like(w); op.function(w.contiguous(), b, r, gamma, exponent); return r.clone()
    F(u)                                                   # leaves eta, eta', grad u of u as the handle's state
    Jv = torch.empty_like(v); op.mult(v, Jv); Jv = Jv.clone()
    deta = op.get_state(1)
    assert np.abs(deta).max() > 0.1
    P = max(dims)
    # rounding of the quotient grows like P^4 eps_mach / e: larger steps on the long lines
    eps_pair = (2e-3, 1e-3) if P < 64 else (8e-3, 4e-3)
    # F(u +- e v) overwrite the state: restore it before nothing else is asked of the Jacobian (Jv was taken first)
    errs = fd_errors(F, Jv, u, v, eps_pair)
    assert errs[1] < (2e-5 if P < 64 else 4e-4), errs
    assert 3.0 < errs[0] / errs[1] < 5.0, errs
    op.destroy()


POWER = (1, 1.0, 3.0, 1e-2, 1.0)
ST_CASES = [((10, 9), POWER), ((8, 7, 6), POWER), ((8, 7, 6), (1, 1.3, 2.0, 1e-1, 0.7)), ((9, 8), (0, 1.0, 1.0, 1.0, 1.0)),
            ((68, 66, 66), POWER),                   # six-component stress storage (lines of more than 64 points, stokes_op::sym)
            ((40, 36, 34), POWER)]                   # node pairs, nine components


@pytest.mark.parametrize("dims,rheology", ST_CASES)
def test_stokes_jacobian_is_the_derivative_of_the_residual(dims, rheology):
    rng = np.random.default_rng(43)
    d = len(dims)
    op = sp.StokesOp(dims)
    op.set_rheology(*rheology)
    op.set_dirichlet(rng.standard_normal(op.dirichlet_size)); op.set_force(rng.standard_normal(op.global_size))
    x = dev(rng.standard_normal(op.global_size)); v = dev(rng.standard_normal(op.global_size))

    def F(w):
        r = torch.empty_like(w); op.function(w.contiguous(), r); return r.clone()
    F(x)
    Jv = torch.empty_like(v); op.mult(v, Jv); Jv = Jv.clone()
    vel = v.reshape(-1, d + 1).clone(); vel[:, d] = 0.0
    Jvel = torch.empty_like(v); op.mult(vel.reshape(-1).contiguous(), Jvel); Jvel = Jvel.reshape(-1, d + 1)[:, :d].reshape(-1).clone()
    vv = torch.empty(op.velocity_size, dtype=torch.float64, device="cuda")
    op.mult_vv(vel[:, :d].reshape(-1).contiguous(), vv)
    assert (torch.linalg.norm(vv - Jvel) / torch.linalg.norm(Jvel)).item() < 1e-12       # MatVV = velocity rows of the full apply on [v; 0]
    deta = op.get_state(1)
    P = max(dims)
    errs = fd_errors(F, Jv, x, v, (2e-4, 1e-4) if P < 64 else (2e-3, 1e-3))
    if rheology[0] == 0:
        assert errs[1] < 1e-8 and np.abs(deta).max() == 0.0
    else:
        assert np.abs(deta).max() > 0
        assert errs[1] < (2e-5 if P < 64 else 4e-4), errs
        assert 3.0 < errs[0] / errs[1] < 5.0, errs
    op.destroy()


def test_power_law_residual_at_the_analytic_field_decays_spectrally():
    rh = ident.pl_rheology()
    for family in ([c for c in ident.pl_cases() if len(c) == 2], [c for c in ident.pl_cases() if len(c) == 3]):
        res = []
        for dims in family:
            xG, fG, dvals, m = ident.pl_vectors(dims)
            op = sp.StokesOp(dims); op.set_rheology(*rh); op.set_dirichlet(dvals); op.set_force(fG)
            y = torch.empty(op.global_size, dtype=torch.float64, device="cuda")
            op.function(dev(xG), y); torch.cuda.synchronize()
            res.append(np.abs(y.cpu().numpy()).max() / np.abs(fG).max())
            op.destroy()
        assert res[1] < res[0] * 2e-3 and res[2] < res[1] * 5e-3 and res[2] < 2e-7, res


@pytest.mark.parametrize("dims", [(28, 26), (20, 18, 16)])
def test_power_law_state_equals_its_closed_form(dims):
    d = len(dims)
    tag = "pl_" + "x".join(map(str, dims))
    xG, fG, dvals, m = ident.pl_vectors(dims)
    op = sp.StokesOp(dims); op.set_rheology(*ident.pl_rheology()); op.set_dirichlet(dvals); op.set_force(fG)
    y = torch.empty(op.global_size, dtype=torch.float64, device="cuda")
    op.function(dev(xG), y); torch.cuda.synchronize()
    PL = ident.PL
    assert relerr(op.get_state(0), PL[tag + "_eta"].ravel()) < 1e-12
    assert relerr(op.get_state(1), PL[tag + "_deta"].ravel()) < 1e-12
    S = PL[tag + "_strain"].reshape(-1, d, d)
    for j in range(d):
        assert relerr(op.get_state(2 + j).reshape(-1, d), S[:, j, :]) < 1e-12
    op.destroy()
