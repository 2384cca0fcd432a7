"""The CPU oracle against the reference's transform recipe evaluated with torch's CPU FFT (tests/fft_recipe.py) -- an
implementation of the same FFTW definitions that shares no code with the oracle's own transforms.  Together with
tests/test_gpu_fft_route.py (HIP path vs the same recipe on the device) this closes the triangle oracle = recipe = HIP on
random inputs of arbitrary shape, where the golden vectors pin fixed cases only.  No GPU."""
import numpy as np
import pytest
import torch

import fft_recipe as fr
import oracle_lib as orc

SEED = 20240229


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("shape", [(9,), (33, 20), (32, 31, 30), (6, 5, 4, 3), (64, 64, 40), (129, 12), (10, 256)], ids=lambda s: "x".join(map(str, s)))
def test_oracle_chebmult_equals_the_fft_recipe(shape):
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal(shape)
    for tr in range(len(shape)):
        ref = fr.cheb_fft(torch.from_numpy(x), tr).numpy()
        assert rel(orc.cheb_mult(x, tr, orc.FAST), ref) < 1e-11
        if np.prod(shape) <= 40000:
            assert rel(orc.cheb_mult_truth(x, tr), ref) < 1e-11          # the long-double direct summation


@pytest.mark.parametrize("dims,exponent", [((20, 18), 2.0), ((14, 12, 10), 3.0), ((40, 33, 20), 2.5), ((66, 68), 2.0)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else "e%g" % v)
def test_oracle_elliptic_equals_the_fft_recipe(dims, exponent):
    gamma = 1.5
    rng = np.random.default_rng(SEED + 1)
    full = torch.from_numpy(rng.random(dims) + 0.5)
    dv = full[fr.boundary_mask(dims, "cpu")].numpy()
    u = fr.interior(full).reshape(-1).contiguous()
    b = torch.from_numpy(rng.standard_normal(u.numel()))
    U = torch.from_numpy(rng.standard_normal(u.numel()))
    assert rel(orc.elliptic_mult(dims, U.numpy(), mode=orc.FAST), fr.poisson_ref(dims, U).numpy()) < 1e-10
    ref, eta, deta, grads = fr.elliptic_function_ref(dims, full, b, gamma, exponent)
    rhs_o, eta_o, deta_o, gradu_o = orc.elliptic_function(dims, u.numpy(), b.numpy(), dv, gamma, exponent, mode=orc.FAST)
    assert rel(rhs_o, ref.numpy()) < 1e-10
    assert rel(eta_o, eta.numpy()) < 1e-13 and rel(deta_o, deta.numpy()) < 1e-13
    for k in range(len(dims)):
        assert rel(gradu_o[k], grads[k].numpy()) < 1e-10
    ref = fr.elliptic_jacobian_ref(dims, U, eta, deta, grads)
    assert rel(orc.elliptic_mult(dims, U.numpy(), eta_o, deta_o, gradu_o, mode=orc.FAST), ref.numpy()) < 1e-10


@pytest.mark.parametrize("dims", [(16, 14), (12, 11, 10), (24, 20, 18)], ids=lambda s: "x".join(map(str, s)))
def test_oracle_stokes_equals_the_fft_recipe(dims):
    """Linear blocks and the power-law Function / Newton-linearised MatMult of the oracle (which follows stokes.C line by line,
    Neville table included) against the recipe (formulas, Lagrange end-point weights)."""
    d = len(dims)
    N, I, gv, gp_n, g, ndv = orc.stokes_sizes(dims)
    rng = np.random.default_rng(SEED + 2)
    x = torch.from_numpy(rng.standard_normal(g))
    s, gp, div = fr.stokes_fields(dims, x)
    full = fr.stokes_assemble(dims, s, gp, div)
    assert rel(orc.stokes_mult(dims, x.numpy(), mode=orc.FAST), full.numpy()) < 1e-9
    X = x.view(*[n - 2 for n in dims], d + 1)
    v_in = X[..., :d].reshape(-1).contiguous().numpy(); p_in = X[..., d].reshape(-1).contiguous().numpy()
    assert rel(orc.stokes_mult_vv(dims, v_in, mode=orc.FAST), fr.stokes_assemble(dims, s, None, div)[..., :d].numpy()) < 1e-10
    assert rel(orc.stokes_divergence(dims, v_in, mode=orc.FAST), div.numpy()) < 1e-10
    assert rel(orc.stokes_mult_vp(dims, p_in, mode=orc.FAST), torch.stack(gp, dim=-1).numpy()) < 1e-9
    rheo = (1.0, 3.0, 1e-4, 1.0)
    yo, eta_o, deta_o, strain_o = orc.stokes_function(dims, x.numpy(), np.zeros(ndv), np.zeros(g), rheology=(1,) + rheo, mode=orc.FAST)
    eta, deta = fr.power_law(s, *rheo)
    ref = fr.stokes_assemble(dims, [[eta * s[j][k] for k in range(d)] for j in range(d)], gp, div)
    assert rel(yo, ref.numpy()) < 1e-9
    z = torch.from_numpy(rng.standard_normal(g))
    s1, gp1, div1 = fr.stokes_fields(dims, z)
    zz = sum(s1[j][k] * s[j][k] for j in range(d) for k in range(d))
    ref = fr.stokes_assemble(dims, [[eta * s1[j][k] + deta * s[j][k] * zz for k in range(d)] for j in range(d)], gp1, div1)
    assert rel(orc.stokes_mult(dims, z.numpy(), eta_o, deta_o, strain_o, mode=orc.FAST), ref.numpy()) < 1e-9
