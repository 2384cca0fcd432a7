"""cheb_apply against the reference's transform recipe run on the vendor FFT (tests/fft_recipe.py): the HIP path is a dense
product, the reference an FFT-based one -- this is the comparison of the two algorithms on the device itself, on N(0,1)
inputs at sizes up to BASELINE config 3, in every direction.  Tolerance 1e-11 normwise (observed 1e-15 .. 1e-13; the FFT
route loses a few digits to the division by sin(theta) near the end points)."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from fft_recipe import cheb_fft

pytestmark = pytest.mark.gpu
sp = ge.load()


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 128, 128), (33, 20, 17), (200, 130), (7, 256, 12), (256, 256, 256), (100,)],
                         ids=lambda s: "x".join(map(str, s)))
def test_cheb_apply_equals_the_fft_recipe(shape):
    torch.manual_seed(20240229 + len(shape))
    x = torch.randn(shape, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for tr in range(len(shape)):
        if shape[tr] < 3:
            continue
        plan = sp.ChebPlan(shape, tr)
        plan.mult(x.reshape(-1), y.reshape(-1))
        ref = cheb_fft(x, tr)
        torch.cuda.synchronize()
        err = float((ref - y).norm() / ref.norm())
        plan.destroy()
        assert err < 1e-11, (shape, tr, err)


def _interior(t):
    return t[tuple(slice(1, -1) for _ in range(t.dim()))]


def _boundary_mask(dims, device):
    m = torch.zeros(dims, dtype=torch.bool, device=device)
    for k, n in enumerate(dims):
        idx = [slice(None)] * len(dims)
        idx[k] = 0; m[tuple(idx)] = True
        idx[k] = n - 1; m[tuple(idx)] = True
    return m


@pytest.mark.parametrize("dims", [(64, 64, 64), (128, 128, 128), (256, 256, 256), (136, 200), (68, 70, 72)], ids=lambda s: "x".join(map(str, s)))
def test_poisson_matvec_equals_the_fft_recipe(dims):
    """MatMult_Elliptic, linear state (elliptic.C:297-339 with eta = 1): V = -sum_k D_k D_k w0 on the interior, w0 = U with
    zero boundary values -- with both derivatives of every direction taken by the FFT recipe.  The HIP path applies the
    interior block of D D in one dense product per direction."""
    op = sp.EllipticOp(dims)
    torch.manual_seed(20240229)
    U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    V = torch.empty_like(U)
    op.mult(U, V)
    w0 = torch.zeros(dims, dtype=torch.float64, device="cuda")
    _interior(w0).copy_(U.view([n - 2 for n in dims]))
    acc = torch.zeros_like(w0)
    for k in range(len(dims)):
        acc -= cheb_fft(cheb_fft(w0, k), k)
    ref = _interior(acc).reshape(-1)
    torch.cuda.synchronize()
    assert float((ref - V).norm() / ref.norm()) < 1e-10
    op.destroy()


@pytest.mark.parametrize("dims,exponent", [((68, 70, 72), 2.0), ((128, 128, 128), 2.0), ((136, 200), 3.0), ((40, 33, 20), 2.5)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else "e%g" % v)
def test_function_and_jacobian_equal_the_fft_recipe(dims, exponent):
    """FormFunction (elliptic.C:481-533) and the Jacobian apply (:297-339) with variable coefficients, every derivative by the
    FFT recipe: F = -sum_k D_k(eta D_k w0) - b, eta = 1 + gamma w0^e, w0 = scatter(u) with the Dirichlet values on the boundary;
    J x = -sum_k D_k(eta D_k x0 + deta x0 D_k w0), x0 = x with zero boundary values."""
    gamma = 1.5
    op = sp.EllipticOp(dims)
    dev = "cuda"
    torch.manual_seed(7 + len(dims))
    full = torch.rand(dims, dtype=torch.float64, device=dev) + 0.5           # positive state incl. boundary values
    mask = _boundary_mask(dims, dev)
    op.set_dirichlet(full[mask].cpu().numpy())                               # boundary nodes in row-major (BlockIt) order
    u = _interior(full).reshape(-1).contiguous()
    b = torch.randn_like(u); r = torch.empty_like(u)
    op.function(u, b, r, gamma, exponent)
    eta = 1.0 + gamma * full ** exponent
    deta = exponent * gamma * full ** (exponent - 1.0)
    acc = torch.zeros_like(full); grads = []
    for k in range(len(dims)):
        g = cheb_fft(full, k); grads.append(g)
        acc -= cheb_fft(eta * g, k)
    ref = _interior(acc).reshape(-1) - b
    torch.cuda.synchronize()
    assert float((ref - r).norm() / ref.norm()) < 1e-10
    x = torch.randn_like(u); y = torch.empty_like(u)
    op.mult(x, y)
    x0 = torch.zeros_like(full); _interior(x0).copy_(x.view([n - 2 for n in dims]))
    acc = torch.zeros_like(full)
    for k in range(len(dims)):
        acc -= cheb_fft(eta * cheb_fft(x0, k) + deta * x0 * grads[k], k)
    ref = _interior(acc).reshape(-1)
    torch.cuda.synchronize()
    assert float((ref - y).norm() / ref.norm()) < 1e-10
    op.destroy()


def _end_weights(P):
    """Values at the two end points of the polynomial through the interior Gauss-Lobatto values of a line of P points
    (StokesPressureReduceOrder, stokes.C:1029-1080, as a linear functional): Lagrange weights in long double."""
    x = np.cos(np.pi * np.arange(P, dtype=np.longdouble) / (P - 1))
    xi = x[1:-1]
    w = np.empty((2, P - 2), dtype=np.longdouble)
    for e, xe in enumerate((x[0], x[-1])):
        for j in range(P - 2):
            others = np.delete(xi, j)
            w[e, j] = np.prod((xe - others) / (xi[j] - others))
    return torch.from_numpy(w.astype(np.float64)).cuda()


@pytest.mark.parametrize("dims", [(40, 33, 20), (64, 64, 64), (30, 26)], ids=lambda s: "x".join(map(str, s)))
def test_stokes_blocks_equal_the_fft_recipe(dims):
    """StokesMatMultVV / PV / VP and StokesMatMult in the linear state (stokes.C:499-676) with every ChebMult taken by the FFT
    recipe: VV = -sum_j D_j (sym grad v)_{j.}, PV = div v on the interior, VP = grad of the pressure whose face values are
    the end-point extrapolations of the interior values along the face normal."""
    d = len(dims)
    op = sp.StokesOp(dims)
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
    torch.manual_seed(99 + d)
    x = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    idims = [n - 2 for n in dims]
    X = x.view(*idims, d + 1)
    vL = []
    for k in range(d):
        t = torch.zeros(dims, dtype=torch.float64, device="cuda"); _interior(t).copy_(X[..., k]); vL.append(t)
    g = [[cheb_fft(vL[k], j) for k in range(d)] for j in range(d)]              # g[j][k] = D_j v_k
    yv = []
    for k in range(d):
        acc = torch.zeros(dims, dtype=torch.float64, device="cuda")
        for j in range(d):
            acc -= cheb_fft(0.5 * (g[j][k] + g[k][j]), j)                         # eta = 1
        yv.append(_interior(acc))
    div = _interior(sum(g[k][k] for k in range(d)))
    pL = torch.zeros(dims, dtype=torch.float64, device="cuda"); _interior(pL).copy_(X[..., d])
    for k in range(d):                                                           # face values along the normal
        w = _end_weights(dims[k])
        inner = pL.narrow(k, 1, dims[k] - 2)
        shape = [1] * d; shape[k] = -1
        pL.narrow(k, 0, 1).copy_((inner * w[0].view(shape)).sum(k, keepdim=True))
        pL.narrow(k, dims[k] - 1, 1).copy_((inner * w[1].view(shape)).sum(k, keepdim=True))
    gp = [_interior(cheb_fft(pL, k)) for k in range(d)]

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    v_in = X[..., :d].reshape(-1).contiguous(); p_in = X[..., d].reshape(-1).contiguous()
    out_v = torch.empty(op.velocity_size, dtype=torch.float64, device="cuda"); out_p = torch.empty(op.pressure_size, dtype=torch.float64, device="cuda")
    op.mult_vv(v_in, out_v); torch.cuda.synchronize()
    assert rel(out_v.view(*idims, d), torch.stack(yv, dim=-1)) < 1e-10
    op.mult_pv(v_in, out_p); torch.cuda.synchronize()
    assert rel(out_p.view(*idims), div) < 1e-10
    op.mult_vp(p_in, out_v); torch.cuda.synchronize()
    assert rel(out_v.view(*idims, d), torch.stack(gp, dim=-1)) < 1e-9
    y = torch.empty_like(x)
    op.mult(x, y); torch.cuda.synchronize()
    ref = torch.cat([torch.stack([yv[k] + gp[k] for k in range(d)], dim=-1), div.unsqueeze(-1)], dim=-1)
    assert rel(y.view(*idims, d + 1), ref) < 1e-9
    op.destroy()


@pytest.mark.parametrize("dims", [(24, 20, 18), (34, 30)], ids=lambda s: "x".join(map(str, s)))
def test_stokes_power_law_equals_the_fft_recipe(dims):
    """StokesFunction with the power law of README:52 (stokes.C:680-758, rheology :1920-1944) and the Newton-linearised
    StokesMatMult that follows it (:647-662), every ChebMult by the FFT recipe; zero Dirichlet values and force."""
    d = len(dims)
    hardness, expo, eps, gamma0 = 1.0, 3.0, 1e-4, 1.0
    op = sp.StokesOp(dims)
    op.set_rheology(1, hardness, expo, eps, gamma0)
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
    idims = [n - 2 for n in dims]
    w = [_end_weights(n) for n in dims]

    def fields(x):
        X = x.view(*idims, d + 1)
        vL = []
        for k in range(d):
            t = torch.zeros(dims, dtype=torch.float64, device="cuda"); _interior(t).copy_(X[..., k]); vL.append(t)
        g = [[cheb_fft(vL[k], j) for k in range(d)] for j in range(d)]
        s = [[0.5 * (g[j][k] + g[k][j]) for k in range(d)] for j in range(d)]
        pL = torch.zeros(dims, dtype=torch.float64, device="cuda"); _interior(pL).copy_(X[..., d])
        for k in range(d):
            inner = pL.narrow(k, 1, dims[k] - 2); shape = [1] * d; shape[k] = -1
            pL.narrow(k, 0, 1).copy_((inner * w[k][0].view(shape)).sum(k, keepdim=True))
            pL.narrow(k, dims[k] - 1, 1).copy_((inner * w[k][1].view(shape)).sum(k, keepdim=True))
        gp = [_interior(cheb_fft(pL, k)) for k in range(d)]
        div = _interior(sum(g[k][k] for k in range(d)))
        return s, gp, div

    def assemble(tau, gp, div):
        yv = []
        for k in range(d):
            acc = torch.zeros(dims, dtype=torch.float64, device="cuda")
            for j in range(d):
                acc -= cheb_fft(tau[j][k], j)
            yv.append(_interior(acc) + gp[k])
        return torch.cat([torch.stack(yv, dim=-1), div.unsqueeze(-1)], dim=-1)

    torch.manual_seed(5 + d)
    x = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    op.function(x, y); torch.cuda.synchronize()
    s0, gp, div = fields(x)
    gam = sum(0.5 * s0[j][k] * s0[j][k] for j in range(d) for k in range(d))
    p = (1.0 - expo) / (2.0 * expo)
    q = eps + gam / gamma0
    eta = hardness * q ** p
    deta = hardness * p / gamma0 * q ** (p - 1.0)
    ref = assemble([[eta * s0[j][k] for k in range(d)] for j in range(d)], gp, div)
    assert float((y.view(*idims, d + 1) - ref).norm() / ref.norm()) < 1e-9
    z = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    op.mult(z, y); torch.cuda.synchronize()
    s1, gp1, div1 = fields(z)
    zz = sum(s1[j][k] * s0[j][k] for j in range(d) for k in range(d))
    ref = assemble([[eta * s1[j][k] + deta * s0[j][k] * zz for k in range(d)] for j in range(d)], gp1, div1)
    assert float((y.view(*idims, d + 1) - ref).norm() / ref.norm()) < 1e-9
    op.destroy()
