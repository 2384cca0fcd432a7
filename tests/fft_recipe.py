"""The reference's ChebMult recipe -- DCT-I -> times k -> DST-I -> scale, chebyshev.c:157-193 -- evaluated with the vendor FFT
(torch.fft = rocFFT on the real even / odd extensions of length 2(P-1)) and elementwise torch ops.  A third restatement of
the same arithmetic, independent both of the HIP kernels (dense parity-split products) and of the CPU oracle (its own
transforms): tests/test_gpu_fft_route.py checks cheb_apply against it, tools/fft_route_bench.py times it."""
import math

import numpy as np
import torch


def cheb_fft(x, dim):
    P = x.shape[dim]; n = P - 1
    shape = [1] * x.dim(); shape[dim] = -1
    k = torch.arange(0, n + 1, dtype=torch.float64, device=x.device).view(shape)
    xe = torch.cat([x, x.flip(dim).narrow(dim, 1, n - 1)], dim)                 # even extension, length 2n
    Y = torch.fft.rfft(xe, dim=dim).real                                        # REDFT00: Y_0..Y_n            (:157)
    W = Y * k                                                                   # k * Y_k                      (:171)
    Wi = W.narrow(dim, 1, n - 1)
    sgn = torch.where(torch.arange(0, n + 1, device=x.device) % 2 == 0, 1.0, -1.0).to(torch.float64).view(shape)
    y0 = (W * k).narrow(dim, 1, n - 1).sum(dim, keepdim=True) / n + 0.5 * n * Y.narrow(dim, n, 1)                       # (:172,176)
    yn = ((W * k) * (-sgn)).narrow(dim, 1, n - 1).sum(dim, keepdim=True) / n + 0.5 * (-1.0) ** (n + 1) * n * Y.narrow(dim, n, 1)   # (:173,177)
    z = torch.zeros_like(x.narrow(dim, 0, 1))
    oe = torch.cat([z, Wi, z, -Wi.flip(dim)], dim)                              # odd extension, length 2n
    Z = -torch.fft.rfft(oe, dim=dim).imag.narrow(dim, 1, n - 1)                 # RODFT00: 2 sum W_k sin(pi j k / n)  (:181)
    j = torch.arange(1, n, dtype=torch.float64, device=x.device).view(shape)
    yi = Z / (2.0 * n * torch.sin(math.pi * j / n))                             # (:190)
    return torch.cat([y0, yi, yn], dim)


def interior(t):
    return t[tuple(slice(1, -1) for _ in range(t.dim()))]


def boundary_mask(dims, device):
    m = torch.zeros(dims, dtype=torch.bool, device=device)
    for k, n in enumerate(dims):
        idx = [slice(None)] * len(dims)
        idx[k] = 0; m[tuple(idx)] = True
        idx[k] = n - 1; m[tuple(idx)] = True
    return m


def end_weights(P, device):
    """Values at the two end points of the polynomial through the interior Gauss-Lobatto values of a line of P points
    (StokesPressureReduceOrder, stokes.C:1029-1080, as a linear functional): Lagrange weights in long double."""
    x = np.cos(np.pi * np.arange(P, dtype=np.longdouble) / (P - 1))
    xi = x[1:-1]
    w = np.empty((2, P - 2), dtype=np.longdouble)
    for e, xe in enumerate((x[0], x[-1])):
        for j in range(P - 2):
            others = np.delete(xi, j)
            w[e, j] = np.prod((xe - others) / (xi[j] - others))
    return torch.from_numpy(w.astype(np.float64)).to(device)


def poisson_ref(dims, U):
    """MatMult_Elliptic, linear state (elliptic.C:297-339 with eta = 1): -sum_k D_k D_k w0 on the interior, w0 = U with zero
    boundary values."""
    w0 = torch.zeros(dims, dtype=torch.float64, device=U.device)
    interior(w0).copy_(U.view([n - 2 for n in dims]))
    acc = torch.zeros_like(w0)
    for k in range(len(dims)):
        acc -= cheb_fft(cheb_fft(w0, k), k)
    return interior(acc).reshape(-1)


def elliptic_function_ref(dims, full, b, gamma, exponent):
    """FormFunction (elliptic.C:481-533): F = -sum_k D_k(eta D_k w0) - b with eta = 1 + gamma w0^e; `full` is w0 on the whole
    local grid (interior values and Dirichlet values).  Returns F, eta, deta, [D_k w0]."""
    eta = 1.0 + gamma * full ** exponent
    deta = exponent * gamma * full ** (exponent - 1.0)
    acc = torch.zeros_like(full); grads = []
    for k in range(len(dims)):
        g = cheb_fft(full, k); grads.append(g)
        acc -= cheb_fft(eta * g, k)
    F = interior(acc).reshape(-1)
    return (F - b if b is not None else F), eta, deta, grads


def elliptic_jacobian_ref(dims, x, eta, deta, grads):
    """MatMult_Elliptic (elliptic.C:297-339): J x = -sum_k D_k(eta D_k x0 + deta x0 D_k w0), x0 = x with zero boundary values."""
    x0 = torch.zeros(dims, dtype=torch.float64, device=x.device)
    interior(x0).copy_(x.view([n - 2 for n in dims]))
    acc = torch.zeros_like(x0)
    for k in range(len(dims)):
        acc -= cheb_fft(eta * cheb_fft(x0, k) + deta * x0 * grads[k], k)
    return interior(acc).reshape(-1)


def stokes_fields(dims, x, weights=None):
    """Symmetrised velocity gradient s[j][k] on the local grid, pressure gradient and divergence on the interior, of a global
    Stokes vector x (interior nodes x (d + 1), zero Dirichlet values): stokes.C:583-591, 605-614, 634-646."""
    d = len(dims); idims = [n - 2 for n in dims]
    X = x.view(*idims, d + 1)
    vL = []
    for k in range(d):
        t = torch.zeros(dims, dtype=torch.float64, device=x.device); interior(t).copy_(X[..., k]); vL.append(t)
    g = [[cheb_fft(vL[k], j) for k in range(d)] for j in range(d)]              # g[j][k] = D_j v_k
    s = [[0.5 * (g[j][k] + g[k][j]) for k in range(d)] for j in range(d)]
    pL = torch.zeros(dims, dtype=torch.float64, device=x.device); interior(pL).copy_(X[..., d])
    for k in range(d):                                                           # face values along the normal
        w = weights[k] if weights is not None else end_weights(dims[k], x.device)
        inner = pL.narrow(k, 1, dims[k] - 2); shape = [1] * d; shape[k] = -1
        pL.narrow(k, 0, 1).copy_((inner * w[0].view(shape)).sum(k, keepdim=True))
        pL.narrow(k, dims[k] - 1, 1).copy_((inner * w[1].view(shape)).sum(k, keepdim=True))
    gp = [interior(cheb_fft(pL, k)) for k in range(d)]
    div = interior(sum(g[k][k] for k in range(d)))
    return s, gp, div


def stokes_assemble(dims, tau, gp, div):
    """Velocity rows -sum_j D_j tau[j][k] (+ grad p when gp is given), pressure rows div: the global vector as (nodes, d + 1)."""
    d = len(dims)
    yv = []
    for k in range(d):
        acc = torch.zeros(dims, dtype=torch.float64, device=div.device)
        for j in range(d):
            acc -= cheb_fft(tau[j][k], j)
        yv.append(interior(acc) + (gp[k] if gp is not None else 0.0))
    return torch.cat([torch.stack(yv, dim=-1), div.unsqueeze(-1)], dim=-1)


def power_law(s, hardness, expo, eps, gamma0):
    """eta and its derivative with respect to the second invariant (stokes.C:1920-1944) of a strain-rate field s[j][k]."""
    d = len(s)
    gam = sum(0.5 * s[j][k] * s[j][k] for j in range(d) for k in range(d))
    p = (1.0 - expo) / (2.0 * expo)
    q = eps + gam / gamma0
    return hardness * q ** p, hardness * p / gamma0 * q ** (p - 1.0)
