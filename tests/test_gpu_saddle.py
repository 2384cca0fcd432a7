"""The Stokes block preconditioners StokesPCApply0..3 (stokes.C:1714-1817) and the solve phase of stokes.C:213-235
(Newton, continuation, FGMRES) on the device, against dense algebra on the oracle's operators."""
import numpy as np
import pytest
import torch
from importlib import import_module

import __graft_entry__ as ge
import oracle_lib as orc
from conftest import relerr

pytestmark = pytest.mark.gpu
sp = ge.load()
solve = import_module(sp.__name__ + ".solve")
SEED = 20240229


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def dense(fn, nin, nout):
    A = np.empty((nout, nin)); e = np.zeros(nin)
    for j in range(nin):
        e[j] = 1.0; A[:, j] = fn(e); e[j] = 0.0
    return A


def blocks(dims):
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    VV = dense(lambda e: orc.stokes_mult_vv(dims, e, mode=orc.DIRECT), gv, gv)
    VP = dense(lambda e: orc.stokes_mult_vp(dims, e, mode=orc.DIRECT), gp, gv)
    PV = dense(lambda e: orc.stokes_divergence(dims, e, mode=orc.DIRECT), gv, gp)
    return VV, VP, PV


def split(x, d):
    X = x.reshape(-1, d + 1)
    return np.ascontiguousarray(X[:, :d]).ravel(), np.ascontiguousarray(X[:, d])


def merge(v, p, d):
    return np.concatenate([v.reshape(-1, d), p[:, None]], axis=1).ravel()


def blocks_eta(dims, eta):
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    VV = dense(lambda e: orc.stokes_mult_vv(dims, e, eta=eta, mode=orc.DIRECT), gv, gv)
    VP = dense(lambda e: orc.stokes_mult_vp(dims, e, mode=orc.DIRECT), gp, gv)
    PV = dense(lambda e: orc.stokes_divergence(dims, e, mode=orc.DIRECT), gv, gp)
    return VV, VP, PV


@pytest.mark.parametrize("dims", [(7, 6), (6, 5, 5)], ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("visc", ["unit", "variable"])
def test_saddle_types_vs_dense(dims, kind, visc):
    """With the inner solves run to convergence each PCApply is the block formula of its comment (stokes.C:1712,1745,
    1770,1795), checked against dense algebra on the oracle's blocks (pressure up to its constant).  `variable`: a
    viscosity field varying by 20x, so that KSPSchur's Jacobi scaling by eta (stokes.C:330-331, 538-553) takes part:
    the converged Schur solve satisfies P diag(eta) S x = P diag(eta) b on zero-mean vectors."""
    d = len(dims)
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    rng = np.random.default_rng(SEED)
    eta = np.ones(N) if visc == "unit" else np.exp(rng.uniform(np.log(0.5), np.log(10.0), N))
    st = sp.StokesOp(dims)
    st.set_state(0, eta)
    pc = sp.StokesSaddlePc(st, kind, vel=(400, 1e-14), schur=(400, 1e-13), svel=(400, 1e-14))
    pc.setup()
    x = rng.standard_normal(st.global_size)
    y = pc.apply(dev(x), torch.empty(st.global_size, dtype=torch.float64, device="cuda")).cpu().numpy()
    VV, VP, PV = blocks_eta(dims, eta)
    S = -PV @ np.linalg.solve(VV, VP)                                   # MatSchur (stokes.C:523-535)
    # KSPSchur: left-preconditioned GMRES, PCJACOBI with diagonal 1/eta (interior nodes), constant null space attached
    # (stokes.C:1020-1021): P D S x = P D b on zero-mean vectors, D = diag(eta_interior)
    n_p = S.shape[0]
    idx = np.arange(N).reshape(dims)
    interior = idx[tuple(slice(1, -1) for _ in dims)].ravel()
    Dm = np.diag(eta[interior])
    Q = np.linalg.qr(np.eye(n_p) - 1.0 / n_p)[0][:, :n_p - 1]           # orthonormal basis of the zero-mean subspace
    Sinv = Q @ np.linalg.solve(Q.T @ Dm @ S @ Q, Q.T @ Dm)
    xv, xp = split(x, d)
    Ai = lambda b: np.linalg.solve(VV, b)
    if kind == 0:
        v1 = Ai(xv); p1 = Sinv @ (xp - PV @ v1); yv = v1 + Ai(-VP @ p1)
    elif kind == 1:
        p1 = Sinv @ xp; yv = Ai(xv - VP @ p1)
    elif kind == 2:
        yv = Ai(xv); p1 = Sinv @ xp
    else:
        yv = Ai(xv); p1 = Sinv @ (xp - PV @ yv)
    gv_, gp_ = split(y, d)
    assert abs(gp_.mean()) < 1e-10 * (1 + np.abs(gp_).max())           # KSPSetNullSpace: zero-mean pressure
    assert relerr(gv_, yv) < 1e-8
    assert relerr(gp_ - gp_.mean(), p1 - p1.mean()) < 1e-8
    pc.destroy(); st.destroy()


@pytest.mark.parametrize("dims,kind", [((12, 11), 0), ((12, 11), 1), ((12, 12, 12), 0), ((12, 11, 13), 3), ((12, 12, 12), 2), ((20, 20, 20), 0)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_linear_stokes_solve_exact2(dims, kind):
    """./stokes -exact 2 -dim ... -ksp_type fgmres -ksp_rtol 1e-10 with the README's inner settings (README:43):
    the discrete solution approaches the analytic one (stokes.C:222-234, error with the pressure constant removed)."""
    d = len(dims)
    st = sp.StokesOp(dims)
    U, U2, dv = orc.stokes_exact(dims, 2)
    st.set_dirichlet(dv); st.set_force(U2)
    x = torch.zeros(st.global_size, dtype=torch.float64, device="cuda")
    log = solve.stokes_solve(sp, st, x, rheology=(0, 1.0, 1.0, 1.0, 1.0), saddle_type=kind, snes_rtol=1e-9, ksp_rtol=1e-10,
                             ksp_restart=60, ksp_max_it=400)
    torch.cuda.synchronize()
    e, r, its, kits, fn = log[-1]
    assert its <= 3 and kits <= 200, log
    xs = x.cpu().numpy()
    xv, xp = split(xs, d); uv, up = split(U, d)
    bound = 5e-6                                                        # spectral accuracy of sin/cos(pi x / 2) on >= 11 points
    assert np.abs(xv - uv).max() < bound and np.abs((xp - xp.mean()) - (up - up.mean())).max() < 50 * bound
    st.destroy()


def test_power_law_continuation_root_of_oracle_residual():
    """./stokes -exact 2 -cont 2 -rheology 1 -eps 1e-2 -exponent 3 with the README's inner settings (README:52): every
    continuation stage of stokes.C:217-221 converges in a few Newton steps (quadratically: the Jacobian apply is the
    exact linearisation), and the final state is a root of the ORACLE's StokesFunction with the final rheology."""
    dims = (16, 16, 16)
    rheo = (1, 1.0, 3.0, 1e-2, 1.0)
    st = sp.StokesOp(dims)
    U, U2, dv = orc.stokes_exact(dims, 2)
    st.set_dirichlet(dv); st.set_force(U2)
    x = torch.zeros(st.global_size, dtype=torch.float64, device="cuda")
    hist = []
    log = solve.stokes_solve(sp, st, x, rheology=rheo, cont0=0, cont=2, snes_rtol=1e-8, ksp_rtol=1e-5, ksp_restart=60, ksp_max_it=200,
                             monitor=lambda e, r, it, fn, k, lam: hist.append((e, it, fn, k)))
    torch.cuda.synchronize()
    assert [round(s[0], 4) for s in log] == [1.0, round(1.0 + 0.5 ** 0.8 * 2.0, 4), 3.0]          # exponents, stokes.C:219
    assert np.allclose([s[1] for s in log], [1.0, 0.1, 1e-2])                                       # regularisation, :220
    assert all(s[2] <= 6 for s in log) and all(h[3] <= 40 for h in hist), (log, hist)
    last = [h[2] for h in hist if abs(h[0] - 3.0) < 1e-12]
    assert len(last) >= 2 and last[-1] < 1e-3 * last[-2]                                            # Newton's last step contracts fast
    F = orc.stokes_function(dims, x.cpu().numpy(), dv, U2, rheology=rheo, mode=orc.FAST, nthreads=16)[0]
    assert np.linalg.norm(F) <= 1e-7 * np.linalg.norm(U2)
    st.destroy()


def _cm(v, d):
    """node-major (n d + c) -> component-major (c I + n)"""
    return np.ascontiguousarray(v.reshape(-1, d).T).ravel()


@pytest.mark.parametrize("dims", [(9, 8), (10, 9, 8), (20, 18, 16)], ids=lambda d: "x".join(map(str, d)))
def test_component_major_blocks_equal_the_node_major_ones(dims):
    """stokes_op_mult_vv_cm / _pv_cm / _vp_cm and chebhip_fdpc_apply_cm are the node-major entry points on permuted vectors:
    the same kernels between a gather and a scatter that index differently -- the same bits.  Variable viscosity, eta' = 0."""
    d = len(dims)
    st = sp.StokesOp(dims)
    N, I, gv, gp, g, ndv = orc.stokes_sizes(dims)
    rng = np.random.default_rng(SEED + 21)
    st.set_state(0, np.exp(rng.uniform(np.log(0.5), np.log(10.0), N)))
    v = rng.standard_normal(gv); p = rng.standard_normal(gp)
    o = lambda n: torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    assert np.array_equal(st.mult_vv_cm(dev(_cm(v, d)), o(gv)).cpu().numpy(), _cm(st.mult_vv(dev(v), o(gv)).cpu().numpy(), d))
    assert np.array_equal(st.mult_pv_cm(dev(_cm(v, d)), o(gp)).cpu().numpy(), st.mult_pv(dev(v), o(gp)).cpu().numpy())
    assert np.array_equal(st.mult_vp_cm(dev(p), o(gv)).cpu().numpy(), _cm(st.mult_vp(dev(p), o(gv)).cpu().numpy(), d))
    pc = sp.FdPc(st, sweeps=0)
    assert np.array_equal(pc.apply_cm(dev(_cm(v, d)), o(gv)).cpu().numpy(), _cm(pc.apply(dev(v), o(gv)).cpu().numpy(), d))
    pc.destroy(); st.destroy()


@pytest.mark.parametrize("dims,kind", [((12, 11), 0), ((12, 11, 13), 3), ((20, 20, 20), 0), ((20, 20, 20), 1)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_saddle_inner_layouts_agree(dims, kind):
    """The block preconditioners keep the vectors of their inner velocity solves component-major (no (de)interleaving around
    MatVVPC); option `saddle_node_major` restores the reference's layout.  With the README's truncated inner solves (4 / 3
    iterations, README:43) the two applies are the same preconditioner up to the rounding of their inner products."""
    rng = np.random.default_rng(SEED + 22)
    N = int(np.prod(dims))
    st = sp.StokesOp(dims)
    st.set_state(0, np.exp(rng.uniform(np.log(0.5), np.log(4.0), N)))
    x = dev(rng.standard_normal(st.global_size))
    ys = []
    for nm in (0, 1):
        sp.set_option("saddle_node_major", nm)
        try:
            pc = sp.StokesSaddlePc(st, kind)
        finally:
            sp.set_option("saddle_node_major", 0)
        pc.setup()
        ys.append(pc.apply(x, torch.empty(st.global_size, dtype=torch.float64, device="cuda")).cpu().numpy())
        its = pc.inner_iterations
        pc.destroy()
    assert relerr(ys[0], ys[1]) < 1e-9, relerr(ys[0], ys[1])
    st.destroy()
