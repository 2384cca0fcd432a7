"""The finite-difference preconditioner on the device: FormJacobian's matrix P (elliptic.C:537-590),
MatVVPC of StokesPCSetUp0 (stokes.C:1160-1241) and the approximate solves with them, against the oracle's
assembled matrices (oracle_lib.fd_matrix restates the reference loops) and scipy's sparse direct solver."""
import numpy as np
import pytest
import torch
import scipy.sparse as sps
import scipy.sparse.linalg as spl

import __graft_entry__ as ge
import oracle_lib as orc
from conftest import relerr

pytestmark = pytest.mark.gpu
sp = ge.load()
SEED = 20240229


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def out(n):
    return torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")


@pytest.mark.parametrize("dims", [(9,), (12, 10), (8, 7, 6), (33, 20), (66, 12, 5), (5, 4, 4, 3)], ids=lambda d: "x".join(map(str, d)))
def test_fd_matrix_and_exact_solve_linear_state(dims):
    """eta == 1: P is the separable operator; fd_mult = oracle matrix, apply = its exact inverse."""
    op = sp.EllipticOp(dims)
    pc = sp.FdPc(op, sweeps=0)
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal(op.global_size)
    P = orc.fd_matrix(dims)
    y = pc.mult(dev(x), out(op.global_size)).cpu().numpy()
    assert relerr(y, P @ x) < 1e-13
    z = pc.apply(dev(x), out(op.global_size)).cpu().numpy()
    assert relerr(z, spl.spsolve(P.tocsc(), x)) < 1e-9
    pc.destroy(); op.destroy()


@pytest.mark.parametrize("dims", [(130, 70), (131, 70), (258, 20), (70, 68, 40)], ids=lambda d: "x".join(map(str, d)))
def test_exact_solve_long_lines(dims):
    """The same on lines of more than 64 interior points: the one-launch line transforms of the KS = 16 / 32 straight-line
    kernel (halves stored / read unsplit), even and odd line lengths, both tilings.  P P^-1 x = x through the stencil apply
    keeps the check free of a large sparse factorisation."""
    op = sp.EllipticOp(dims)
    pc = sp.FdPc(op, sweeps=0)
    rng = np.random.default_rng(SEED + 5)
    x = rng.standard_normal(op.global_size)
    z = pc.apply(dev(x), out(op.global_size))
    back = pc.mult(z, out(op.global_size)).cpu().numpy()
    assert relerr(back, x) < 1e-9
    if len(dims) == 2:
        P = orc.fd_matrix(dims)
        assert relerr(z.cpu().numpy(), spl.spsolve(P.tocsc(), x)) < 1e-9
    pc.destroy(); op.destroy()


def test_apply_on_vectors_that_are_only_8_byte_aligned():
    """Sub-vectors at an odd offset: the 16-byte line-transform kernels do not apply, the two-launch general route must give
    the same preconditioner."""
    dims = (130, 70)
    op = sp.EllipticOp(dims)
    pc = sp.FdPc(op, sweeps=0)
    n = op.global_size
    x = np.random.default_rng(SEED + 6).standard_normal(n)
    z_al = pc.apply(dev(x), out(n)).cpu().numpy()
    bx = torch.empty(n + 1, dtype=torch.float64, device="cuda"); bz = torch.full((n + 1,), float("nan"), dtype=torch.float64, device="cuda")
    xo, zo = bx[1:], bz[1:]
    assert xo.data_ptr() % 16 == 8 and zo.data_ptr() % 16 == 8
    xo.copy_(torch.from_numpy(x))
    pc.apply(xo, zo); torch.cuda.synchronize()
    assert relerr(zo.cpu().numpy(), z_al) < 1e-12
    pc.destroy(); op.destroy()


@pytest.mark.parametrize("dims", [(14, 12), (10, 9, 8)], ids=lambda d: "x".join(map(str, d)))
def test_fd_matrix_nonlinear_state(dims):
    """After FormFunction with gamma != 0 the stencil carries eta, deta and grad u (elliptic.C:571-575)."""
    op = sp.EllipticOp(dims)
    rng = np.random.default_rng(SEED)
    u, u2, dv = orc.elliptic_exact(dims, 0, gamma=2.0, exponent=2.0, cos_scale=2.0)
    op.set_dirichlet(dv)
    U = (u + 1.5) * (1.0 + 0.05 * rng.standard_normal(op.global_size))
    op.function_host(U, None, 2.0, 2.0)
    pc = sp.FdPc(op, sweeps=3)
    _, eta, deta, gradu = orc.elliptic_function(dims, U, None, dv, 2.0, 2.0, mode=orc.DIRECT)
    P = orc.fd_matrix(dims, eta, deta, gradu)
    x = rng.standard_normal(op.global_size)
    assert relerr(pc.mult(dev(x), out(op.global_size)).cpu().numpy(), P @ x) < 1e-12
    # the inner GMRES on P is monotone in the residual and converges to P^-1 r
    res = []
    for s in (0, 3, 12, 40):
        sp.lib().chebhip_fdpc_set_sweeps(pc._h, s)
        z = pc.apply(dev(x), out(op.global_size)).cpu().numpy()
        res.append(np.linalg.norm(x - P @ z) / np.linalg.norm(x))
    assert res[1] < res[0] and res[2] < res[1] and res[3] < 1e-6, res
    assert relerr(z, spl.spsolve(P.tocsc(), x)) < 1e-5
    pc.destroy(); op.destroy()


@pytest.mark.parametrize("dims", [(48, 48), (24, 22, 20), (64, 64, 64)], ids=lambda d: "x".join(map(str, d)))
def test_preconditioned_fgmres_converges_fast(dims):
    """KSPFGMRES + the finite-difference preconditioner (elliptic.C:181-185): a handful of iterations where the
    unpreconditioned solve needs hundreds -- the condition number of P^-1 A is bounded independently of the order."""
    op = sp.EllipticOp(dims)
    pc = sp.FdPc(op, sweeps=0)
    rng = np.random.default_rng(SEED)
    xs = rng.standard_normal(op.global_size)
    b = op.mult(dev(xs), out(op.global_size))
    x = torch.zeros_like(b)
    ks = sp.Fgmres(op.global_size, restart=30, rtol=1e-10, max_it=200)
    ks.solve(op, b, x, M=pc)
    torch.cuda.synchronize()
    assert ks.reason > 0 and ks.iterations <= 40, (ks.reason, ks.iterations)
    assert relerr(x.cpu().numpy(), xs) < 1e-7
    ks.destroy(); pc.destroy(); op.destroy()


@pytest.mark.parametrize("dims", [(10, 9), (7, 6, 5), (20, 18, 16)], ids=lambda d: "x".join(map(str, d)))
def test_stokes_velocity_pc(dims):
    """MatVVPC (stokes.C:1181-1226): per component the eta-only stencil, on node-major velocity vectors."""
    d = len(dims)
    st = sp.StokesOp(dims)
    pc = sp.FdPc(st, sweeps=0)
    rng = np.random.default_rng(SEED)
    v = rng.standard_normal(st.velocity_size)
    P1 = orc.fd_matrix(dims)                                  # linear rheology: eta == 1
    P = sps.kron(P1, sps.identity(d)).tocsr()                 # row = node * d + component (stokes.C:1211)
    assert relerr(pc.mult(dev(v), out(st.velocity_size)).cpu().numpy(), P @ v) < 1e-13
    assert relerr(pc.apply(dev(v), out(st.velocity_size)).cpu().numpy(), spl.spsolve(P.tocsc(), v)) < 1e-9
    # as the preconditioner of KSPVelocity (operators MatVV, MatVVPC; stokes.C:328-333)
    b = st.mult_vv(dev(v), out(st.velocity_size))
    x = torch.zeros_like(b)
    ks = sp.Fgmres(st.velocity_size, restart=30, rtol=1e-10, max_it=300)
    ks.solve(st, b, x, M=pc, a_entry="mult_vv")
    torch.cuda.synchronize()
    assert ks.reason > 0 and ks.iterations <= 80, (ks.reason, ks.iterations)
    assert relerr(x.cpu().numpy(), v) < 1e-6
    ks.destroy(); pc.destroy(); st.destroy()


@pytest.mark.parametrize("visc", ["unit", "variable"])
@pytest.mark.parametrize("dims,stokes", [((20, 18, 16), False), ((34, 34, 34), False), ((66, 66, 66), False), ((70, 68, 40), False), ((40, 68, 130), False),
                                         ((130, 70), False), ((40, 131), False), ((131, 40), False), ((34, 34, 34), True), ((20, 18, 130), True), ((70, 20, 18), True),
                                         ((20, 18, 128), True), ((34, 20, 100), False), ((30, 128), False), ((36, 70, 68), True)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else ("stokes" if v else "scalar"))
def test_pointwise_steps_inside_the_line_transforms(dims, stokes, visc):
    """z = P_1^-1 (r / eta): the division by eta rides on the LOAD of the first forward line transform (IN_MUL, lines of more than
    64 points) and the modal scaling on the STORE of the last one (OUT_MUL) -- multiplications by reciprocals, sweep.h -- where
    those transforms are one launch of the 16-byte kernels: short and long lines, both tilings, one and d stacked fields; odd
    interior extents keep the passes.  Where the last dimension has 66 .. 128 interior points (an even number) the last forward
    transform, the scaling and the first backward transform are ONE launch (k_fdm_zsolve16; H even and odd, partial tiles).  Option `fdm_passes` restores the two passes that divide: the routes agree to rounding,
    and with eta == 1 the solve is still the inverse of the stencil."""
    op = sp.StokesOp(dims) if stokes else sp.EllipticOp(dims)
    n = op.velocity_size if stokes else op.global_size
    rng = np.random.default_rng(SEED + 11)
    if visc == "variable":
        if stokes:
            op.set_state(0, np.exp(rng.uniform(np.log(0.5), np.log(10.0), int(np.prod(dims)))))
        else:                                             # eta = 1 + u^2 of a FormFunction (elliptic.C:507-513, gamma = 1)
            u = dev(rng.uniform(0.0, 2.0, n))
            op.function(u, dev(np.zeros(n)), out(n), 1.0, 2.0)
    x = dev(rng.standard_normal(n))
    pc = sp.FdPc(op, sweeps=0)
    z = pc.apply(x, out(n)).cpu().numpy()
    if visc == "unit":
        back = pc.mult(dev(z), out(n)).cpu().numpy()
        assert relerr(back, x.cpu().numpy()) < 1e-9
    zc = None
    if stokes:                                            # ... and on component-major vectors (the block preconditioners' layout)
        d = len(dims)
        xc = dev(np.ascontiguousarray(x.cpu().numpy().reshape(-1, d).T).ravel())
        zc = np.ascontiguousarray(pc.apply_cm(xc, out(n)).cpu().numpy().reshape(d, -1).T).ravel()
    pc.destroy()
    sp.set_option("fdm_passes", 1)
    try:
        pc = sp.FdPc(op, sweeps=0)
        z1 = pc.apply(x, out(n)).cpu().numpy()
        pc.destroy()
    finally:
        sp.set_option("fdm_passes", 0)
    assert relerr(z, z1) < 1e-12
    if zc is not None:
        assert relerr(zc, z1) < 1e-12
    op.destroy()
