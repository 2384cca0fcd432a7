"""Newton-Krylov driver on device vectors: the roles of SNESSolve and KSPSolve(KSPFGMRES) around the elliptic
callbacks (elliptic.C:177-185, 213), for end-to-end solves where no PETSc exists.

Each Newton step evaluates FormFunction (which leaves eta, eta', grad u behind, elliptic.C:498-509), then
solves J dx = -F with the matrix-free Jacobian MatMult_Elliptic (elliptic.C:297-339) by restarted FGMRES
(chebhip_fgmres_*), and updates x along dx with a backtracking line search; `M` is the slot for a right preconditioner
(the reference uses ILU(2) of a finite-difference matrix, elliptic.C:184-185, which stays PETSc's).
"""
import torch


def newton_krylov(sp, op, b, x, gamma=0.0, exponent=2.0, snes_rtol=1e-8, snes_atol=1e-50, snes_max_it=50,
                  ksp_rtol=1e-5, ksp_restart=30, ksp_max_it=10000, M=None, monitor=None, norm=None, line_search=True, ks=None):
    """Solve FormFunction(x) = A(x) x - b = 0 in place in x (device tensor).  Returns (newton_its, total_ksp_its, |F|).
    On several ranks (vectors = this rank's pieces) `op` is a callable driver of dist.py, `sp.Fgmres` must return a
    solver with its reduction set, and `norm` the global 2-norm.
    ks: a caller's Fgmres to use and keep (as a KSP object outlives its solves, elliptic.C:181-185); None: one is made and destroyed here."""
    n = op.global_size
    norm = norm or (lambda t: float(t.norm()))
    F = torch.empty_like(x)
    dx = torch.empty_like(x)
    own_ks = ks is None
    if own_ks:
        ks = sp.Fgmres(n, restart=ksp_restart, rtol=ksp_rtol, max_it=ksp_max_it)
    total = 0
    op.function(x, b, F, gamma, exponent)
    f0 = fn = norm(F)
    it = 0
    try:
        while it < snes_max_it and fn > max(snes_rtol * f0, snes_atol):
            F.neg_()
            ks.solve(op, F, dx, M=M)                    # J dx = -F, state of the last FormFunction
            total += ks.iterations
            if ks.reason < 0:                           # KSP_DIVERGED_*: SNES stops with SNES_DIVERGED_LINEAR_SOLVE
                raise RuntimeError("Newton step %d: linear solve diverged (reason %d after %d iterations, residual %.3e)"
                                   % (it + 1, ks.reason, ks.iterations, ks.residual))
            # backtracking line search on |F| (the role of SNES's default line search, elliptic.C:177-179:
            # SNESCreate leaves SNESLS in place): full step first, halved until sufficient decrease
            lam, fold = 1.0, fn
            x.add_(dx)
            op.function(x, b, F, gamma, exponent)
            fn = norm(F)
            while line_search and not (fn <= (1.0 - 1e-4 * lam) * fold) and lam > 1e-6:
                x.add_(dx, alpha=-0.5 * lam)
                lam *= 0.5
                op.function(x, b, F, gamma, exponent)
                fn = norm(F)
            it += 1
            if monitor:
                monitor(it, fn, ks.iterations)
    finally:
        if own_ks:
            ks.destroy()
    return it, total, fn


def continuation_schedule(exponent, regularization, cont0=0, cont=1):
    """The (exponent, regularization) pairs of the Newton continuation loop, stokes.C:217-221:
    exponent_i = 1 + (i/cont)^0.8 (exponent - 1), regularization_i = exp(log(regularization) i/cont)."""
    import math
    out = []
    for i in range(cont0, cont + 1):
        out.append((1.0 + math.pow(1.0 * i / cont, 0.8) * (exponent - 1.0), math.exp(math.log(regularization) * i / cont)))
    return out


def stokes_solve(sp, op, x, rheology=(0, 1.0, 1.0, 1.0, 1.0), cont0=0, cont=1, saddle_type=0,
                 snes_rtol=1e-8, snes_atol=1e-50, snes_max_it=50, ksp_rtol=1e-5, ksp_restart=30, ksp_max_it=10000,
                 vel=(4, 1e-5), schur=(3, 1e-5), svel=(0, 1e-5), pc_sweeps=0, line_search=True, monitor=None, max_linear_fail=1,
                 schur_jacobi=True, stats=None, dist=None, ks=None, pc=None):
    """The solve phase of stokes.C:213-235 on device vectors: for every continuation stage, SNESSolve = Newton with a
    backtracking line search around StokesFunction (stokes.C:680-758), each step KSPSolve(KSPFGMRES) on the
    Newton-linearised StokesMatMult (stokes.C:499-519) right-preconditioned by StokesPCApply<saddle_type>
    (stokes.C:1714-1817; MatVVPC re-assembled per step as StokesPCSetUp0 does).  Dirichlet values and force must be
    set on `op`; x (device tensor, global_size) holds the initial guess and the result.
    `max_linear_fail`: linear solves that may end on their iteration limit before the Newton iteration gives up
    (-snes_max_linear_solve_fail, PETSc's default 1); the step of such a solve is still tried by the line search.
    `stats` (a dict) receives "linear_fails": the number of linear solves that ended on their iteration limit.
    `dist`: a slab driver of dist.py (DistStokesC) whose slab-mode operator `op` is: the vectors are this rank's pieces, every
    rank calls collectively; norms, the Krylov inner products and the block preconditioner's sums go through its communicator.
    ks, pc: the caller's outer Fgmres and block preconditioner to use and keep (the KSP / PC objects of stokes.C:155-176 outlive their
    solves); None: made and destroyed here.
    Returns a list of (exponent, regularization, newton_its, ksp_its, |F|) per stage."""
    kind, hardness, exponent, regularization, gamma0 = rheology
    n = op.global_size
    F = torch.empty_like(x); dx = torch.empty_like(x)
    own_ks, own_pc = ks is None, pc is None
    if own_ks:
        ks = sp.Fgmres(n, restart=ksp_restart, rtol=ksp_rtol, max_it=ksp_max_it)
    if dist is not None:
        if own_ks:
            ks.set_reduce_raw(*dist.comm.reduce_fn())
        if own_pc:
            pc = dist.saddle(saddle_type, vel, schur, svel, schur_jacobi)
        gnorm = dist.comm.norm
    else:
        if own_pc:
            pc = sp.StokesSaddlePc(op, saddle_type, vel, schur, svel, pc_sweeps, schur_jacobi)
        gnorm = lambda t: float(t.norm())
    stages = continuation_schedule(exponent, regularization, cont0, cont) if kind == 1 else [(exponent, regularization)]
    out = []
    fails = 0
    try:
        for (e_i, r_i) in stages:
            op.set_rheology(kind, hardness, e_i, r_i, gamma0)                  # stokes.C:219-220
            op.function(x, F)
            f0 = fn = gnorm(F); it = 0; total = 0
            while it < snes_max_it and fn > max(snes_rtol * f0, snes_atol):
                pc.setup()                                                      # StokesPCSetUp0 after the new viscosity
                F.neg_()
                ks.solve(op, F, dx, M=pc)
                total += ks.iterations
                if ks.reason < 0:
                    fails += 1
                    if fails >= max_linear_fail or ks.reason != -3:
                        raise RuntimeError("stage (%g, %g) Newton step %d: linear solve diverged (reason %d, %d its, residual %.3e)"
                                           % (e_i, r_i, it + 1, ks.reason, ks.iterations, ks.residual))
                lam, fold = 1.0, fn
                x.add_(dx)
                op.function(x, F); fn = gnorm(F)
                while line_search and not (fn <= (1.0 - 1e-4 * lam) * fold) and lam > 1e-6:
                    x.add_(dx, alpha=-0.5 * lam); lam *= 0.5
                    op.function(x, F); fn = gnorm(F)
                it += 1
                if monitor:
                    monitor(e_i, r_i, it, fn, ks.iterations, lam)
            out.append((e_i, r_i, it, total, fn))
    finally:
        if own_ks:
            ks.destroy()
        if own_pc:
            pc.destroy()
        if stats is not None:
            stats["linear_fails"] = fails
    return out
