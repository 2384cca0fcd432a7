"""Newton-Krylov driver on device vectors: the roles of SNESSolve and KSPSolve(KSPFGMRES) around the elliptic
callbacks (elliptic.C:177-185, 213), for end-to-end solves where no PETSc exists.

Each Newton step evaluates FormFunction (which leaves eta, eta', grad u behind, elliptic.C:498-509), then
solves J dx = -F with the matrix-free Jacobian MatMult_Elliptic (elliptic.C:297-339) by restarted FGMRES
(chebhip_fgmres_*), and updates x along dx with a backtracking line search; `M` is the slot for a right preconditioner
(the reference uses ILU(2) of a finite-difference matrix, elliptic.C:184-185, which stays PETSc's).
"""
import torch


def newton_krylov(sp, op, b, x, gamma=0.0, exponent=2.0, snes_rtol=1e-8, snes_atol=1e-50, snes_max_it=50,
                  ksp_rtol=1e-5, ksp_restart=30, ksp_max_it=10000, M=None, monitor=None, norm=None, line_search=True):
    """Solve FormFunction(x) = A(x) x - b = 0 in place in x (device tensor).  Returns (newton_its, total_ksp_its, |F|).
    On several ranks (vectors = this rank's pieces) `op` is a callable driver of dist.py, `sp.Fgmres` must return a
    solver with its reduction set, and `norm` the global 2-norm."""
    n = op.global_size
    norm = norm or (lambda t: float(t.norm()))
    F = torch.empty_like(x)
    dx = torch.empty_like(x)
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
        ks.destroy()
    return it, total, fn
