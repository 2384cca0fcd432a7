#!/usr/bin/env python3
"""Randomised hardening run of the multi-rank GPU tests (tests/test_gpu_dist.py): random world sizes and grid
shapes for the Poisson, general-coefficient and Stokes slab drivers.  Not part of the suite; run on the GPU box:

    python tools/fuzz_dist.py [seconds] [seed]  > gpurun_out/fuzz_dist.log

Every case prints `ok ...` or `FAIL ...` with the traceback; the exit code is the number of failures.
History: round 1's run logged `FAIL 1 3 (12, 7)` -- full-step Newton from x = 0 does not contract on that
unresolved grid, so the comparison with a 15-step dense Newton was ill-posed (fixed in the test: the
distributed solution is checked as a root of the oracle's residual; see tests/test_gpu_dist.py)."""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dist as t

def serial_newton_converges(dims):
    """The acceptance solve of the elliptic case (Newton with backtracking from x = 0, gamma = 4, cos_scale = 3) stalls at a
    local minimum of |F| on some unresolved grids -- (11, 5), (7, 9), (5, 9, 7), (7, 6, 9), (7, 7, 6) ... -- on ONE GPU too;
    such draws say nothing about the slab code and are skipped."""
    import torch
    from importlib import import_module
    import __graft_entry__ as ge
    import oracle_lib as orc
    sp = ge.load(); solve = import_module(sp.__name__ + ".solve")
    u, u2, dv = orc.elliptic_exact(dims, 0, gamma=4.0, exponent=2.0, cos_scale=3.0)
    op = sp.EllipticOp(dims); op.set_dirichlet(dv)
    b = torch.from_numpy(u2.copy()).cuda(); x = torch.zeros_like(b)
    G = int(np.prod([v - 2 for v in dims]))
    # Round 6: on such a grid the serial outcome itself flips with the rounding of the Krylov solves ((11, 5): stalls with the three-launch
    # Gram-Schmidt step, converges with the one-reduction step, and the 2- and 4-rank runs stall: tools/r06_case_11_5.py) -- the draw is kept
    # only if the serial Newton converges with BOTH steps
    ok = True
    for exact in (0, 1):
        sp.set_option("krylov_exact_norm", exact)
        try:
            x.zero_()
            its, kits, fn = solve.newton_krylov(sp, op, b, x, 4.0, 2.0, snes_rtol=1e-11, ksp_rtol=1e-12, ksp_restart=min(256, G), ksp_max_it=20000, snes_max_it=100)
        finally:
            sp.set_option("krylov_exact_norm", 0)
        ok = ok and fn <= 1e-9 * float(np.linalg.norm(u2))
    op.destroy()
    return ok


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time(); nfail = 0; ncase = 0
    while time.time() - t0 < budget:
        kind = int(rng.integers(0, 3))
        world = int(rng.integers(2, 5))
        try:
            if kind == 0:
                d = int(rng.integers(2, 4))
                dims = tuple(int(v) for v in rng.integers(max(world + 2, 5), 14, size=d))
                t.test_distributed_poisson_solve(world, dims)
                tag = "poisson"
            elif kind == 1:
                d = int(rng.integers(2, 4))
                dims = tuple(int(v) for v in rng.integers(max(world + 1, 4), 13 if d == 2 else 10, size=d))
                if not serial_newton_converges(dims):
                    print("skip elliptic", world, dims, "(the serial Newton stalls on this grid too)", flush=True); continue
                t.test_elliptic_slab_ranks_match_oracle_and_solve(world, dims)
                tag = "elliptic"
            else:
                dims = tuple(int(v) for v in rng.integers(max(world + 1, 5), 10, size=3))
                t.test_slab_ranks_match_oracle(world, dims)
                tag = "stokes"
            print("ok", tag, world, dims, flush=True)
        except Exception as e:                                   # noqa: BLE001 -- a fuzz driver reports and goes on
            nfail += 1
            print("FAIL", kind, world, dims, repr(e), flush=True)
            traceback.print_exc()
        ncase += 1
        if nfail >= 5:
            break
    print("cases %d failures %d in %.0f s" % (ncase, nfail, time.time() - t0))
    sys.exit(min(nfail, 100))


# (the rank processes are spawned: they import this file again, and must not start a fuzz run of their own)
if __name__ == "__main__":
    main()
