#!/usr/bin/env python3
"""One draw of tools/fuzz_dist.py looked at closely: elliptic, 4 ranks, dims (11, 5) -- one of the unresolved grids on which full Newton from
x = 0 is erratic (fuzz_dist.py docstring).  Serial Newton and the N-rank test under both Gram-Schmidt steps (krylov_exact_norm 0 / 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import __graft_entry__ as ge
    import test_gpu_dist as t
    import fuzz_dist as fz
    sp = ge.load()
    for dims, world in (((11, 5), 4), ((11, 5), 2), ((12, 6), 4)):
        for exact in (0, 1):
            sp.set_option("krylov_exact_norm", exact)
            ok_serial = fz.serial_newton_converges(dims)
            try:
                t.test_elliptic_slab_ranks_match_oracle_and_solve(world, dims); res = "ok"
            except Exception as e:
                res = "FAIL " + repr(e)[:80]
            print("dims %s world %d krylov_exact_norm=%d (serial leg only: the ranks are fresh processes with the default): serial Newton converges: %s; %d-rank test: %s" % (
                dims, world, exact, ok_serial, world, res), flush=True)
    sp.set_option("krylov_exact_norm", 0)


if __name__ == "__main__":
    main()
