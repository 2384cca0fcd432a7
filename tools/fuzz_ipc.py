#!/usr/bin/env python3
"""Randomised run of the IPC direct route (process ranks sharing the box's GPU, csrc/comm.hip chebhip_comm_create_ipc over the gloo
callback transport): random world sizes and grid shapes -- small ones (pull route / segment route under the IPC communicator) and ones
with 66..130-point lines (gather loader, one-launch form, in-place pencil sweeps of the Stokes / elliptic drivers) -- through
tests/test_gpu_dist.py's process-group cases against the oracle.  Not part of the suite; run on the GPU box:

    python tools/fuzz_ipc.py [seconds] [seed]  > gpurun_out/fuzz_ipc.log

Every case prints `ok ...` or `FAIL ...` with the traceback; the exit code is the number of failures."""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dist as t

SMALL = list(range(6, 15))
LONG = [34, 40, 66, 68, 70, 72, 96, 130]


def draw_dims(rng, world, d):
    while True:
        dims = tuple(int(rng.choice(LONG if rng.random() < 0.45 else SMALL)) for _ in range(d))
        if dims[0] - 2 >= world and dims[1] - 2 >= world and int(np.prod(dims)) <= 450000:
            return dims


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time(); nfail = 0; ncase = 0
    while time.time() - t0 < budget:
        kind = int(rng.integers(0, 2))
        world = int(rng.integers(2, 5))
        dims = draw_dims(rng, world, 3 if kind == 1 or rng.random() < 0.7 else 2)
        try:
            if kind == 0:
                t.test_dist_c_ranks_match_oracle(world, dims, "gloo-ipc")
                tag = "poisson"
            else:
                t.test_slabx_c_drivers_over_process_group(world, dims, "gloo-ipc")
                tag = "stokes+elliptic"
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
