#!/usr/bin/env python3
"""Stress run of the direct-transport routes (thread ranks reading each other's arrays in place): the tests that exercise them, repeated, so
that an ordering bug between rank threads (events, rendezvous tables, collective destroy) would have many chances to show.
usage: r06_stress_direct.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_dist_emul as em
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    t0 = time.time(); n = 0
    cases = [
        lambda: em.test_poisson_direct_pull_equals_packed_exchange_to_the_bit(),
        lambda: em.test_poisson_thread_ranks_small(8, (70, 68, 66), 0),
        lambda: em.test_poisson_thread_ranks_small(3, (21, 16, 11), 0),
        lambda: em.test_poisson_batch_thread_ranks(8, (70, 68, 66), 2),
        lambda: em.test_stokes_thread_ranks_read_the_peers_fields_in_place(3),
        lambda: em.test_stokes_thread_ranks_read_the_peers_fields_in_place(4),
        lambda: em.test_elliptic_general_thread_ranks((72, 40, 34), 3),
        lambda: em.test_poisson_256_over_8_ranks(),
    ]
    while time.time() - t0 < budget:
        for c in cases:
            c(); n += 1
        print("... %d cases, %.0f s" % (n, time.time() - t0), flush=True)
    print("stress (direct transports): %d cases in %.0f s, all passed" % (n, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
