"""One rank's compute side of the slab-mode Stokes callbacks at 128^3 (power law), NULL transport: StokesFunction / StokesMatMult
per rank count and exchange kind (option dist_packed_exchange: 0 = in-place gather on direct transports, 1 = packed segments).
usage: python tools/stokes_rank.py [packed ...]   (default: 0)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as ge


def t_us(fn, reps=60):
    t0 = time.perf_counter(); n = 0
    while n < 15 or time.perf_counter() - t0 < 0.03:
        fn(); n += 1
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[1]


def main():
    sp = ge.load(); dsp = ge.load_dist()
    for packed in [int(a) for a in sys.argv[1:]] or [0]:
        sp.set_option("dist_packed_exchange", packed)
        for G in (8, 4, 2):
            comm = dsp.Comm(sp, null=(G, 0))
            D = dsp.DistStokesC((128, 128, 128), sp, comm=comm)
            D.op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
            D.op.set_dirichlet(np.zeros(D.dirichlet_size)); D.op.set_force(np.zeros(D.global_size))
            x = torch.randn(D.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
            print("packed=%d G=%d: StokesFunction %.1f us  StokesMatMult %.1f us"
                  % (packed, G, t_us(lambda: D.function(x, y)), t_us(lambda: D.mult(x, y))), flush=True)
            D.destroy(); comm.destroy()
    sp.set_option("dist_packed_exchange", 0)


if __name__ == "__main__":
    main()
