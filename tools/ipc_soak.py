"""Soak of the IPC direct route: G process ranks sharing the GPU run thousands of slab matvecs back to back (no host synchronisation between
calls: the streams run ahead of each other, which is when a missing "your reads of my array have ended" wait would show), alternating between
two inputs in the SAME arrays, and count on the device every call whose result differs by a bit from the first result for that input
(the route is deterministic).  Poisson (push form: remote stores into the peers' result arrays) and the Stokes callbacks.
usage: python tools/ipc_soak.py [G] [seconds]   (default 4, 120)"""
import os, socket, sys, time
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, seconds, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as ge
        sp = ge.load(); dsp = ge.load_dist()
        comm = dsp.Comm(sp, ipc=True)
        assert comm.transport == "ipc+callback", (comm.transport, comm.ipc_error)
        out = {}
        # ---- Poisson, 130 x 72 x 66: gather loader, one-launch form, push stores
        op = dsp.DistPoissonC((130, 72, 66), sp, comm=comm)
        n = op.local_size
        g = torch.Generator(device="cuda").manual_seed(1234 + rank)
        A = torch.randn(n, dtype=torch.float64, device="cuda", generator=g); B = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        U = torch.empty_like(A); V = torch.empty_like(A)
        exp = []
        for X in (A, B):
            U.copy_(X); op.mult(U, V); torch.cuda.synchronize(); dist.barrier(); exp.append(V.clone())
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        calls, t0 = 0, time.perf_counter()
        budget = torch.zeros(1)
        while True:
            for _ in range(200):
                k = calls & 1
                U.copy_(A if k == 0 else B)                  # rewritten right after the previous call: legal once that call has returned
                op.mult(U, V)
                bad += (V != exp[k]).any().to(torch.int64)
                calls += 1
            budget[0] = 1.0 if time.perf_counter() - t0 > seconds / 2 else 0.0
            dist.all_reduce(budget, op=dist.ReduceOp.MAX)    # every rank stops after the same number of calls
            if budget[0] > 0:
                break
        torch.cuda.synchronize()
        out["poisson"] = (calls, int(bad.item()))
        op.destroy()
        # ---- Stokes callbacks, 70 x 36 x 34 power law: two gather launches per callback, fields pushed into the peers' receive buffers
        D = dsp.DistStokesC((70, 36, 34), sp, comm=comm)
        D.op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
        D.op.set_dirichlet(np.zeros(D.dirichlet_size)); D.op.set_force(np.zeros(D.global_size))
        m = D.global_size
        A = torch.randn(m, dtype=torch.float64, device="cuda", generator=g); B = torch.randn(m, dtype=torch.float64, device="cuda", generator=g)
        x = torch.empty_like(A); y = torch.empty_like(A)
        x.copy_(A); D.function(x, y); torch.cuda.synchronize(); dist.barrier()          # fixes the state the Jacobian is linearised at
        exp = []
        for X in (A, B):
            x.copy_(X); D.mult(x, y); torch.cuda.synchronize(); dist.barrier(); exp.append(y.clone())
        bad.zero_(); calls, t0 = 0, time.perf_counter()
        while True:
            for _ in range(100):
                k = calls & 1
                x.copy_(A if k == 0 else B)
                D.mult(x, y)
                bad += (y != exp[k]).any().to(torch.int64)
                calls += 1
            budget[0] = 1.0 if time.perf_counter() - t0 > seconds / 2 else 0.0
            dist.all_reduce(budget, op=dist.ReduceOp.MAX)
            if budget[0] > 0:
                break
        torch.cuda.synchronize()
        out["stokes_mult"] = (calls, int(bad.item()))
        D.destroy(); comm.destroy()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=worker, args=(r, G, port, seconds, q)) for r in range(G)]
    for p in procs:
        p.start()
    res = []
    while len(res) < G:
        try:
            res.append(q.get(timeout=2))
        except Exception:
            if any(p.exitcode not in (None, 0) for p in procs):
                for p in procs:
                    p.kill()
                raise SystemExit("a rank failed")
    for p in procs:
        p.join(timeout=120)
    res.sort()
    fails = 0
    for kind in ("poisson", "stokes_mult"):
        calls = res[0][1][kind][0]; bad = sum(r[1][kind][1] for r in res)
        fails += bad
        print("ipc soak, %d process ranks, %s: %d calls per rank back to back, results differing from the first one for their input: %d" % (G, kind, calls, bad), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
