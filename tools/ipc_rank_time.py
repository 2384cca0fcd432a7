"""Process ranks sharing the one GPU of the box (gloo process group): a 256^3 slab matvec per rank on the IPC direct route
(Comm(ipc=True): chebhip_comm_create_ipc over the gloo callback transport) -- wall time per matvec with all ranks running, and the HOST
time a rank spends enqueueing one (two rendezvous: post, publish, barrier, one polling launch each).  The ranks time-share the device, so
the wall time is about G x one rank's kernels; what the number shows is that the route runs at full size across address spaces and what
the host side costs.  usage: python tools/ipc_rank_time.py [G ...]   (default: 2 4)"""
import os, socket, sys, time
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as ge
        sp = ge.load(); dsp = ge.load_dist()
        op = dsp.DistPoissonC((256, 256, 256), sp, ipc=True)
        n = op.local_size
        U = torch.randn(n, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
        for _ in range(20):
            op.mult(U, V)
        torch.cuda.synchronize(); dist.barrier()
        reps = 100
        t0 = time.perf_counter()
        for _ in range(reps):
            op.mult(U, V)
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / reps
        dist.barrier()
        t_host = 1e9                                 # short bursts into an empty queue: no back-pressure from the device in the host time
        for _ in range(10):
            t1 = time.perf_counter()
            for _ in range(6):
                op.mult(U, V)
            t_host = min(t_host, (time.perf_counter() - t1) / 6)
            torch.cuda.synchronize(); dist.barrier()
        q.put((rank, op.transport, t_host * 1e6, t_all * 1e6, float(V.abs().max())))
        op.destroy()
    finally:
        dist.destroy_process_group()


def main():
    for G in [int(a) for a in sys.argv[1:]] or [2, 4]:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=worker, args=(r, G, port, q)) for r in range(G)]
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
        res.sort()
        for p in procs:
            p.join(timeout=120)
        print("G = %d process ranks on one GPU, transport %s: %.1f us wall per matvec (all ranks running), host enqueue %.1f us per matvec (max over ranks)"
              % (G, res[0][1], max(r[3] for r in res), max(r[2] for r in res)), flush=True)


if __name__ == "__main__":
    main()
