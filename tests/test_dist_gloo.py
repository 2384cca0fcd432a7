"""World-size-2 (and 3) CPU tests of the slab-partitioned Poisson matvec: the exchange logic of
spectral-petsc_amd/dist.py under the gloo backend, with the local arithmetic supplied by the CPU
oracle (test-only backend).  The serial reference has no multi-rank code; the requirement is that
any G reproduces the G = 1 answer to rounding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge
import oracle_lib as orc


class OracleBackend:
    """Test-only stand-in for HipBackend: D_k D_k on zero-extended lines via the CPU oracle."""
    device = torch.device("cpu")

    def lap1d(self, x, shape, axis, out, acc=None, alpha=1.0):
        a = x.numpy().reshape(shape)
        pad = [(0, 0)] * len(shape)
        pad[axis] = (1, 1)
        full = np.pad(a, pad)
        g = orc.cheb_mult(full, axis, orc.DIRECT)
        t = orc.cheb_mult(g, axis, orc.DIRECT)
        sl = [slice(None)] * len(shape)
        sl[axis] = slice(1, -1)
        res = alpha * t[tuple(sl)].reshape(-1)
        if acc is not None:
            res = acc.numpy() + res
        out.copy_(torch.from_numpy(np.ascontiguousarray(res)))
        return out

    def side_stream(self):
        return None

    def pack(self, slab, buf, m0, M1, R, c1):
        a = slab.numpy().reshape(m0, M1, R)
        buf.copy_(torch.from_numpy(np.concatenate([a[:, c1[s]:c1[s + 1], :].ravel() for s in range(len(c1) - 1)])))
        return buf

    def unpack_add(self, buf, acc, out, m0, M1, R, c1):
        b = buf.numpy()
        t = np.empty((m0, M1, R))
        off = 0
        for s in range(len(c1) - 1):
            w = c1[s + 1] - c1[s]
            t[:, c1[s]:c1[s + 1], :] = b[off:off + m0 * w * R].reshape(m0, w, R)
            off += m0 * w * R
        res = t.reshape(-1) + (acc.numpy() if acc is not None else 0.0)
        out.copy_(torch.from_numpy(np.ascontiguousarray(res)))
        return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dims, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dsp = ge.load_dist()
        op = dsp.DistPoissonOp(dims, OracleBackend())
        U = op.random_input(20240229)
        V = torch.empty_like(U)
        op.mult(U, V)
        op.mult(U, V)   # second call: buffers are reusable
        q.put((rank, int(op.s0[rank]), V.numpy().copy(), U.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims", [(2, (12, 11, 10)), (3, (9, 10)), (2, (8, 9, 5, 4)), (8, (20, 19, 7)), (4, (6, 9, 4))])
def test_slab_matvec_matches_serial(world, dims):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[1])
    V = np.concatenate([r[2] for r in res])
    U = np.concatenate([r[3] for r in res])
    # every rank count sees the same global input
    g = torch.Generator(device="cpu").manual_seed(20240229)
    Uref = torch.randn(int(np.prod([d - 2 for d in dims])), dtype=torch.float64, generator=g).numpy()
    assert np.array_equal(U, Uref)
    ref = orc.elliptic_mult(dims, Uref, mode=orc.DIRECT)
    assert np.linalg.norm(V - ref) / np.linalg.norm(ref) < 1e-12


def test_split_sizes():
    dsp = ge.load_dist()
    assert dsp.split_sizes(254, 8) == [32] * 6 + [31] * 2
    assert sum(dsp.split_sizes(7, 3)) == 7


def test_serial_ranges_against_brute_force():
    """Where a rank's pieces sit in the serial vectors (slab-mode drivers): closed form vs counting nodes."""
    dsp = ge.load_dist()
    rng = np.random.default_rng(5)
    for _ in range(200):
        d = int(rng.integers(2, 5))
        dims = tuple(int(v) for v in rng.integers(3, 9, size=d))
        G = int(rng.integers(1, dims[0] + 1))
        m0 = dsp.split_sizes(dims[0], G)
        s0 = [0] + list(np.cumsum(m0))
        inner = list(np.ndindex(*dims[1:]))
        is_b = lambda i0, rest: i0 in (0, dims[0] - 1) or any(r in (0, n - 1) for r, n in zip(rest, dims[1:]))
        for r in range(G):
            stub = type("S", (), {"dims": dims, "s0": [int(v) for v in s0], "rank": r})()
            (n0, n1), (b0, b1) = dsp._SlabPencil.serial_ranges(stub)
            cnt = lambda lo, hi, want: sum(1 for i0 in range(lo, hi) for rest in inner if is_b(i0, rest) == want)
            assert n0 == cnt(0, s0[r], False) and n1 == cnt(0, s0[r + 1], False), (dims, G, r)
            assert b0 == cnt(0, s0[r], True) and b1 == cnt(0, s0[r + 1], True), (dims, G, r)


# ---- the slab <-> pencil exchange of the slab-mode drivers (DistEllipticOp / DistStokesOp), on CPU tensors ----------
class _CpuCopies:
    """Stand-in for the three library calls the exchange machinery makes (pack, unpack, view), on CPU tensors."""
    def __init__(self):
        self.keep = {}

    def register(self, t):
        self.keep[t.data_ptr()] = t
        return t.data_ptr()

    def device_view(self, ptr, n):
        t = self.keep[ptr]
        assert t.numel() == n
        return t

    def slab_pack(self, slab, buf, m0, M1, R, c1):
        return OracleBackend().pack(slab, buf, m0, M1, R, c1)

    def slab_unpack_add(self, buf, acc, out, m0, M1, R, c1, alpha=1.0):
        return OracleBackend().unpack_add(buf * alpha, acc, out, m0, M1, R, c1)


def _exchange_worker(rank, world, port, dims, nf, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dsp = ge.load_dist()
        sp = _CpuCopies()
        ex = dsp._SlabPencil()
        ex._setup(dims, sp, None, nf)
        P0, P1 = dims[0], dims[1]
        R = int(np.prod(dims[2:])) if len(dims) > 2 else 1
        full = np.arange(nf * P0 * P1 * R, dtype=np.float64).reshape(nf, P0, P1, R)     # field f, node (i0, i1, r)
        lo, hi = ex.s0[rank], ex.s0[rank + 1]
        slab = torch.from_numpy(np.ascontiguousarray(full[:, lo:hi]).reshape(-1))
        ex._to_pencil(nf, sp.register(slab))
        c0, c1 = ex.s1[rank], ex.s1[rank + 1]
        want = np.ascontiguousarray(full[:, :, c0:c1]).reshape(-1)
        ok_fwd = np.array_equal(ex.pen_in.numpy()[:want.size], want)
        ex._pencil = lambda kind, n: ex.pen_out.copy_(ex.pen_in)                       # identity on the pencil
        acc = torch.ones_like(slab); out = torch.empty_like(slab)
        ex._dim0(0, nf, sp.register(slab), sp.register(acc), -2.0, sp.register(out), None)
        ok_back = np.array_equal(out.numpy(), 1.0 - 2.0 * slab.numpy())
        q.put((rank, ok_fwd, ok_back))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,nf", [(2, (7, 6, 5), 3), (3, (8, 9), 1), (4, (5, 7, 3), 2)], ids=str)
def test_slab_pencil_round_trip(world, dims, nf):
    """Every field of a slab reaches the pencils in the right place, and the way back restores it (with the AXPY)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, dims, nf, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res), res


# ---- the IPC transport's refusal path (csrc/comm.hip): no GPU here, so the shared segment cannot be registered with a device ----
def _ipc_refusal_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        sp = ge.load(); dsp = ge.load_dist()
        t0 = time.perf_counter()
        comm = dsp.Comm(sp, ipc=True)
        q.put((rank, comm.transport, comm.ipc_error, time.perf_counter() - t0, os.path.exists("/dev/shm") and [f for f in os.listdir("/dev/shm") if f.startswith("chebhip-")]))
        comm.destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_transport_falls_back_on_every_rank_alike_when_the_node_refuses(world):
    """Comm(ipc=True) where the direct route cannot be had (here: no device to register the shared segment with): every rank keeps the
    message transport, says why, does so within seconds (a rank that fails marks the segment, the others do not search a vanished
    name until their time limit), and the shared-memory name is gone afterwards."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the node grants the IPC group (tests/test_gpu_dist.py covers that side)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ipc_refusal_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] == "callback" and r[2] for r in res), res
    assert max(r[3] for r in res) < 30.0, res
    assert not any(r[4] for r in res), res
