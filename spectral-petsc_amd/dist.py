"""Slab-partitioned linear Poisson matvec across G ranks (one process per GPU, RCCL over xGMI).

The reference is strictly serial (elliptic.C:262 VecCreateSeq; nk.c:63 refuses size != 1), so
there is no reference behaviour to copy here except the answer: for any G the result equals the
G = 1 vector to rounding.

Layout.  Everything lives in the interior layout of the reference's global vector
(M0, M1, ..) = dims - 2, row-major (SetupBC, elliptic.C:372-434).  Rank r owns the slab of
interior planes [s0[r], s0[r+1]) along dim 0.  V = -(L0 + L1 + ... ) U with L_k = D_k D_k on
zero-extended lines (MatMult_Elliptic with eta = 1, deta = 0, elliptic.C:297-339):

  local      W   = -L1 U - L2 U - ...     fused launches on the slab           (cheb_apply_lap1d)
  exchange   UT  = all-to-all(U)          slab (m0, M1, R) -> pencil (M0, m1, R), split along dim 1
  pencil     TT  = -L0 UT                 one fused launch on the pencil
  exchange   T   = all-to-all(TT)         back to the slab
  combine    V   = W + T

The two exchanges are the only collectives (torch.distributed all_to_all_single = RCCL
all-to-all; each GPU sends 7 direct xGMI messages).  The forward exchange overlaps the first
local launches (separate HIP stream).  The local terms are kept in arrays of their own and summed
with the exchanged term in the serial order k = 0, 1, 2 (elliptic.C:331-334).

The product driver is DistPoissonC below: the same algorithm in C++ behind the C ABI (csrc/dist.hip,
chebhip_dist_*) with RCCL as its transport.  DistPoissonOp is its Python twin: the exchange logic with the
local arithmetic delegated to a backend, which lets tests/ run it under gloo on CPU with an oracle backend.

The local arithmetic is delegated to a backend: HipBackend (the product: C-ABI calls on device
tensors).  tests/ supplies an oracle-based CPU backend to exercise the exchange logic under gloo.
"""
import os
import weakref

import numpy as np
import torch
import torch.distributed as dist


def _torch_stream(ptr):
    """The torch stream object of a hipStream_t handed to a callback.  NULL is the device's default stream:
    torch.cuda.ExternalStream(0) is NOT (work queued through it is not ordered with the default stream's kernels --
    found as a nondeterministic Krylov solve when the exchange chain moved to the caller's stream)."""
    p = int(ptr or 0)
    return torch.cuda.default_stream() if p == 0 else torch.cuda.ExternalStream(p)


def split_sizes(n, parts):
    """Near-equal contiguous split of n planes over `parts` ranks (first n % parts get one more)."""
    q, r = divmod(n, parts)
    return [q + (1 if i < r else 0) for i in range(parts)]


class HipBackend:
    """Local arithmetic on device tensors through libchebhip.so (cheb_apply_lap1d)."""

    def __init__(self, sp):
        self.sp = sp
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._plans = {}

    def lap1d(self, x, shape, axis, out, acc=None, alpha=1.0):
        key = (tuple(shape), axis)
        if key not in self._plans:
            self._plans[key] = self.sp.Lap1dPlan(shape, axis)
        return self._plans[key].apply(x, out, acc, alpha)

    def side_stream(self):
        return torch.cuda.Stream()

    def pack(self, slab, buf, m0, M1, R, c1):
        return self.sp.slab_pack(slab, buf, m0, M1, R, c1)

    def unpack_add(self, buf, acc, out, m0, M1, R, c1):
        return self.sp.slab_unpack_add(buf, acc, out, m0, M1, R, c1)


class DistPoissonOp:
    def __init__(self, dims, backend, group=None, force_a2a=False, serial=False):
        """force_a2a: run the collectives even with one rank (rehearsals of the real backend); serial: everything on the
        caller's stream (no overlap of the local sweeps with the exchanges, but no cross-stream dependencies either)."""
        assert len(dims) >= 2, "slab partitioning needs at least two dimensions"
        self.dims = tuple(int(d) for d in dims)
        self.M = tuple(d - 2 for d in self.dims)
        assert min(self.M) >= 1
        self.backend = backend
        self.group = group
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        G, M = self.G, self.M
        if M[0] < G or M[1] < G:
            raise ValueError("slab partition of interior extents %s over %d ranks: every rank needs at least one plane "
                             "along dims 0 and 1" % (M[:2], G))
        self.R = int(np.prod(M[2:])) if len(M) > 2 else 1          # trailing dims flattened
        self.m0 = split_sizes(M[0], G)
        self.m1 = split_sizes(M[1], G)
        self.s0 = np.concatenate([[0], np.cumsum(self.m0)])
        self.s1 = np.concatenate([[0], np.cumsum(self.m1)])
        r = self.rank
        self.slab_shape = (self.m0[r],) + M[1:]
        self.pencil_shape = (M[0], self.m1[r]) + M[2:]
        self.local_size = int(np.prod(self.slab_shape))
        self.pencil_size = int(np.prod(self.pencil_shape))
        dev = backend.device
        # forward exchange: slab -> send buffer ordered by destination rank s = block U[:, s1[s]:s1[s+1], :]
        # (backend.pack / backend.unpack_add: one pass each, no index tables)
        self.c1 = [int(v) for v in self.s1]
        self.fwd_send = [self.m0[r] * self.m1[s] * self.R for s in range(G)]
        self.fwd_recv = [self.m0[s] * self.m1[r] * self.R for s in range(G)]   # lands as the pencil, no unpack
        # backward exchange: pencil rows s0[s]:s0[s+1] are contiguous -> no pack; unpack fused with the final sum
        self.W = torch.empty(self.local_size, dtype=torch.float64, device=dev)
        self.sendbuf = torch.empty(self.local_size, dtype=torch.float64, device=dev)
        self.UT = torch.empty(self.pencil_size, dtype=torch.float64, device=dev)
        self.TT = torch.empty(self.pencil_size, dtype=torch.float64, device=dev)
        forced = self.force_a2a = bool(force_a2a) and dist.is_initialized()
        self.comm_stream = backend.side_stream() if ((G > 1 or forced) and not serial) else None
        self.A = []

    # ---- helpers -------------------------------------------------------------------------------
    def random_input(self, seed):
        """Rank-local slab of the global N(0,1) vector: every G sees the same global field."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        full = torch.randn(int(np.prod(self.M)), dtype=torch.float64, generator=g).reshape(self.M[0], -1)
        r = self.rank
        return full[self.s0[r]:self.s0[r + 1]].reshape(-1).contiguous().to(self.backend.device)

    def _a2a(self, out, inp, out_split, in_split):
        if self.G == 1 and not self.force_a2a:
            out.copy_(inp)
        elif inp.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks sharing one GPU, BENCH_DIST_BACKEND=gloo): stage through the host
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(ho, inp.cpu(), out_split, in_split, group=self.group)
            out.copy_(ho)
        else:
            dist.all_to_all_single(out, inp, out_split, in_split, group=self.group)

    # ---- the matvec ----------------------------------------------------------------------------
    def mult(self, U, V):
        """V = ((T_0 + A_1) + A_2) + ..: the local terms A_k = -L_k U go to arrays of their own (they overlap both
        exchanges) and are added to the exchanged term T_0 = -L_0 U in the serial order k = 0, 1, 2 (elliptic.C:331-334)."""
        be, M = self.backend, self.M
        d = len(M)
        cs = self.comm_stream
        if len(self.A) < d - 1:
            self.A = [torch.empty(self.local_size, dtype=torch.float64, device=be.device) for _ in range(d - 1)]

        def exchange_chain():
            be.pack(U, self.sendbuf, self.m0[self.rank], M[1], self.R, self.c1)
            self._a2a(self.UT, self.sendbuf, self.fwd_recv, self.fwd_send)
            be.lap1d(self.UT, self.pencil_shape, 0, self.TT, None, -1.0)
            self._a2a(self.sendbuf, self.TT, self.fwd_send, self.fwd_recv)       # back: roles of the splits swap
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                exchange_chain()
        else:
            exchange_chain()
        for k in range(1, d):                                                    # overlap both exchanges
            be.lap1d(U, self.slab_shape, k, self.A[k - 1], None, -1.0)
        if cs is not None:
            torch.cuda.current_stream().wait_stream(cs)
        be.unpack_add(self.sendbuf, None, self.W, self.m0[self.rank], M[1], self.R, self.c1)   # W = T_0
        for k in range(1, d):
            out = V if k == d - 1 else self.W
            torch.add(self.W, self.A[k - 1], out=out)                           # (T_0 + A_1) + A_2 ...
        return V


class DistPoissonC:
    """The slab driver that lives behind the C ABI (csrc/dist.hip: chebhip_dist_*): partition, pack, the two
    exchanges, pencil launch, overlap on a side stream and the fixed-order sum are all C++; this class only hands it
    a transport -- an RCCL communicator made from a unique id that rank 0 broadcasts through the process group
    (backend "nccl"), or, for rehearsals on one GPU under gloo, a callback that stages the exchange through the host."""

    def __init__(self, dims, sp, group=None, comm=None, force_a2a=False, legacy_exchange=False, ipc=False):
        """ipc: the direct route among the processes of one node on top of the group's transport (Comm(ipc=True); `transport`
        tells whether the node granted it).  comm: a Comm (e.g. one rank of a LocalGroup) instead of the process group's transport; force_a2a: go through
        the transport even with one rank (one-rank rehearsals of the real backend); legacy_exchange: under gloo, plug the
        host staging in as a chebhip_exchange_fn (chebhip_dist_set_exchange: one vector per exchange, every block moved)
        instead of a chebhip_comm callback transport (which also carries chebhip_dist_mult_batch)."""
        import ctypes as C
        self.sp, self.group = sp, group
        self.dims = tuple(int(v) for v in dims)
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if comm is not None:
            self.G, self.rank = comm.G, comm.rank
        L = sp.lib()
        h = C.c_void_p()
        sp._chk(L.chebhip_dist_create(len(self.dims), (C.c_int * len(self.dims))(*self.dims), self.G, self.rank, C.byref(h)))
        self._h = h
        self.local_size = L.chebhip_dist_local_size(h)
        self.slab_offset = L.chebhip_dist_slab_offset(h)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._comm = None
        self._cb = None
        self._own_comm = None
        forced = bool(force_a2a) and dist.is_initialized()
        self.transport = comm.transport if comm is not None else "none"
        if comm is not None:
            sp._chk(L.chebhip_dist_use_comm(h, comm._h))
        elif self.G > 1 and ipc and not legacy_exchange:
            self._own_comm = Comm(sp, group=group, ipc=True)
            self.transport = self._own_comm.transport
            sp._chk(L.chebhip_dist_use_comm(h, self._own_comm._h))
        elif self.G > 1 or forced:
            self.transport = "rccl" if dist.get_backend(group) == "nccl" else "callback"
            if dist.get_backend(group) == "nccl":
                idbuf = C.create_string_buffer(128)
                if self.rank == 0:
                    sp._chk(L.chebhip_rccl_unique_id(idbuf))
                box = [bytes(idbuf.raw)]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                comm = C.c_void_p()
                sp._chk(L.chebhip_rccl_comm_create(self.G, self.rank, C.create_string_buffer(box[0], 128), C.byref(comm)))
                self._comm = comm
                sp._chk(L.chebhip_dist_use_rccl(h, comm))
            elif legacy_exchange:
                self._cb = self._host_exchange()
                sp._chk(L.chebhip_dist_set_exchange(h, C.cast(self._cb, C.c_void_p), None))
            else:
                self._own_comm = Comm(sp, group=group)       # gloo: chebhip_comm callback transport staged through the host
                sp._chk(L.chebhip_dist_use_comm(h, self._own_comm._h))

    def _host_exchange(self):
        """chebhip_exchange_fn under gloo: device -> host, all_to_all_single, host -> device, ordered on `stream`."""
        import ctypes as C
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_void_p, C.POINTER(C.c_long), C.c_void_p)
        G, sp, group = self.G, self.sp, self.group

        def xfn(ctx, send, sc, recv, rc, stream):
            try:
                scl = [int(sc[i]) for i in range(G)]; rcl = [int(rc[i]) for i in range(G)]
                ext = _torch_stream(stream)
                with torch.cuda.stream(ext):
                    hs = sp.device_view(send, max(sum(scl), 1))[:sum(scl)].cpu()          # synchronises with `stream`
                    hr = torch.empty(sum(rcl), dtype=torch.float64)
                    dist.all_to_all_single(hr, hs, rcl, scl, group=group)
                    if sum(rcl):
                        sp.device_view(recv, sum(rcl)).copy_(hr)
                    ext.synchronize()
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 5
        return FN(xfn)

    def random_input(self, seed):
        """Rank-local slab of the global N(0,1) vector: every G sees the same global field."""
        M = [v - 2 for v in self.dims]
        g = torch.Generator(device="cpu").manual_seed(seed)
        full = torch.randn(int(np.prod(M)), dtype=torch.float64, generator=g)
        return full[self.slab_offset:self.slab_offset + self.local_size].contiguous().to(self.device)

    def mult(self, U, V):
        sp = self.sp
        sp._chk(sp.lib().chebhip_dist_mult(self._h, sp._dev_ptr(U, self.local_size), sp._dev_ptr(V, self.local_size), sp._stream()))
        return V

    def mult_batch(self, U, V):
        """chebhip_dist_mult_batch: U, V of shape (nrhs, local_size), contiguous -- nrhs vectors through ONE exchange each way."""
        sp = self.sp
        nrhs = int(U.shape[0])
        assert U.dim() == 2 and U.shape[1] == self.local_size and V.shape == U.shape and U.is_contiguous() and V.is_contiguous()
        sp._chk(sp.lib().chebhip_dist_mult_batch(self._h, nrhs, sp._dev_ptr(U.view(-1), nrhs * self.local_size), sp._dev_ptr(V.view(-1), nrhs * self.local_size), sp._stream()))
        return V

    def reduce_fn(self):
        """(chebhip_reduce_fn, ctx) completing Krylov inner products over the ranks: ncclAllReduce on the C side."""
        import ctypes as C
        if self._comm is not None:
            return C.cast(self.sp.lib().chebhip_rccl_reduce, C.c_void_p), self._comm
        if self._own_comm is not None and self._own_comm.transport.endswith("rccl"):
            return self._own_comm.reduce_fn()
        return None, None

    def destroy(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize()
            self.sp.lib().chebhip_dist_destroy(self._h)
            self._h = None
        if getattr(self, "_comm", None):
            self.sp.lib().chebhip_rccl_comm_destroy(self._comm)
            self._comm = None
        if getattr(self, "_own_comm", None):
            self._own_comm.destroy()
            self._own_comm = None


class _SlabPencil:
    """Slab <-> pencil machinery shared by the slab-mode drivers: the grid's dimension 0 is split into slabs of
    planes, one per rank; work along dimension 0 is done on pencils (all planes, a share of dimension 1):

        slab fields --pack, exchange--> pencil fields --pencil launch--> pencil result --exchange, unpack--> slab

    One exchange moves all fields of a call as a single batch of point-to-point messages (RCCL groups them into
    one launch).  The full local grid (boundary planes included) is split: Dirichlet rows and, for Stokes, the
    pressure end points take part in the sweeps."""

    def _setup(self, dims, sp, group, nf_max):
        self.sp = sp
        self.dims = tuple(int(v) for v in dims)
        d = self.d = len(self.dims)
        self.group = group
        self.G = G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = r = dist.get_rank(group) if dist.is_initialized() else 0
        P0, P1 = self.dims[0], self.dims[1]
        self.R = int(np.prod(self.dims[2:])) if d > 2 else 1
        if P0 < G or P1 < G:
            raise ValueError("slab partition of extents %s over %d ranks: every rank needs a plane along dims 0 and 1" % (self.dims[:2], G))
        self.m0 = split_sizes(P0, G); self.m1 = split_sizes(P1, G)
        self.s0 = [int(v) for v in np.concatenate([[0], np.cumsum(self.m0)])]
        self.s1 = [int(v) for v in np.concatenate([[0], np.cumsum(self.m1)])]
        # (CPU tensors only in the exchange tests of tests/test_dist_gloo.py, which supply their own pack / unpack)
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.Ns = self.m0[r] * P1 * self.R                     # nodes of the slab
        self.ncol = self.m1[r] * self.R                        # lines of the pencil
        self.Np = P0 * self.ncol
        z = lambda n: torch.empty(n, dtype=torch.float64, device=self.device)
        self.sendbuf, self.recvbuf = z(nf_max * self.Ns), z(nf_max * self.Ns)
        self.pen_in, self.pen_out = z(nf_max * self.Np), z(nf_max * self.Np)

    # where this rank's pieces sit in the serial vectors (dimension 0 outermost => contiguous): node ranges
    def serial_ranges(self):
        inner_int = int(np.prod([v - 2 for v in self.dims[1:]]))
        inner_all = int(np.prod(self.dims[1:]))
        lo, hi = self.s0[self.rank], self.s0[self.rank + 1]
        P0 = self.dims[0]
        ilo, ihi = max(lo, 1) - 1, max(min(hi, P0 - 1) - 1, max(lo, 1) - 1)     # interior planes before / up to this slab
        def bnodes(plane_hi):      # boundary nodes in planes [0, plane_hi)
            full = min(plane_hi, 1) + max(plane_hi - (P0 - 1), 0)
            return full * inner_all + (plane_hi - full) * (inner_all - inner_int)
        return (ilo * inner_int, ihi * inner_int), (bnodes(lo), bnodes(hi))

    # ---- exchanges: every field of a call in one batch of point-to-point messages ------------------------------
    def _exchange(self, sends, recvs):
        """sends / recvs: per peer s a list of contiguous tensors (views)."""
        for a, b in zip(sends[self.rank], recvs[self.rank]):       # own block: no message
            b.copy_(a)
        if self.G == 1:
            return
        staged = sends[0][0].is_cuda and dist.get_backend(self.group) == "gloo"      # rehearsal: several ranks on one GPU
        ops, back = [], []
        for s in range(self.G):
            if s == self.rank:
                continue
            for t in recvs[s]:
                if t.numel():
                    buf = torch.empty(t.shape, dtype=t.dtype) if staged else t
                    if staged:
                        back.append((t, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, s, self.group))
            for t in sends[s]:
                if t.numel():
                    ops.append(dist.P2POp(dist.isend, t.cpu() if staged else t, s, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for t, buf in back:
            t.copy_(buf)

    def _to_pencil(self, nf, inp):
        """nf slab fields at device address inp -> self.pen_in as (nf, P0, m1, R)."""
        r, P1, R = self.rank, self.dims[1], self.R
        src = self.sp.device_view(inp, nf * self.Ns)
        for f in range(nf):
            self.sp.slab_pack(src[f * self.Ns:(f + 1) * self.Ns], self.sendbuf[f * self.Ns:(f + 1) * self.Ns], self.m0[r], P1, R, self.s1)
        sends, recvs = [], []
        for s in range(self.G):
            blk = self.m0[r] * self.m1[s] * R
            off = self.m0[r] * self.s1[s] * R
            sends.append([self.sendbuf[f * self.Ns + off: f * self.Ns + off + blk] for f in range(nf)])
            rows = self.m0[s] * self.ncol
            roff = self.s0[s] * self.ncol
            recvs.append([self.pen_in[f * self.Np + roff: f * self.Np + roff + rows] for f in range(nf)])
        self._exchange(sends, recvs)

    def _to_slab(self, nf, acc, alpha, out):
        """self.pen_out (nf, P0, m1, R) -> out = acc + alpha * slab fields."""
        r, P1, R = self.rank, self.dims[1], self.R
        sends, recvs = [], []
        for s in range(self.G):
            rows = self.m0[s] * self.ncol
            roff = self.s0[s] * self.ncol
            sends.append([self.pen_out[f * self.Np + roff: f * self.Np + roff + rows] for f in range(nf)])
            blk = self.m0[r] * self.m1[s] * R
            off = self.m0[r] * self.s1[s] * R
            recvs.append([self.recvbuf[f * self.Ns + off: f * self.Ns + off + blk] for f in range(nf)])
        self._exchange(sends, recvs)
        dst = self.sp.device_view(out, nf * self.Ns)
        accv = self.sp.device_view(acc, nf * self.Ns) if acc else None
        for f in range(nf):
            sl = slice(f * self.Ns, (f + 1) * self.Ns)
            self.sp.slab_unpack_add(self.recvbuf[sl], accv[sl] if accv is not None else None, dst[sl], self.m0[r], P1, R, self.s1, alpha)

    def _dim0(self, kind, nf, inp, acc, alpha, out, stream):
        # the handle's launches and ours share torch's current stream (the callers pass it down)
        self._to_pencil(nf, inp)
        self._pencil(kind, nf)
        self._to_slab(nf, acc, alpha, out)
        return 0

    def destroy(self):
        self.op.destroy()


class DistEllipticOp(_SlabPencil):
    """MatMult_Elliptic and FormFunction (elliptic.C:297-339, 481-533) for ANY coefficient state on slabs of planes
    (SURVEY 8e).  Each rank owns a slab-mode handle (ell_op_create_slab); per callback the gradient and the
    divergence along dimension 0 each make one round trip to pencils (4 exchanges), everything else is local.
    For the linear state DistPoissonOp above needs only 2 exchanges and one fused launch per direction; this class
    is the general path (nonlinear residuals and Jacobians)."""

    def __init__(self, dims, sp, group=None):
        self._setup(dims, sp, group, 1)
        r = self.rank
        self.op = sp.EllipticOp(self.dims, slab=(self.s0[r], self.s0[r + 1]), dim0=self._dim0)
        self.global_size, self.dirichlet_size, self.local_size = self.op.global_size, self.op.dirichlet_size, self.op.local_size

    def _pencil(self, kind, nf):
        self.op.pencil_sweep(self.ncol, self.pen_in, self.pen_out)

    def mult(self, U, V):
        return self.op.mult(U, V)

    def function(self, U, b, rhs, gamma=0.0, exponent=2.0):
        return self.op.function(U, b, rhs, gamma, exponent)


class DistStokesOp(_SlabPencil):
    """The Stokes callbacks (StokesMatMult, StokesFunction and the MatVV / MatPV / MatVP blocks) on slabs of planes
    (SURVEY 8e).  Each rank owns a slab-mode handle (stokes_op_create_slab): gathers, node loops, sweeps along
    dimensions 1.., pressure extrapolation along them and the final scatter are local launches on the slab;
    DV[0] / DP[0] and the x-line pressure extrapolation come back here through the handle's callback.
    Per StokesMatMult: gradient (d fields there and back), stress divergence (d fields there and back), pressure
    (1 field there and back)."""

    def __init__(self, dims, sp, group=None):
        self._setup(dims, sp, group, len(dims) + 1)
        r = self.rank
        self.op = sp.StokesOp(self.dims, slab=(self.s0[r], self.s0[r + 1]), dim0=self._dim0)
        for name in ("global_size", "velocity_size", "pressure_size", "dirichlet_size", "local_nodes", "interior_nodes"):
            setattr(self, name, getattr(self.op, name))

    def _pencil(self, kind, nf):
        if kind == 0:
            self.op.pencil_sweep(nf, self.ncol, self.pen_in, self.pen_out)
        elif kind == 1:
            self.op.pencil_pressure(self.ncol, self.pen_in, self.pen_out)
        else:                          # kind 2: nf - 1 velocity fields and the pressure field in one round trip
            self.op.pencil_sweep(nf - 1, self.ncol, self.pen_in, self.pen_out)
            off = (nf - 1) * self.Np
            self.op.pencil_pressure(self.ncol, self.pen_in[off:off + self.Np], self.pen_out[off:off + self.Np])

    def mult(self, x, y):
        return self.op.mult(x, y)

    def function(self, x, y):
        return self.op.function(x, y)

    def mult_vv(self, v, out):
        return self.op.mult_vv(v, out)

    def mult_pv(self, v, pout):
        return self.op.mult_pv(v, pout)

    def mult_vp(self, p, vout):
        return self.op.mult_vp(p, vout)

    def mult_schur(self, p, pout, **kw):
        """StokesMatMultSchur (stokes.C:523-535) on slabs: VP and PV are slab calls, the built-in inner GMRES on
        MatVV completes its inner products with all-reduces."""
        if self.G > 1 and not getattr(self, "_inner_reduce", False):
            self.op.set_inner_reduce(self.group)
            self._inner_reduce = True
        return self.op.mult_schur(p, pout, **kw)


# ---------------------------------------------------------------------------------------------------------------
# Transports and the C++ slab drivers behind the ABI (csrc/comm.hip, csrc/slabx.hip)
# ---------------------------------------------------------------------------------------------------------------
class LocalGroup:
    """G ranks as host threads of ONE process (chebhip_local_group): each thread drives its own handles on its own
    stream; exchanges are event-ordered device copies between the ranks' buffers.  On one GPU this rehearses an
    N-rank run at full size; on a multi-GPU node driven from one process each thread sets its own device first."""

    def __init__(self, sp, nranks):
        import ctypes as C
        self.sp, self.G = sp, int(nranks)
        h = C.c_void_p()
        sp._chk(sp.lib().chebhip_local_group_create(self.G, C.byref(h)))
        self._h = h

    def comm(self, rank):
        return Comm(self.sp, local=(self, rank))

    def abort(self):
        """Called by a rank that failed: the ranks waiting for it get an error instead of the time limit."""
        self.sp.lib().chebhip_local_group_abort(self._h)

    def destroy(self):
        if getattr(self, "_h", None):
            self.sp.lib().chebhip_local_group_destroy(self._h)
            self._h = None


EXCHANGEV_FN = None


def _exchangev_type():
    import ctypes as C
    global EXCHANGEV_FN
    if EXCHANGEV_FN is None:
        EXCHANGEV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_long),
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_long), C.c_void_p)
    return EXCHANGEV_FN


class Comm:
    """chebhip_comm: from a torch.distributed process group (backend "nccl": an RCCL communicator made from a unique
    id that rank 0 broadcasts; backend "gloo": the rehearsal transport, staged through the host), or one rank of a
    LocalGroup.  ipc=True (process groups of one node): the direct route on top of that transport -- the slab drivers read
    the peers' arrays in place through IPC mappings (chebhip_comm_create_ipc), the process group's transport keeps the
    segment exchanges and the reductions.  Falls back to the plain transport (on every rank alike) when the node
    refuses; `transport` says what was made."""

    def __init__(self, sp, group=None, local=None, null=None, ipc=False):
        """null = (nranks, rank): no wire at all (chebhip_comm_create_null) -- times one rank's compute side alone."""
        import ctypes as C
        self.sp = sp
        L = sp.lib()
        h = C.c_void_p()
        self._nccl = None
        self._cbs = None
        self._ipc = None
        self._inner = None
        self.ipc_error = None
        if null is not None:
            self.G, self.rank = int(null[0]), int(null[1])
            sp._chk(L.chebhip_comm_create_null(self.G, self.rank, C.byref(h)))
            self.transport = "null"
        elif local is not None:
            lg, rank = local
            self.G, self.rank = lg.G, int(rank)
            sp._chk(L.chebhip_comm_create_local(lg._h, self.rank, C.byref(h)))
            self.transport = "local"
        else:
            self.G = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            if self.G > 1 and dist.get_backend(group) == "nccl":
                idbuf = C.create_string_buffer(128)
                if self.rank == 0:
                    sp._chk(L.chebhip_rccl_unique_id(idbuf))
                box = [bytes(idbuf.raw)]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                comm = C.c_void_p()
                sp._chk(L.chebhip_rccl_comm_create(self.G, self.rank, C.create_string_buffer(box[0], 128), C.byref(comm)))
                self._nccl = comm
                sp._chk(L.chebhip_comm_create_rccl(comm, self.G, self.rank, C.byref(h)))
                self.transport = "rccl"
            else:
                xfn = _exchangev_type()(self._host_exchangev(group))
                rfn = sp.allreduce_trampoline(group) if self.G > 1 else None
                self._cbs = (xfn, rfn)
                sp._chk(L.chebhip_comm_create_callback(self.G, self.rank, C.cast(xfn, C.c_void_p),
                                                       C.cast(rfn, C.c_void_p) if rfn is not None else None, None, C.byref(h)))
                self.transport = "callback"
        self._h = h
        if ipc and null is None and local is None and self.G > 1:
            self._wrap_ipc(group)

    def _wrap_ipc(self, group):
        """Collective: a shared-memory group under a name rank 0 draws, the IPC communicator over this one.  Every rank ends with the
        same kind: if any of them could not set it up, all keep the message transport."""
        import ctypes as C
        import uuid
        sp, L = self.sp, self.sp.lib()
        src = dist.get_global_rank(group, 0) if group is not None else 0
        box = ["/chebhip-%s" % uuid.uuid4().hex[:24]]
        dist.broadcast_object_list(box, src=src, group=group)
        g, h2 = C.c_void_p(), C.c_void_p()
        rc = L.chebhip_ipc_group_open(box[0].encode(), self.G, self.rank, C.byref(g))
        if rc == 0:
            rc = L.chebhip_comm_create_ipc(g, self._h, C.byref(h2))
        if rc != 0:
            self.ipc_error = L.chebhip_last_error().decode()
        on_gpu = dist.get_backend(group) == "nccl"
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 1:
            self._ipc, self._inner, self._h = g, self._h, h2
            self.transport = "ipc+" + self.transport
            return
        if h2:
            L.chebhip_comm_destroy(h2)
        if g:
            L.chebhip_ipc_group_close(g)
        if self.ipc_error is None:
            self.ipc_error = "another rank could not set the IPC group up"

    def _host_exchangev(self, group):
        """chebhip_exchangev_fn under gloo: device -> host, batched isend / irecv, host -> device, ordered on `stream`."""
        sp = self.sp

        def xfn(ctx, nseg, peers, sends, scounts, recvs, rcounts, stream):
            try:
                ext = _torch_stream(stream)
                with torch.cuda.stream(ext):
                    ops, back = [], []
                    for i in range(nseg):           # receives first, in segment order; matching is by order per peer
                        if rcounts[i] > 0:
                            buf = torch.empty(int(rcounts[i]), dtype=torch.float64)
                            back.append((int(recvs[i]), buf))
                            ops.append(dist.P2POp(dist.irecv, buf, int(peers[i]), group))
                    for i in range(nseg):
                        if scounts[i] > 0:
                            ops.append(dist.P2POp(dist.isend, sp.device_view(sends[i], int(scounts[i])).cpu(), int(peers[i]), group))
                    if ops:
                        for w in dist.batch_isend_irecv(ops):
                            w.wait()
                    for ptr, buf in back:
                        sp.device_view(ptr, buf.numel()).copy_(buf)
                    ext.synchronize()
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 5
        return xfn

    def set_null_shadow(self, k, tensors):
        """NULL transport: tensors[r] (None = the rank's own) stands for peer r's k-th posted array (chebhip_comm_null_set_shadow) --
        the direct route's kernels then touch G distinct arrays.  The tensors are kept alive by this object."""
        import ctypes as C
        arr = (C.c_void_p * self.G)(*[(t.data_ptr() if t is not None else None) for t in tensors])
        self.sp._chk(self.sp.lib().chebhip_comm_null_set_shadow(self._h, int(k), arr))
        self._shadow = getattr(self, "_shadow", {}); self._shadow[int(k)] = list(tensors)

    def allreduce_sum(self, value):
        """Sum of one float over the ranks (chebhip_comm_reduce on a device scalar, on torch's current stream)."""
        t = torch.tensor([float(value)], dtype=torch.float64, device="cuda")
        self.sp._chk(self.sp.lib().chebhip_comm_reduce(self._h, t.data_ptr(), 1, self.sp._stream()))
        return float(t.item())

    def norm(self, t):
        """Global 2-norm of a vector distributed over the ranks."""
        return self.allreduce_sum(float((t * t).sum())) ** 0.5

    def reduce_fn(self):
        """(chebhip_reduce_fn, ctx) for Fgmres.set_reduce-style hooks: chebhip_comm_reduce on this communicator."""
        import ctypes as C
        return C.cast(self.sp.lib().chebhip_comm_reduce, C.c_void_p), self._h

    def destroy(self):
        if getattr(self, "_h", None):
            self.sp.lib().chebhip_comm_destroy(self._h)
            self._h = None
        if getattr(self, "_inner", None):
            self.sp.lib().chebhip_comm_destroy(self._inner)
            self._inner = None
        if getattr(self, "_ipc", None):
            self.sp.lib().chebhip_ipc_group_close(self._ipc)
            self._ipc = None
        if getattr(self, "_nccl", None):
            self.sp.lib().chebhip_rccl_comm_destroy(self._nccl)
            self._nccl = None


class _DistC:
    def _ranges(self, fn):
        import ctypes as C
        r = (C.c_long * 4)()
        self.sp._chk(fn(self._h, r))
        return (int(r[0]), int(r[1])), (int(r[2]), int(r[3]))

    def serial_ranges(self):
        """((interior node lo, hi), (boundary node lo, hi)) of this rank's pieces in the serial vectors."""
        return self._rng

    def destroy(self):
        for w in getattr(self, "_saddles", []):       # borrowers of this driver's handles go first
            M = w()
            if M is not None:
                M.destroy()
        self._saddles = []
        if getattr(self, "_h", None):
            torch.cuda.synchronize()
            self._destroy(self._h)
            self._h = None
        if getattr(self, "_own_comm", None):
            self._own_comm.destroy()
            self._own_comm = None


class DistStokesC(_DistC):
    """The Stokes callbacks on slabs with the host in C++ (csrc/slabx.hip: chebhip_dist_stokes_*): partition, pack, the
    grouped exchanges, pencil launches and the AXPY-fused unpack are behind the ABI; `op` is the slab-mode StokesOp."""

    def __init__(self, dims, sp, comm=None, group=None, ipc=False):
        import ctypes as C
        self.sp = sp
        self.dims = tuple(int(v) for v in dims)
        self._own_comm = None
        if comm is None and dist.is_initialized() and dist.get_world_size(group) > 1:
            comm = self._own_comm = Comm(sp, group=group, ipc=ipc)    # ipc: the direct route among the processes of one node
        self.comm = comm
        L = sp.lib()
        h = C.c_void_p()
        sp._chk(L.chebhip_dist_stokes_create(len(self.dims), (C.c_int * len(self.dims))(*self.dims), comm._h if comm else None, C.byref(h)))
        self._h, self._destroy = h, L.chebhip_dist_stokes_destroy
        self.op = sp.StokesOp(self.dims, handle=L.chebhip_dist_stokes_op(h))
        self._rng = self._ranges(L.chebhip_dist_stokes_ranges)
        for name in ("global_size", "velocity_size", "pressure_size", "dirichlet_size", "local_nodes", "interior_nodes"):
            setattr(self, name, getattr(self.op, name))
        for name in ("mult", "function", "mult_vv", "mult_pv", "mult_vp", "mult_schur"):
            setattr(self, name, getattr(self.op, name))

    def pc(self):
        """MatVVPC (stokes.C:1160-1241) for the slab's velocity unknowns: FdPc in slab mode, owned by this driver."""
        import ctypes as C
        if getattr(self, "_pc", None) is None:
            h = C.c_void_p()
            self.sp._chk(self.sp.lib().chebhip_dist_stokes_pc(self._h, C.byref(h)))
            self._pc = self.sp.FdPc(self.op, sweeps=0, handle=h)
        return self._pc

    def saddle(self, saddle_type=0, vel=(4, 1e-5), schur=(3, 1e-5), svel=(0, 1e-5), schur_jacobi=True):
        """StokesPCApply0..3 on slabs (collective): the block preconditioner of the whole saddle-point system."""
        if self.comm is None:
            raise ValueError("slab preconditioner needs a communicator")
        rfn, rctx = self.comm.reduce_fn()
        M = self.sp.StokesSaddlePc(self.op, saddle_type, vel, schur, svel, 0, schur_jacobi, slab=(self.pc(), rfn, rctx))
        # the saddle borrows the driver's pc handle and the communicator's reduce context: it keeps both alive, and
        # destroy() of this driver destroys the saddles it handed out first (chebhip.h: stokes_saddle_create_slab)
        M._slab_owner = (self, self.comm)
        self._saddles = [w for w in getattr(self, "_saddles", []) if w() is not None] + [weakref.ref(M)]
        return M


class DistEllipticC(_DistC):
    """MatMult_Elliptic / FormFunction for any coefficient state on slabs, host in C++ (chebhip_dist_ell_*)."""

    def __init__(self, dims, sp, comm=None, group=None, ipc=False):
        import ctypes as C
        self.sp = sp
        self.dims = tuple(int(v) for v in dims)
        self._own_comm = None
        if comm is None and dist.is_initialized() and dist.get_world_size(group) > 1:
            comm = self._own_comm = Comm(sp, group=group, ipc=ipc)    # ipc: the direct route among the processes of one node
        self.comm = comm
        L = sp.lib()
        h = C.c_void_p()
        sp._chk(L.chebhip_dist_ell_create(len(self.dims), (C.c_int * len(self.dims))(*self.dims), comm._h if comm else None, C.byref(h)))
        self._h, self._destroy = h, L.chebhip_dist_ell_destroy
        self.op = sp.EllipticOp(self.dims, handle=L.chebhip_dist_ell_op(h))
        self._rng = self._ranges(L.chebhip_dist_ell_ranges)
        self.global_size, self.dirichlet_size, self.local_size = self.op.global_size, self.op.dirichlet_size, self.op.local_size
        self.mult, self.function = self.op.mult, self.op.function

    def pc(self):
        """FormJacobian's preconditioner (elliptic.C:537-590) for the slab's unknowns: FdPc in slab mode, owned by this driver."""
        import ctypes as C
        if getattr(self, "_pc", None) is None:
            h = C.c_void_p()
            self.sp._chk(self.sp.lib().chebhip_dist_ell_pc(self._h, C.byref(h)))
            self._pc = self.sp.FdPc(self.op, sweeps=0, handle=h)
        return self._pc
