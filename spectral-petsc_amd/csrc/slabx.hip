// slabx.hip -- the Stokes callbacks and the general-coefficient elliptic callbacks on slabs of planes, host side in C++
// behind the C ABI (BASELINE config 5; the reference itself is serial: stokes.C:121, elliptic.C:262).
//
// Each rank owns a slab-mode operator handle (stokes_op_create_slab / ell_op_create_slab) on its planes [s0[r], s0[r+1])
// of grid dimension 0 -- the full local grid, boundary planes included: Dirichlet rows and the pressure end points
// take part in the sweeps.  Gathers, node loops, sweeps along dimensions 1.., the pressure extrapolation along them
// and the final scatter are the serial launches on the slab.  Whatever runs along dimension 0 (DV[0], DP[0], D_0 and the
// x-line extrapolation of StokesPressureReduceOrder, stokes.C:1064-1074) comes back here through the handle's callback:
//
//   slab fields --pack (one launch), exchange--> pencil fields --pencil launch--> pencil result --exchange, unpack+AXPY--> slab
//
// One exchange moves all fields of a call in ONE grouped launch (comm.hip).  The pencil side of the forward exchange and
// the pencil-row side of the backward one need no (un)packing: a peer's block is a run of whole pencil planes.
#include "comm.h"
#include "sweep.h"
#include <cstdlib>
#include <new>
#include <vector>

int chebhip_fail(int code, const char *fmt, ...);   // chebhip.hip
using chebhip::XSeg;

#define XHIPCHK(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {

struct Split { int G; long c1[65]; };
// One workgroup per (plane i0, peer s, field f): slab[f][i0, c1[s]:c1[s+1], :] is a contiguous run of (c1[s+1] - c1[s]) R
// doubles, and so is its place in the field's exchange buffer (peer-major: for peer s the block slab[:, c1[s]:c1[s+1], :]).
// The rank's own block (peer `own`) goes straight to / comes straight from the pencil arrays: it needs no message.
// V2: every run starts on a 16-byte boundary and has even length.
template <bool V2>
__global__ __launch_bounds__(256) void k_xpack(Split sp, long m0, long M1, long R, long Ns, const double *__restrict__ slab, double *__restrict__ buf,
                                              int own, double *__restrict__ own_ptr, long own_fstride) {
  const int s = (int)(blockIdx.x % (unsigned)sp.G); const long i0 = blockIdx.x / (unsigned)sp.G, f = blockIdx.y;
  const long w = sp.c1[s + 1] - sp.c1[s], len = w * R;
  const double *src = slab + f * Ns + (i0 * M1 + sp.c1[s]) * R;
  double *dst = (s == own) ? own_ptr + f * own_fstride + i0 * len : buf + f * Ns + m0 * sp.c1[s] * R + i0 * len;
  if (V2) { for (long t = threadIdx.x; t < (len >> 1); t += blockDim.x) ((double2 *)dst)[t] = ((const double2 *)src)[t]; }
  else { for (long t = threadIdx.x; t < len; t += blockDim.x) dst[t] = src[t]; }
}
// out = (acc ? acc : 0) + alpha * slab-ordered(buf), nf fields, one launch
template <bool V2>
__global__ __launch_bounds__(256) void k_xunpack(Split sp, long m0, long M1, long R, long Ns, const double *__restrict__ buf, int own,
                                                const double *__restrict__ own_ptr, long own_fstride, const double *acc, double alpha, double *out) {
  const int s = (int)(blockIdx.x % (unsigned)sp.G); const long i0 = blockIdx.x / (unsigned)sp.G, f = blockIdx.y;
  const long w = sp.c1[s + 1] - sp.c1[s], len = w * R, e0 = f * Ns + (i0 * M1 + sp.c1[s]) * R;
  const double *src = (s == own) ? own_ptr + f * own_fstride + i0 * len : buf + f * Ns + m0 * sp.c1[s] * R + i0 * len;
  if (V2) {
    for (long t = threadIdx.x; t < (len >> 1); t += blockDim.x) {
      const double2 b = ((const double2 *)src)[t];
      double2 v = make_double2(alpha * b.x, alpha * b.y);
      if (acc) { const double2 a = ((const double2 *)(acc + e0))[t]; v.x = a.x + v.x; v.y = a.y + v.y; }
      ((double2 *)(out + e0))[t] = v;
    }
  } else {
    for (long t = threadIdx.x; t < len; t += blockDim.x) { const double v = alpha * src[t]; out[e0 + t] = acc ? acc[e0 + t] + v : v; }
  }
}
// Direct transports (comm.h): out = (acc ? acc : 0) + alpha * T with T read IN PLACE from the peers' pencil results: the run of the
// (c1[s+1] - c1[s]) R doubles of row s0r + i0 of field f of peer s (field stride src.fs[s], row pitch src.pitch[s]).
struct XPull { const double *p[64]; long fs[64]; long pitch[64]; };
template <bool V2>
__global__ __launch_bounds__(256) void k_xpull_unpack(Split sp, XPull src, long s0r, long M1, long R, long Ns, const double *acc, double alpha, double *out) {
  const int s = (int)(blockIdx.x % (unsigned)sp.G); const long i0 = blockIdx.x / (unsigned)sp.G, f = blockIdx.y;
  const long w = sp.c1[s + 1] - sp.c1[s], len = w * R, e0 = f * Ns + (i0 * M1 + sp.c1[s]) * R;
  const double *from = src.p[s] + f * src.fs[s] + (s0r + i0) * src.pitch[s];
  if (V2) {
    for (long t = threadIdx.x; t < (len >> 1); t += blockDim.x) {
      const double2 b = ((const double2 *)from)[t];
      double2 v = make_double2(alpha * b.x, alpha * b.y);
      if (acc) { const double2 a = ((const double2 *)(acc + e0))[t]; v.x = a.x + v.x; v.y = a.y + v.y; }
      ((double2 *)(out + e0))[t] = v;
    }
  } else {
    for (long t = threadIdx.x; t < len; t += blockDim.x) { const double v = alpha * from[t]; out[e0 + t] = acc ? acc[e0 + t] + v : v; }
  }
}
void split_sizes(long n, int parts, std::vector<long> &sz) { sz.resize(parts); for (int i = 0; i < parts; i++) sz[i] = n / parts + (i < n % parts ? 1 : 0); }

// The slab <-> pencil machinery for fields on the full local grid (P0, P1, R): slabs of planes of dimension 0, pencils
// holding all P0 planes and a share of dimension 1.
struct SlabX {
  chebhip_comm *comm = nullptr;
  int d = 0, G = 1, rank = 0, nf_max = 1;
  std::vector<int> dims;
  long P0 = 0, P1 = 0, R = 1;
  std::vector<long> m0, m1, s0, s1;
  long Ns = 0, ncol = 0, Np = 0;               // nodes of the slab, lines of the pencil, nodes of the pencil
  double *sendbuf = nullptr, *recvbuf = nullptr, *pen_in = nullptr, *pen_out = nullptr;
  Split split;
  std::vector<XSeg> segs;
  bool used_direct = false;                    // the peers' kernels have read this rank's arrays in place (collective destroy)

  ~SlabX() { double *all[] = {sendbuf, recvbuf, pen_in, pen_out}; for (double *p : all) if (p) (void)hipFree(p); }

  int setup(int d_, const int *dims_, chebhip_comm *c, int nf) {
    if (!dims_ || d_ < 2 || d_ > 10) return chebhip_fail(CHEBHIP_ERR_DIMS, "slab partitioning needs 2 <= d <= 10");
    comm = c; d = d_; dims.assign(dims_, dims_ + d_); nf_max = nf;
    G = chebhip::comm_size(c); rank = chebhip::comm_rank(c);
    if (G < 1 || G > 64) return chebhip_fail(CHEBHIP_ERR_ARG, "1..64 ranks");
    P0 = dims[0]; P1 = dims[1]; R = 1; for (int k = 2; k < d; k++) R *= dims[k];
    if (P0 < G || P1 < G) return chebhip_fail(CHEBHIP_ERR_SIZE, "slab partition over %d ranks: every rank needs a plane along dimensions 0 and 1", G);
    split_sizes(P0, G, m0); split_sizes(P1, G, m1);
    s0.assign(G + 1, 0); s1.assign(G + 1, 0);
    for (int s = 0; s < G; s++) { s0[s + 1] = s0[s] + m0[s]; s1[s + 1] = s1[s] + m1[s]; }
    Ns = m0[rank] * P1 * R; ncol = m1[rank] * R; Np = P0 * ncol;
    split.G = G; for (int s = 0; s <= G; s++) split.c1[s] = s1[s];
    // (one row of slack behind the pencils: with the NULL transport a rank reads its own pencil result with the peers' column counts)
    long wmax = 0; for (int s = 0; s < G; s++) wmax = m1[s] > wmax ? m1[s] : wmax;
    const size_t sb = (size_t)nf * (size_t)(Ns > 0 ? Ns : 1) * sizeof(double), pb = ((size_t)nf * (size_t)(Np > 0 ? Np : 1) + (size_t)(wmax + 1) * R) * sizeof(double);
    XHIPCHK(hipMalloc((void **)&sendbuf, sb)); XHIPCHK(hipMalloc((void **)&recvbuf, sb));
    XHIPCHK(hipMalloc((void **)&pen_in, pb)); XHIPCHK(hipMalloc((void **)&pen_out, pb));
    return 0;
  }

  // Direct transports: can the dimension-0 sweeps read the ranks' slab fields in place?  Decided from what EVERY rank knows (the
  // geometry of all ranks), so that all ranks take the same route: 16-byte rows and column blocks of at least 16 doubles on every rank.
  bool direct_geometry() const {
    const int pk = chebhip::opt(chebhip::OPT_DIST_PACKED_EXCHANGE);        // (4: the direct route in its pull form, direct_open)
    if (!comm || !chebhip::comm_direct(comm) || (pk != 0 && pk != 4) || G > chebhip::GATHER_MAX) return false;
    if (((P1 * R) & 1) || P1 * R >= 0x7fffffffL) return false;
    for (int s = 0; s < G; s++) if (m0[s] < 1 || m1[s] * R < 16 || ((m1[s] * R) & 1) || ((s1[s] * R) & 1) || ((m0[s] * P1 * R) & 1) || m0[s] * P1 * R >= 0x7fffffffL) return false;
    return (Np & 1) == 0;
  }
  // forward half: rendezvous on `in` (nf slab fields), the GatherSrc of this rank's pencil over the peers' fields.  PUSH (default; pull
  // form with dist_packed_exchange = 4, read at every round trip -- the same on every rank): every rank also posts its receive buffer
  // (nf slab-shaped fields) and the pencil sweeps store each output row into the buffer of the rank that owns the plane, field by
  // field where the operand came from -- loads and stores of a round trip in one launch (csrc/dist.hip) -- so the unpack is local.
  bool push = true;
  int direct_open(const double *in, hipStream_t st, chebhip::GatherSrc *g) {
    used_direct = true;
    push = chebhip::opt(chebhip::OPT_DIST_PACKED_EXCHANGE) != 4;
    const bool null = chebhip::comm_is_null(comm);
    chebhip::PeerView pv;
    const double *post[2] = {in, push ? recvbuf : nullptr};
    int rc = chebhip::comm_rendezvous(comm, post, 2, 0, 3, st, &pv);          // the peers' fields are complete; they have finished reading my previous pencil result / their receive buffers
    if (rc) return rc;
    *g = chebhip::GatherSrc{};
    g->G = G; g->rowlen = (unsigned)(P1 * R); g->col0 = (unsigned)(s1[rank] * R);
    for (int s = 0; s < G; s++) {
      if (!pv.ptr[s][0] || ((size_t)pv.ptr[s][0] & 15)) return chebhip_fail(CHEBHIP_ERR_ARG, "slab exchange: rank %d posted an unaligned field array", s);
      g->p[s] = pv.ptr[s][0]; g->s0[s] = (int)s0[s]; g->lq[s] = (unsigned)(null ? Ns : m0[s] * P1 * R); g->pmax[s] = (int)(null ? m0[rank] : m0[s]);
      if (push && (!pv.ptr[s][1] || ((size_t)pv.ptr[s][1] & 15))) return chebhip_fail(CHEBHIP_ERR_ARG, "slab exchange: rank %d posted no receive buffer (the ranks disagree on dist_packed_exchange)", s);
      g->dp[s] = push ? (double *)pv.ptr[s][1] : nullptr;
    }
    g->s0[G] = (int)s0[G];
    g->push = push ? 1 : 0;
    return 0;
  }
  // backward half: my reads of the peers' fields end here; rendezvous on pen_out; out = (acc ? acc : 0) + alpha * (the peers' pencil results)
  int direct_close(int nf, const double *acc, double alpha, double *out, hipStream_t st) {
    const bool null = chebhip::comm_is_null(comm);
    int rc = chebhip::comm_mark(comm, 2, st); if (rc) return rc;
    chebhip::PeerView pv;
    const double *post[1] = {push ? nullptr : pen_out};
    rc = chebhip::comm_rendezvous(comm, post, 1, 1, 2, st, &pv);              // the peers' pencil results are complete (push: have landed in my receive buffer); they have finished reading my fields
    if (rc) return rc;
    if (Ns > 0) {
      XPull xp;
      for (int s = 0; s < G; s++) {
        const long m1s = null ? m1[rank] : m1[s]; xp.p[s] = pv.ptr[s][0]; xp.pitch[s] = m1s * R; xp.fs[s] = P0 * m1s * R;
        if (push) { xp.p[s] = recvbuf + s1[s] * R; xp.pitch[s] = P1 * R; xp.fs[s] = Ns; }     // the local slab-shaped fields, segment s of a plane
      }
      const dim3 grid((unsigned)(m0[rank] * G), (unsigned)nf);
      bool v2 = vec2(out, acc, nullptr);
      for (int s = 0; s < G && v2; s++) v2 = ((size_t)xp.p[s] & 15) == 0 && (xp.pitch[s] & 1) == 0 && (xp.fs[s] & 1) == 0;
      const long s0r = push ? 0 : s0[rank];
      if (v2) hipLaunchKernelGGL((k_xpull_unpack<true>), grid, dim3(256), 0, st, split, xp, s0r, P1, R, Ns, acc, alpha, out);
      else hipLaunchKernelGGL((k_xpull_unpack<false>), grid, dim3(256), 0, st, split, xp, s0r, P1, R, Ns, acc, alpha, out);
      XHIPCHK(hipGetLastError());
    }
    return chebhip::comm_mark(comm, 3, st);                                   // my reads of the peers' pencil results end here
  }
  // before the arrays go: every rank has drained its device (collective among the rank threads; an aborted group returns at once)
  void direct_quiesce() { if (used_direct && comm) { (void)hipDeviceSynchronize(); (void)chebhip::comm_group_barrier(comm); used_direct = false; } }

  // The same machinery for fields in the INTERIOR layout (M0, M1, R') = dims - 2 of the global vectors, partitioned like the
  // operator's slabs: rank r holds the interior planes that fall into its planes [full_s0[r], full_s0[r+1]) of the full grid
  // (possibly none).  Used by the slab-mode preconditioner (precond.hip), whose vectors are the operator's unknowns.
  int setup_interior(int d_, const int *dims_, chebhip_comm *c, int nf, const std::vector<long> &full_s0) {
    if (!dims_ || d_ < 2 || d_ > 10) return chebhip_fail(CHEBHIP_ERR_DIMS, "slab partitioning needs 2 <= d <= 10");
    comm = c; d = d_; dims.assign(dims_, dims_ + d_); nf_max = nf;
    G = chebhip::comm_size(c); rank = chebhip::comm_rank(c);
    if (G < 1 || G > 64 || (int)full_s0.size() != G + 1) return chebhip_fail(CHEBHIP_ERR_ARG, "1..64 ranks");
    P0 = dims[0] - 2; P1 = dims[1] - 2; R = 1; for (int k = 2; k < d; k++) R *= dims[k] - 2;
    if (P0 < 1 || P1 < 1 || R < 1) return chebhip_fail(CHEBHIP_ERR_SIZE, "no interior nodes");
    m0.assign(G, 0);
    for (int s = 0; s < G; s++) {
      const long lo = full_s0[s] > 1 ? full_s0[s] : 1, hi = full_s0[s + 1] < dims[0] - 1 ? full_s0[s + 1] : dims[0] - 1;
      m0[s] = hi > lo ? hi - lo : 0;
    }
    split_sizes(P1, G, m1);
    s0.assign(G + 1, 0); s1.assign(G + 1, 0);
    for (int s = 0; s < G; s++) { s0[s + 1] = s0[s] + m0[s]; s1[s + 1] = s1[s] + m1[s]; }
    Ns = m0[rank] * P1 * R; ncol = m1[rank] * R; Np = P0 * ncol;
    split.G = G; for (int s = 0; s <= G; s++) split.c1[s] = s1[s];
    const size_t sb = (size_t)nf * (size_t)(Ns > 0 ? Ns : 1) * sizeof(double), pb = (size_t)nf * (size_t)(Np > 0 ? Np : 1) * sizeof(double);
    XHIPCHK(hipMalloc((void **)&sendbuf, sb)); XHIPCHK(hipMalloc((void **)&recvbuf, sb));
    XHIPCHK(hipMalloc((void **)&pen_in, pb)); XHIPCHK(hipMalloc((void **)&pen_out, pb));
    return 0;
  }

  // the peer whose block bypasses the exchange buffers: this rank (none with option rccl_self_messages: one-rank smoke runs of the transport)
  int own() const { return chebhip::opt(chebhip::OPT_RCCL_SELF_MESSAGES) ? -1 : rank; }
  // 16-byte accesses of the pack / unpack launches: every run must start even-aligned and have even length
  bool vec2(const double *a, const double *b, const double *c) const {
    if ((((size_t)a | (size_t)b | (size_t)c) & 15) != 0 || (Ns & 1) || (Np & 1)) return false;
    if ((R & 1) == 0) return true;
    if (P1 & 1) return false;
    for (int s = 0; s <= G; s++) if (s1[s] & 1) return false;
    return true;
  }
  // nf slab fields at `in` -> pen_in as (nf, P0, m1, R)
  int to_pencil(int nf, const double *in, hipStream_t st) {
    if (nf < 1 || nf > nf_max) return chebhip_fail(CHEBHIP_ERR_ARG, "slab exchange: %d fields, at most %d", nf, nf_max);
    double *own_in = pen_in + s0[rank] * ncol;                 // the own block inside the pencil: planes s0[rank] .. s0[rank+1]
    if (Ns > 0) {
      const dim3 grid((unsigned)(m0[rank] * G), (unsigned)nf);
      if (vec2(in, nullptr, nullptr)) hipLaunchKernelGGL((k_xpack<true>), grid, dim3(256), 0, st, split, m0[rank], P1, R, Ns, in, sendbuf, own(), own_in, Np);
      else hipLaunchKernelGGL((k_xpack<false>), grid, dim3(256), 0, st, split, m0[rank], P1, R, Ns, in, sendbuf, own(), own_in, Np);
      XHIPCHK(hipGetLastError());
    }
    segs.clear();
    for (int s = 0; s < G; s++)
      for (int f = 0; f < nf && s != own(); f++)
        segs.push_back(XSeg{s, sendbuf + f * Ns + m0[rank] * s1[s] * R, m0[rank] * m1[s] * R, pen_in + f * Np + s0[s] * ncol, m0[s] * ncol});
    if (segs.empty()) return 0;
    return chebhip::comm_exchange(comm, segs.data(), (int)segs.size(), st);
  }
  // pen_out (nf, P0, m1, R) -> out = (acc ? acc : 0) + alpha * slab fields
  int to_slab(int nf, const double *acc, double alpha, double *out, hipStream_t st) {
    if (nf < 1 || nf > nf_max) return chebhip_fail(CHEBHIP_ERR_ARG, "slab exchange: %d fields, at most %d", nf, nf_max);
    segs.clear();
    for (int s = 0; s < G; s++)
      for (int f = 0; f < nf && s != own(); f++)
        segs.push_back(XSeg{s, pen_out + f * Np + s0[s] * ncol, m0[s] * ncol, recvbuf + f * Ns + m0[rank] * s1[s] * R, m0[rank] * m1[s] * R});
    if (!segs.empty()) { int rc = chebhip::comm_exchange(comm, segs.data(), (int)segs.size(), st); if (rc) return rc; }
    if (Ns > 0) {
      const dim3 grid((unsigned)(m0[rank] * G), (unsigned)nf);
      const double *own_out = pen_out + s0[rank] * ncol;
      if (vec2(out, acc, nullptr)) hipLaunchKernelGGL((k_xunpack<true>), grid, dim3(256), 0, st, split, m0[rank], P1, R, Ns, (const double *)recvbuf, own(), own_out, Np, acc, alpha, out);
      else hipLaunchKernelGGL((k_xunpack<false>), grid, dim3(256), 0, st, split, m0[rank], P1, R, Ns, (const double *)recvbuf, own(), own_out, Np, acc, alpha, out);
      XHIPCHK(hipGetLastError());
    }
    return 0;
  }

  // where this rank's pieces sit in the serial vectors (dimension 0 outermost => contiguous): node ranges
  void ranges(long *r4) const {
    long inner_int = 1, inner_all = 1;
    for (int k = 1; k < d; k++) { inner_int *= dims[k] - 2; inner_all *= dims[k]; }
    const long lo = s0[rank], hi = s0[rank + 1];
    const long lo1 = lo > 1 ? lo : 1, hi1 = hi < P0 - 1 ? hi : P0 - 1;
    const long ilo = lo1 - 1, ihi = (hi1 - 1 > ilo) ? hi1 - 1 : ilo;                 // interior planes before / up to this slab
    auto bnodes = [&](long plane_hi) {                                                // boundary nodes in planes [0, plane_hi)
      const long full = (plane_hi < 1 ? plane_hi : 1) + (plane_hi - (P0 - 1) > 0 ? plane_hi - (P0 - 1) : 0);
      return full * inner_all + (plane_hi - full) * (inner_all - inner_int);
    };
    r4[0] = ilo * inner_int; r4[1] = ihi * inner_int; r4[2] = bnodes(lo); r4[3] = bnodes(hi);
  }
};

}  // namespace

// ---- Stokes ------------------------------------------------------------------------------------------------------
struct chebhip_dist_stokes { SlabX x; stokes_op *op = nullptr; SlabX *xi = nullptr; chebhip_fdpc *pc = nullptr; bool direct = false; };

// dimension 0 of the slab-mode preconditioner (chebhip_fdpc_dim0_fn): interior fields slab -> pencil, line transform, back
static int pc_dim0(SlabX *xi, chebhip_fdpc *pc, int backward, int nf, const double *in, double *out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = xi->to_pencil(nf, in, st);
  if (!rc) rc = chebhip_fdpc_pencil_transform(pc, backward, nf, xi->ncol, xi->pen_in, xi->pen_out, stream);
  if (!rc) rc = xi->to_slab(nf, nullptr, 1.0, out, st);
  if (rc) chebhip::comm_abort(xi->comm);
  return rc;
}
static int dstokes_pc_dim0(void *ctx, int backward, int nf, const double *in, double *out, void *stream) {
  chebhip_dist_stokes *D = (chebhip_dist_stokes *)ctx;
  return pc_dim0(D->xi, D->pc, backward, nf, in, out, stream);
}

static int dstokes_dim0(void *ctx, int kind, int nf, const double *in, const double *acc, double alpha, double *out, void *stream) {
  chebhip_dist_stokes *D = (chebhip_dist_stokes *)ctx;
  hipStream_t st = (hipStream_t)stream;
  // Direct transports (LOCAL thread ranks, NULL): the pencil sweep(s) read the peers' slab fields in place and the unpack reads the
  // peers' pencil results in place -- no pack, no messages; two rendezvous per round trip (csrc/dist.hip has the protocol).
  if (D->direct) {
    chebhip::GatherSrc g;
    bool done = false;
    int rc = D->x.direct_open(in, st, &g);
    if (!rc) rc = chebhip::stokes_pencil_gather_try(D->op, kind, nf, D->x.ncol, g, D->x.pen_out, st, &done);
    if (!rc && !done) rc = chebhip_fail(CHEBHIP_ERR_ARG, "slab exchange: the gather launch refused a geometry the route had accepted");
    if (!rc) rc = D->x.direct_close(nf, acc, alpha, out, st);
    if (rc) chebhip::comm_abort(D->x.comm);
    return rc;
  }
  // a failure between the two exchanges must not leave the peers of a thread-rank group waiting for this rank
  int rc = D->x.to_pencil(nf, in, st); if (rc) { chebhip::comm_abort(D->x.comm); return rc; }
  if (kind == 0) rc = stokes_op_pencil_sweep(D->op, nf, D->x.ncol, D->x.pen_in, D->x.pen_out, stream);     // DV[0] / DP[0]
  else if (kind == 1) rc = stokes_op_pencil_pressure(D->op, D->x.ncol, D->x.pen_in, D->x.pen_out, stream);  // x-line extrapolation + DP[0]
  else {                                                     // kind 2: nf - 1 velocity fields and the pressure field in one round trip
    rc = stokes_op_pencil_sweep_pressure(D->op, nf - 1, D->x.ncol, D->x.pen_in, D->x.pen_out, stream);      // (two jobs of one launch)
  }
  if (!rc) rc = D->x.to_slab(nf, acc, alpha, out, st);
  if (rc) chebhip::comm_abort(D->x.comm);
  return rc;
}

extern "C" int chebhip_dist_stokes_destroy(chebhip_dist_stokes *D) {
  if (!D) return 0;
  D->x.direct_quiesce();
  if (D->pc) chebhip_fdpc_destroy(D->pc);
  delete D->xi;
  if (D->op) stokes_op_destroy(D->op);
  delete D;
  return 0;
}
extern "C" int chebhip_dist_stokes_create(int d, const int *dims, chebhip_comm *comm, chebhip_dist_stokes **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  chebhip_dist_stokes *D = new (std::nothrow) chebhip_dist_stokes;
  if (!D) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  int rc = D->x.setup(d, dims, comm, d + 1);
  if (!rc) rc = stokes_op_create_slab(d, dims, (int)D->x.s0[D->x.rank], (int)D->x.s0[D->x.rank + 1], dstokes_dim0, D, &D->op);
  if (!rc && D->x.G > 1) rc = stokes_op_set_inner_reduce(D->op, chebhip_comm_reduce, comm);
  if (rc) { chebhip_dist_stokes_destroy(D); return rc; }
  // (every rank decides alike: the geometry of all ranks, the options, and matrices that depend on the global extent only)
  D->direct = D->x.direct_geometry() && chebhip::stokes_pencil_gather_supported(D->op);
  *out = D;
  return 0;
}
extern "C" stokes_op *chebhip_dist_stokes_op(chebhip_dist_stokes *D) { return D ? D->op : nullptr; }
// MatVVPC (stokes.C:1160-1241) for the slab's velocity unknowns: the handle of precond.hip in slab mode, its transforms along
// dimension 0 on pencils through this driver's communicator.  Owned by the driver (destroyed with it); one per driver.
extern "C" int chebhip_dist_stokes_pc(chebhip_dist_stokes *D, chebhip_fdpc **out) {
  if (!D || !out) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  *out = nullptr;
  if (!D->pc) {
    const int d = D->x.d;
    SlabX *xi = new (std::nothrow) SlabX;
    if (!xi) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
    int rc = xi->setup_interior(d, D->x.dims.data(), D->x.comm, d, D->x.s0);
    if (rc) { delete xi; return rc; }
    D->xi = xi;
    rc = stokes_pc_create_slab(D->op, xi->s0[xi->rank], dstokes_pc_dim0, D, &D->pc);
    if (rc) { delete D->xi; D->xi = nullptr; D->pc = nullptr; return rc; }
  }
  *out = D->pc;
  return 0;
}
extern "C" int chebhip_dist_stokes_ranges(const chebhip_dist_stokes *D, long *ranges4) {
  if (!D || !ranges4) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  D->x.ranges(ranges4);
  return 0;
}

// ---- general-coefficient elliptic operator -----------------------------------------------------------------------
struct chebhip_dist_ell { SlabX x; ell_op *op = nullptr; SlabX *xi = nullptr; chebhip_fdpc *pc = nullptr; bool direct = false; };
static int dell_pc_dim0(void *ctx, int backward, int nf, const double *in, double *out, void *stream) {
  chebhip_dist_ell *D = (chebhip_dist_ell *)ctx;
  return pc_dim0(D->xi, D->pc, backward, nf, in, out, stream);
}

static int dell_dim0(void *ctx, int kind, int nf, const double *in, const double *acc, double alpha, double *out, void *stream) {
  chebhip_dist_ell *D = (chebhip_dist_ell *)ctx;
  (void)kind;
  hipStream_t st = (hipStream_t)stream;
  if (D->direct && nf == 1) {                  // direct transports: the pencil sweep reads the ranks' slab field in place (see dstokes_dim0)
    chebhip::GatherSrc g;
    bool done = false;
    int rc = D->x.direct_open(in, st, &g);
    if (!rc) rc = chebhip::ell_pencil_gather_try(D->op, D->x.ncol, g, D->x.pen_out, st, &done);
    if (!rc && !done) rc = chebhip_fail(CHEBHIP_ERR_ARG, "slab exchange: the gather launch refused a geometry the route had accepted");
    if (!rc) rc = D->x.direct_close(nf, acc, alpha, out, st);
    if (rc) chebhip::comm_abort(D->x.comm);
    return rc;
  }
  int rc = D->x.to_pencil(nf, in, st);
  if (!rc) rc = ell_op_pencil_sweep(D->op, D->x.ncol, D->x.pen_in, D->x.pen_out, stream);                   // D_0 on the pencil
  if (!rc) rc = D->x.to_slab(nf, acc, alpha, out, st);
  if (rc) chebhip::comm_abort(D->x.comm);
  return rc;
}

extern "C" int chebhip_dist_ell_destroy(chebhip_dist_ell *D) {
  if (!D) return 0;
  D->x.direct_quiesce();
  if (D->pc) chebhip_fdpc_destroy(D->pc);
  delete D->xi;
  if (D->op) ell_op_destroy(D->op);
  delete D;
  return 0;
}
extern "C" int chebhip_dist_ell_create(int d, const int *dims, chebhip_comm *comm, chebhip_dist_ell **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  chebhip_dist_ell *D = new (std::nothrow) chebhip_dist_ell;
  if (!D) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  int rc = D->x.setup(d, dims, comm, 1);
  if (!rc) rc = ell_op_create_slab(d, dims, (int)D->x.s0[D->x.rank], (int)D->x.s0[D->x.rank + 1], dell_dim0, D, &D->op);
  if (rc) { chebhip_dist_ell_destroy(D); return rc; }
  D->direct = D->x.direct_geometry() && chebhip::ell_pencil_gather_supported(D->op);
  *out = D;
  return 0;
}
extern "C" ell_op *chebhip_dist_ell_op(chebhip_dist_ell *D) { return D ? D->op : nullptr; }
// FormJacobian's preconditioner (elliptic.C:537-590) for the slab's unknowns, as chebhip_dist_stokes_pc
extern "C" int chebhip_dist_ell_pc(chebhip_dist_ell *D, chebhip_fdpc **out) {
  if (!D || !out) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  *out = nullptr;
  if (!D->pc) {
    SlabX *xi = new (std::nothrow) SlabX;
    if (!xi) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
    int rc = xi->setup_interior(D->x.d, D->x.dims.data(), D->x.comm, 1, D->x.s0);
    if (rc) { delete xi; return rc; }
    D->xi = xi;
    rc = ell_pc_create_slab(D->op, xi->s0[xi->rank], dell_pc_dim0, D, &D->pc);
    if (rc) { delete D->xi; D->xi = nullptr; D->pc = nullptr; return rc; }
  }
  *out = D->pc;
  return 0;
}
extern "C" int chebhip_dist_ell_ranges(const chebhip_dist_ell *D, long *ranges4) {
  if (!D || !ranges4) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  D->x.ranges(ranges4);
  return 0;
}
