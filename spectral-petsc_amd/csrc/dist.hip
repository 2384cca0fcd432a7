// dist.hip -- the slab-partitioned linear Poisson matvec of BASELINE config 3, host side in C++ behind the C ABI
// (the reference itself is strictly serial: elliptic.C:262 VecCreateSeq, nk.c:63).
//
// Everything lives in the interior layout of the reference's global vector (M0, M1, ..) = dims - 2, row-major
// (SetupBC, elliptic.C:372-434).  Rank r owns the planes [s0[r], s0[r+1]) of dimension 0.  With L_k the interior
// second-derivative operator of a zero-Dirichlet line (MatMult_Elliptic with eta = 1, deta = 0, elliptic.C:297-339):
//
//   side stream                               caller's stream
//   A_1 = -L_1 U   (slab, local)              buf  = pack(U)                     blocks U[:, c1[s]:c1[s+1], :]
//   A_2 = -L_2 U   ...                        UT   = exchange(buf)               slab -> pencil (M0, m1, R)
//                                             TT   = -L_0 UT                     one launch on the pencil
//                                             T    = exchange(TT)                pencil rows -> slab blocks
//                                             V = ((T + A_1) + A_2) + ...        (after the side stream's event)
//
// The exchange chain is the critical path and stays on the caller's stream; the local sweeps, which have slack, go to the
// side stream (a cross-stream dependency costs ~13 us when the waiting stream is the one on the critical path, measured
// with the roles the other way round in round 2).  A rank's own block never passes through the exchange buffers: pack
// writes it straight into the pencil, the final sum reads it straight from the pencil result.  pack and the final sum
// work run-wise (one workgroup per (plane, peer): a contiguous run on both sides, 16-byte accesses).
//
// With option "dist_exact_order" the sum runs in the serial order k = 0, 1, 2 (elliptic.C:331-334), so every G reproduces
// the G = 1 vector to the last bits of the per-line products; by default the local terms are accumulated by the sweeps
// (one array less to write and read) and the result equals the serial one to rounding.  Two exchanges per matvec; the pencil side of the forward exchange and the
// slab-row side of the backward one need no (un)packing.
//
// Transport.  chebhip_dist_use_rccl: grouped ncclSend / ncclRecv (one RCCL launch per exchange, G-1 direct xGMI
// messages per GPU) on a communicator the caller supplies -- or one made by chebhip_rccl_comm_create from a unique
// id the host distributes.  RCCL is looked up at run time (the copy the process already holds, e.g. PyTorch's,
// else the system one) and never linked.  chebhip_dist_set_exchange plugs in any other transport with the same
// contract (the CPU tests run the exchange logic under gloo that way).
#include "comm.h"
#include "sweep.h"
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <vector>

int chebhip_fail(int code, const char *fmt, ...);   // chebhip.hip
using chebhip::XSeg;

#define DHIPCHK(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {

// ---- pack / combine -------------------------------------------------------------------------------------------
struct Split { int G; long c1[65]; };
struct APtrs { const double *p[9]; int n; };
// One workgroup per (plane i0, peer s): slab[i0, c1[s]:c1[s+1], :] is a contiguous run of (c1[s+1] - c1[s]) R doubles, and so
// is its place in the exchange buffer.  own >= 0: that peer's run goes to / comes from own_ptr (the pencil: the rank's own
// block needs no message) instead of the buffer.  V2: every run starts on a 16-byte boundary and has even length.
template <bool V2>
__global__ __launch_bounds__(256) void k_pack(Split sp, long m0, long M1, long R, const double *__restrict__ slab, double *__restrict__ buf,
                                             int own, double *__restrict__ own_ptr, long lq, long pq) {
  const int s = (int)(blockIdx.x % (unsigned)sp.G); const long i0 = blockIdx.x / (unsigned)sp.G;
  const long w = sp.c1[s + 1] - sp.c1[s], len = w * R;
  const long q = blockIdx.z;                                  // vector of a batch (chebhip_dist_mult_batch): slabs / buffers `lq` apart, pencils `pq`
  const double *src = slab + q * lq + (i0 * M1 + sp.c1[s]) * R;
  double *dst = (s == own) ? own_ptr + q * pq + i0 * len : buf + q * lq + m0 * sp.c1[s] * R + i0 * len;
  // gridDim.y workgroups share a run (at G = 2 a run is 32 k values: one workgroup per run left the chip three quarters idle)
  const long T = (long)blockDim.x * gridDim.y, t0 = (long)blockIdx.y * blockDim.x + threadIdx.x;
  if (V2) { for (long t = t0; t < (len >> 1); t += T) ((double2 *)dst)[t] = ((const double2 *)src)[t]; }
  else { for (long t = t0; t < len; t += T) dst[t] = src[t]; }
}
// V = ((T + A_1) + A_2) + ...   T in exchange order: the serial accumulation order k = 0, 1, 2 (elliptic.C:331-334)
template <bool V2>
__global__ __launch_bounds__(256) void k_combine(Split sp, long m0, long M1, long R, const double *__restrict__ buf, int own,
                                                const double *__restrict__ own_ptr, APtrs A, double *__restrict__ out, long lq, long pq) {
  const int s = (int)(blockIdx.x % (unsigned)sp.G); const long i0 = blockIdx.x / (unsigned)sp.G;
  const long q = blockIdx.z;
  const long w = sp.c1[s + 1] - sp.c1[s], len = w * R, e0 = q * lq + (i0 * M1 + sp.c1[s]) * R;
  const double *src = (s == own) ? own_ptr + q * pq + i0 * len : buf + q * lq + m0 * sp.c1[s] * R + i0 * len;
  const long T = (long)blockDim.x * gridDim.y, t0 = (long)blockIdx.y * blockDim.x + threadIdx.x;
  if (V2) {
    for (long t = t0; t < (len >> 1); t += T) {
      double2 v = ((const double2 *)src)[t];
      for (int k = 0; k < A.n; k++) { const double2 a = ((const double2 *)(A.p[k] + e0))[t]; v.x = v.x + a.x; v.y = v.y + a.y; }
      ((double2 *)(out + e0))[t] = v;
    }
  } else {
    for (long t = t0; t < len; t += T) {
      double v = src[t];
      for (int k = 0; k < A.n; k++) v = v + A.p[k][e0 + t];
      out[e0 + t] = v;
    }
  }
}
// ---- direct transports (comm.h): no messages -- a rank's kernels read the peers' arrays in place --------------------------------
// Forward: the pencil of rank r is filled from the slabs of all ranks, plane by plane -- pack and exchange in ONE launch.  One
// workgroup per global plane i (owner s, its local plane pl = i - s0[s]): the run U_s[pl, c1r : c1r + w, :] of w R doubles lands
// as row i of the pencil.  NULL transport: every peer pointer is the rank's own slab; pmax[s] clamps the plane so that the reads
// stay inside it (the timing is that of a rank, the values mean nothing).
struct PullSrc { const double *p[64]; long s0[65]; long lq[64]; long pmax[64]; long pitch[64]; };
template <bool V2>
__global__ __launch_bounds__(256) void k_pull_pack(PullSrc src, int G, long M1, long R, long c1r, long w, double *__restrict__ UT, long pq) {
  const long i = blockIdx.x, q = blockIdx.z;
  int s = 0;
  while (s + 1 < G && i >= src.s0[s + 1]) s++;
  long pl = i - src.s0[s]; if (pl > src.pmax[s] - 1) pl = src.pmax[s] - 1;
  const long len = w * R;
  const double *from = src.p[s] + q * src.lq[s] + (pl * M1 + c1r) * R;
  double *dst = UT + q * pq + i * len;
  const long T = (long)blockDim.x * gridDim.y, t0 = (long)blockIdx.y * blockDim.x + threadIdx.x;
  if (V2) { for (long t = t0; t < (len >> 1); t += T) ((double2 *)dst)[t] = ((const double2 *)from)[t]; }
  else { for (long t = t0; t < len; t += T) dst[t] = from[t]; }
}
// Push mode, a rank whose pencil sweep could not store into the owners' arrays itself (geometry the gather kernels do not take): row i
// of its pencil result TT (w R doubles) goes to plane i - s0[s] of the owner's result array, columns c1r .. c1r + w -- k_pull_pack backwards.
struct PushDst { double *p[64]; long s0[65]; long lq[64]; long pmax[64]; };
template <bool V2>
__global__ __launch_bounds__(256) void k_push_unpack(PushDst dst, int G, long M1, long R, long c1r, long w, const double *__restrict__ TT, long pq) {
  const long i = blockIdx.x, q = blockIdx.z;
  int s = 0;
  while (s + 1 < G && i >= dst.s0[s + 1]) s++;
  long pl = i - dst.s0[s]; if (pl > dst.pmax[s] - 1) pl = dst.pmax[s] - 1;
  const long len = w * R;
  double *to = dst.p[s] + q * dst.lq[s] + (pl * M1 + c1r) * R;
  const double *from = TT + q * pq + i * len;
  const long T = (long)blockDim.x * gridDim.y, t0 = (long)blockIdx.y * blockDim.x + threadIdx.x;
  if (V2) { for (long t = t0; t < (len >> 1); t += T) ((double2 *)to)[t] = ((const double2 *)from)[t]; }
  else { for (long t = t0; t < len; t += T) to[t] = from[t]; }
}
// Backward: V = ((T + A_1) + A_2) + ... with T read straight from the peers' pencil results -- exchange and final sum in ONE launch.
// One workgroup per (own plane i0, peer s): the run of w_s R doubles at row s0r + i0 of TT_s (row pitch src.pitch[s] doubles).
template <bool V2>
__global__ __launch_bounds__(256) void k_pull_combine(Split sp, PullSrc src, long s0r, long M1, long R, APtrs A, double *__restrict__ out, long lq) {
  const int s = (int)(blockIdx.x % (unsigned)sp.G); const long i0 = blockIdx.x / (unsigned)sp.G;
  const long q = blockIdx.z;
  const long w = sp.c1[s + 1] - sp.c1[s], len = w * R, e0 = q * lq + (i0 * M1 + sp.c1[s]) * R;
  const double *from = src.p[s] + q * src.lq[s] + (s0r + i0) * src.pitch[s];
  const long T = (long)blockDim.x * gridDim.y, t0 = (long)blockIdx.y * blockDim.x + threadIdx.x;
  if (V2) {
    for (long t = t0; t < (len >> 1); t += T) {
      double2 v = ((const double2 *)from)[t];
      for (int k = 0; k < A.n; k++) { const double2 a = ((const double2 *)(A.p[k] + e0))[t]; v.x = v.x + a.x; v.y = v.y + a.y; }
      ((double2 *)(out + e0))[t] = v;
    }
  } else {
    for (long t = t0; t < len; t += T) {
      double v = from[t];
      for (int k = 0; k < A.n; k++) v = v + A.p[k][e0 + t];
      out[e0 + t] = v;
    }
  }
}

static void split_sizes(long n, int parts, std::vector<long> &sz) { sz.resize(parts); for (int i = 0; i < parts; i++) sz[i] = n / parts + (i < n % parts ? 1 : 0); }

}  // namespace

struct chebhip_dist {
  int d = 0, G = 1, rank = 0;
  std::vector<long> M;                        // interior extents
  long R = 1;                                 // trailing dimensions flattened
  std::vector<long> m0, m1, s0, s1;           // plane / column counts and offsets per rank
  long local = 0, pencil = 0;
  std::vector<long> fwd_send, fwd_recv;       // doubles per peer (backward: roles swap)
  // plans and work arrays of a matvec on `nrhs` vectors at a time (w[1]: chebhip_dist_mult; w[n]: chebhip_dist_mult_batch, made on first
  // use): the stacked slabs are one tensor (nrhs m0, M1, ..), the stacked pencils one tensor (nrhs, M0, m1, ..) -- ONE launch per
  // direction and ONE grouped exchange each way for all the vectors
  struct Work {
    int nrhs = 1;
    std::vector<cheb_plan *> slab_plan;       // directions 1..d-1 on the slab(s)
    cheb_plan *pencil_plan = nullptr;         // direction 0 on the pencil(s)
    std::vector<double *> A;                  // d-1 local contributions
    double *sendbuf = nullptr, *recvbuf = nullptr, *UT = nullptr, *TT = nullptr;
    double *Tin = nullptr;                    // direct transports, push mode: -L_0 U on this rank's slab, written by the ranks that own the columns
  };
  std::map<int, Work *> w;
  hipStream_t side = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  chebhip_exchange_fn xfn = nullptr; void *xctx = nullptr;
  chebhip_comm *comm = nullptr, *own_comm = nullptr;   // transport (comm.hip); own_comm: made by chebhip_dist_use_rccl
  std::vector<XSeg> segs;
  Split split;
  bool used_direct = false;                   // a matvec has run with the peers reading this rank's arrays in place (chebhip_dist_destroy)
};

static void dist_work_free(chebhip_dist::Work *W) {
  if (!W) return;
  for (auto p : W->slab_plan) if (p) cheb_plan_destroy(p);
  if (W->pencil_plan) cheb_plan_destroy(W->pencil_plan);
  for (auto p : W->A) if (p) (void)hipFree(p);
  double *all[] = {W->sendbuf, W->recvbuf, W->UT, W->TT, W->Tin};
  for (double *p : all) if (p) (void)hipFree(p);
  delete W;
}

extern "C" int chebhip_dist_destroy(chebhip_dist *D) {
  if (!D) return 0;
  if (D->used_direct && D->comm) {
    // the peers' kernels read this rank's pencil results in place: nothing is freed before every rank has drained its device
    // (collective among the rank threads; an aborted group returns at once)
    (void)hipDeviceSynchronize();
    (void)chebhip::comm_group_barrier(D->comm);
  }
  for (auto &kv : D->w) dist_work_free(kv.second);
  if (D->side) (void)hipStreamDestroy(D->side);
  if (D->ev_in) (void)hipEventDestroy(D->ev_in);
  if (D->ev_out) (void)hipEventDestroy(D->ev_out);
  if (D->own_comm) chebhip_comm_destroy(D->own_comm);
  delete D;
  return 0;
}

// builds the work set for `nrhs` vectors; on any failure everything made so far is released and nothing is kept
static int dist_work_build(chebhip_dist *D, int nrhs, chebhip_dist::Work *W) {
  W->nrhs = nrhs;
  const int d = D->d, rank = D->rank;
  if ((double)D->local * nrhs > 2.0e9 || (double)D->pencil * nrhs > 2.0e9) return chebhip_fail(CHEBHIP_ERR_DIMS, "batch of %d vectors: more than 2^31 values per rank", nrhs);
  {  // plans on the stored (interior) tensors: slabs (nrhs m0, M1, M2..), pencils (nrhs, M0, m1, M2..)
    std::vector<int> sd(d), pd(d + 1);
    for (int k = 0; k < d; k++) { sd[k] = (int)D->M[k]; pd[k + 1] = (int)D->M[k]; }
    sd[0] = (int)(D->m0[rank] * nrhs); pd[0] = nrhs; pd[2] = (int)D->m1[rank];
    W->slab_plan.assign(d, nullptr);
    for (int k = 1; k < d; k++) { int rc = cheb_plan_create_trimmed(d, k, sd.data(), &W->slab_plan[k]); if (rc) return rc; }
    int rc = nrhs == 1 ? cheb_plan_create_trimmed(d, 0, pd.data() + 1, &W->pencil_plan) : cheb_plan_create_trimmed(d + 1, 1, pd.data(), &W->pencil_plan);
    if (rc) return rc;
  }
  // (the pencils carry one row of slack: with the NULL transport a rank reads its own pencil with the peers' column counts)
  long wmax = 0; for (int s = 0; s < D->G; s++) wmax = D->m1[s] > wmax ? D->m1[s] : wmax;
  const size_t lb = (size_t)(D->local > 0 ? D->local : 1) * nrhs * sizeof(double), pb = ((size_t)(D->pencil > 0 ? D->pencil : 1) * nrhs + (size_t)(wmax + 1) * D->R) * sizeof(double);
  W->A.assign(d - 1, nullptr);
  for (int k = 0; k < d - 1; k++) DHIPCHK(hipMalloc((void **)&W->A[k], lb));
  DHIPCHK(hipMalloc((void **)&W->sendbuf, lb)); DHIPCHK(hipMalloc((void **)&W->recvbuf, lb));
  DHIPCHK(hipMalloc((void **)&W->UT, pb)); DHIPCHK(hipMalloc((void **)&W->TT, pb));
  DHIPCHK(hipMalloc((void **)&W->Tin, lb + 16384));
  DHIPCHK(hipMemset(W->Tin, 0, lb + 16384));     // (the NULL transport sums blocks of it that nobody writes)
  return 0;
}

// the work set for `nrhs` vectors per call (made on first use).  It enters the handle's map only when it is COMPLETE: a set whose
// size check, plans or allocations failed is released at once, so that a later call with the same nrhs starts from nothing instead
// of finding a half-built entry (an out-of-memory failure followed by a retry would have launched on NULL buffers).
static int dist_work(chebhip_dist *D, int nrhs, chebhip_dist::Work **out) {
  auto it = D->w.find(nrhs);
  if (it != D->w.end()) { if (out) *out = it->second; return 0; }
  chebhip_dist::Work *W = new (std::nothrow) chebhip_dist::Work;
  if (!W) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  const int rc = dist_work_build(D, nrhs, W);
  if (rc) { dist_work_free(W); return rc; }
  D->w[nrhs] = W;
  if (out) *out = W;
  return 0;
}

extern "C" int chebhip_dist_create(int d, const int *dims, int nranks, int rank, chebhip_dist **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!dims || d < 2 || d > 10) return chebhip_fail(CHEBHIP_ERR_DIMS, "slab partitioning needs 2 <= d <= 10");
  if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return chebhip_fail(CHEBHIP_ERR_ARG, "rank %d of %d", rank, nranks);
  // (3..256: what the interior-layout plans take, cheb_plan_create_trimmed -- the matrix L = (D D)[1..n-1, 1..n-1] lives in the
  // register file of a workgroup; longer lines have no slab driver)
  for (int k = 0; k < d; k++) if (dims[k] < 3 || dims[k] > 256) return chebhip_fail(CHEBHIP_ERR_SIZE, "dims[%d] = %d must be in 3..256", k, dims[k]);
  if (dims[0] - 2 < nranks || dims[1] - 2 < nranks)
    return chebhip_fail(CHEBHIP_ERR_SIZE, "every rank needs at least one interior plane along dimensions 0 and 1");
  chebhip_dist *D = new (std::nothrow) chebhip_dist;
  if (!D) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  D->d = d; D->G = nranks; D->rank = rank;
  D->M.resize(d); for (int k = 0; k < d; k++) D->M[k] = dims[k] - 2;
  D->R = 1; for (int k = 2; k < d; k++) D->R *= D->M[k];
  split_sizes(D->M[0], nranks, D->m0); split_sizes(D->M[1], nranks, D->m1);
  D->s0.assign(nranks + 1, 0); D->s1.assign(nranks + 1, 0);
  for (int s = 0; s < nranks; s++) { D->s0[s + 1] = D->s0[s] + D->m0[s]; D->s1[s + 1] = D->s1[s] + D->m1[s]; }
  D->local = D->m0[rank] * D->M[1] * D->R;
  D->pencil = D->M[0] * D->m1[rank] * D->R;
  D->fwd_send.resize(nranks); D->fwd_recv.resize(nranks);
  for (int s = 0; s < nranks; s++) { D->fwd_send[s] = D->m0[rank] * D->m1[s] * D->R; D->fwd_recv[s] = D->m0[s] * D->m1[rank] * D->R; }
  D->split.G = nranks; for (int s = 0; s <= nranks; s++) D->split.c1[s] = D->s1[s];
#define DHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { chebhip_dist_destroy(D); \
    return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
  { int rc = dist_work(D, 1, nullptr); if (rc) { chebhip_dist_destroy(D); return rc; } }
  DHIP(hipStreamCreateWithFlags(&D->side, hipStreamNonBlocking));
  DHIP(hipEventCreateWithFlags(&D->ev_in, hipEventDisableTiming));
  DHIP(hipEventCreateWithFlags(&D->ev_out, hipEventDisableTiming));
#undef DHIP
  *out = D;
  return 0;
}

extern "C" long chebhip_dist_local_size(const chebhip_dist *D) { return D ? D->local : -1; }
extern "C" long chebhip_dist_slab_offset(const chebhip_dist *D) { return D ? D->s0[D->rank] * D->M[1] * D->R : -1; }

extern "C" int chebhip_dist_set_exchange(chebhip_dist *D, chebhip_exchange_fn fn, void *ctx) {
  if (!D) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL handle");
  D->xfn = fn; D->xctx = ctx; D->comm = nullptr;
  return 0;
}
extern "C" int chebhip_dist_use_comm(chebhip_dist *D, chebhip_comm *comm) {
  if (!D || !comm) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (chebhip::comm_size(comm) != D->G || chebhip::comm_rank(comm) != D->rank)
    return chebhip_fail(CHEBHIP_ERR_ARG, "communicator is rank %d of %d, the handle rank %d of %d", chebhip::comm_rank(comm), chebhip::comm_size(comm), D->rank, D->G);
  D->comm = comm; D->xfn = nullptr; D->xctx = nullptr;
  return 0;
}
extern "C" int chebhip_dist_use_rccl(chebhip_dist *D, void *nccl_comm) {
  if (!D || !nccl_comm) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (D->own_comm) { chebhip_comm_destroy(D->own_comm); D->own_comm = nullptr; }
  int rc = chebhip_comm_create_rccl(nccl_comm, D->G, D->rank, &D->own_comm); if (rc) return rc;
  return chebhip_dist_use_comm(D, D->own_comm);
}

// send[s] (doubles, peer-major, contiguous) -> recv[s], for each of nrhs vectors (their buffers lq / pq doubles apart): ONE grouped
// launch of the transport.  own >= 0: that rank's block is not moved (the kernels on either side read / write it in place)
static int exchange(chebhip_dist *D, int nrhs, const double *send, long sq, const long *sc, double *recv, long rq, const long *rc_, int own, hipStream_t st) {
  if (D->xfn) {
    if (nrhs != 1) return chebhip_fail(CHEBHIP_ERR_ARG, "chebhip_dist_mult_batch needs a chebhip_comm transport (chebhip_dist_use_comm / _use_rccl)");
    return D->xfn(D->xctx, send, sc, recv, rc_, (void *)st);
  }
  if (D->G > 1 && !D->comm) return chebhip_fail(CHEBHIP_ERR_ARG, "chebhip_dist: no transport set (chebhip_dist_use_comm / _use_rccl / _set_exchange)");
  D->segs.clear();
  for (int q = 0; q < nrhs; q++) {           // (the k-th segment for a peer meets that peer's k-th segment for this rank: both run q = 0, 1, ..)
    long so = 0, ro = 0;
    for (int s = 0; s < D->G; s++) { if (s != own) D->segs.push_back(XSeg{s, send + q * sq + so, sc[s], recv + q * rq + ro, rc_[s]}); so += sc[s]; ro += rc_[s]; }
  }
  if (D->segs.empty()) return 0;
  return chebhip::comm_exchange(D->comm, D->segs.data(), (int)D->segs.size(), st);
}

constexpr long FUSE3_MAX = 12000000L;      // values per rank and batch up to which the three directions of a direct-transport matvec are ONE launch

static int dist_mult(chebhip_dist *D, chebhip_dist::Work *W, const double *U, double *V, hipStream_t st) {
  const int d = D->d, r = D->rank, nrhs = W->nrhs;
  const long m0 = D->m0[r], M1 = D->M[1], R = D->R;
  const long lq = D->local, pq = D->pencil;                  // vector q of a batch: slab-sized arrays lq apart, pencils pq apart
  // Direct transports (LOCAL thread ranks with peer access, NULL): no pack, no messages, no unpack -- a rank's kernels read the peers'
  // slabs and pencil results in place, ordered by two host rendezvous per matvec.  Option "dist_packed_exchange" = 1: pack /
  // comm_exchange / combine as for RCCL (A/B, tests).
  const int pk = chebhip::opt(chebhip::OPT_DIST_PACKED_EXCHANGE);
  const bool direct = !D->xfn && D->comm && chebhip::comm_direct(D->comm) && pk != 1 && D->G <= 64;
  // "dist_single_stream": 1 = the local sweeps stay on the caller's stream, 2 = always on the side stream, 0 (default) = by transport:
  // the side stream overlaps the local sweeps with the exchanges, and costs 8-27 % when nothing travels off the device (one rank, the
  // NULL transport, thread ranks sharing one GPU: bench.py dist_rank_compute, round 5) -- there it is not used.
  const int ss = chebhip::opt(chebhip::OPT_DIST_SINGLE_STREAM);
  const bool one_stream = ss == 1 || (ss == 0 && !(D->xfn ? D->G > 1 : chebhip::comm_overlap_pays(D->comm)));
  hipStream_t side = one_stream ? st : D->side;
  // "dist_exact_order" = 0 (default): the local terms are summed by the sweeps, A_1 = -L_1 U (STORE), A_1 -= L_k U (ACC): the
  // final sum reads T and ONE array, V = T + (A_1 + A_2 ..) -- the serial vector to rounding (north_star 1e-10);
  // 1: every term in an array of its own and V = ((T + A_1) + A_2) in the order of elliptic.C:331-334 -- the serial bits.
  bool exact = chebhip::opt(chebhip::OPT_DIST_EXACT_ORDER) != 0;
  int rc = 0;
  hipError_t e1 = hipSuccess;
  // Small slabs (fewer than 6 M values: 256^3 over 4 ranks and more), d >= 3: the d - 1 local directions are ONE launch of d - 1 jobs
  // into arrays of their own, and the final sum reads them in the serial order -- at these sizes a launch costs its fixed 10-13 us
  // (matrix fetch, fill, drain), not its bytes, and the extra array sits in the Infinity Cache.  The one-GPU bits, as with the option.
  const bool small = d >= 3 && D->local > 0 && D->local * nrhs < 6000000L && !chebhip::opt(chebhip::OPT_SEPARATE_LAUNCHES);
  // the local directions, each into its own array (on the side stream they overlap both exchanges); U is ready when the caller's stream gets here
  auto local_sweeps = [&]() {
    if (!one_stream) { hipError_t e = hipEventRecord(D->ev_in, st); if (e == hipSuccess) e = hipStreamWaitEvent(side, D->ev_in, 0); if (e != hipSuccess) { rc = chebhip_fail(CHEBHIP_ERR_DEVICE, "chebhip_dist_mult: stream fork failed"); return; } }
    bool local_done = false;
    if (small) {
      rc = chebhip::lap1d_multi_try(d - 1, W->slab_plan.data() + 1, U, W->A.data(), -1.0, side, &local_done);
      if (local_done) exact = true;
    }
    for (int k = 1; k < d && !rc && !local_done; k++) {
      if (exact || k == 1) rc = cheb_apply_lap1d(W->slab_plan[k], U, nullptr, -1.0, W->A[exact ? k - 1 : 0], side);
      else rc = cheb_apply_lap1d(W->slab_plan[k], U, W->A[0], -1.0, W->A[0], side);
    }
    e1 = one_stream ? hipSuccess : hipEventRecord(D->ev_out, side);
  };
  // 16-byte accesses: every run (c1[s+1] - c1[s]) R long starting at (i0 M1 + c1[s]) R must be even-aligned
  bool v2 = ((R & 1) == 0 || ((M1 & 1) == 0)) && (((size_t)U | (size_t)V) & 15) == 0 && (nrhs == 1 || ((lq & 1) == 0 && (pq & 1) == 0));
  if (v2 && (R & 1)) for (int s = 0; s <= D->G; s++) v2 = v2 && (D->s1[s] & 1) == 0;
  const unsigned grid1 = (unsigned)(m0 * D->G);
  // workgroups per run: about 2048 values (1024 16-byte pieces) per workgroup pass, at most 64
  unsigned gy = 1;
  { long wmax = 0; for (int s = 0; s < D->G; s++) wmax = D->m1[s] > wmax ? D->m1[s] : wmax; const long len = wmax * R; gy = (unsigned)((len + 2047) / 2048); if (gy < 1) gy = 1; if (gy > 64) gy = 64; }
  const dim3 grid(grid1, gy, (unsigned)nrhs);

  if (direct) {
    D->used_direct = true;
    const bool null = chebhip::comm_is_null(D->comm);
    chebhip::PeerView pv;
    PullSrc ps;
    // PUSH (default): every rank also posts its result array Tin, and the pencil sweeps store each output row into the array of the
    // rank that owns the plane -- remote stores in the same launch as the remote loads, both directions of every link at once -- so
    // that the final sum reads local memory only.  dist_packed_exchange = 4: the pull form (pencil results stay where they are
    // computed, the final sum reads the peers' TT), kept for A/B and tests.  The choice depends on nothing a rank knows alone.
    const bool push = pk != 4;
    const double *post[2] = {U, push ? W->Tin : nullptr};
    // slot 0: "my U is complete"; wait for the peers' slot 3 of the previous matvec: they have finished reading my TT / my own final
    // sum has finished reading my Tin (push: the peers' stores of THIS matvec land in it), both rewritten below
    rc = chebhip::comm_rendezvous(D->comm, post, 2, 0, 3, st, &pv);
    if (!rc && push) for (int s = 0; s < D->G; s++) if (!pv.ptr[s][1]) rc = chebhip_fail(CHEBHIP_ERR_ARG, "chebhip_dist_mult: rank %d posted no result array (the ranks disagree on dist_packed_exchange)", s);
    // The pencil direction reads the peers' slabs IN PLACE where it can (strided lines of 66 .. 256 points, 16-byte geometry, at most
    // GATHER_MAX ranks): the loader slots of a thread are fixed rows = fixed planes of fixed peers, so the pencil is never materialised.
    // On a small slab it is one more job of the local directions' launch: the THREE directions of the matvec are then ONE launch
    // (one matrix fetch, one pipeline fill; the local jobs run while the pencil job's remote reads are in flight) and the matvec is
    // two kernels.  Otherwise (and with dist_packed_exchange = 2, the A/B switch) k_pull_pack fills UT first.
    bool gathered = false, fused = false;
    chebhip::GatherSrc gsrc = {};
    bool gfits = !rc && D->pencil > 0 && D->G <= chebhip::GATHER_MAX && pk != 2 && M1 * R < 0x7fffffffL;
    if (gfits) {
      gsrc.G = D->G; gsrc.rowlen = (unsigned)(M1 * R); gsrc.col0 = (unsigned)(D->s1[r] * R);
      for (int s = 0; s < D->G; s++) {
        const long lqs = null ? lq : D->m0[s] * M1 * R;
        gfits = gfits && lqs < 0x7fffffffL;
        gsrc.p[s] = pv.ptr[s][0]; gsrc.s0[s] = (int)D->s0[s]; gsrc.lq[s] = (unsigned)lqs; gsrc.pmax[s] = (int)(null ? (m0 > 0 ? m0 : 1) : D->m0[s]);
        gsrc.dp[s] = push ? (double *)pv.ptr[s][1] : nullptr;
      }
      gsrc.s0[D->G] = (int)D->s0[D->G];
      gsrc.push = push ? 1 : 0;
    }
    // (the one-launch form pays up to about 12 M values per rank and batch -- 8 B/point more in the final sum for two launches less;
    // measured at G = 2 and at four vectors per exchange over 8 ranks, profiles/r06_dist/fused_threshold.txt)
    const bool fuse3 = d >= 3 && D->local > 0 && D->local * nrhs < FUSE3_MAX && !chebhip::opt(chebhip::OPT_SEPARATE_LAUNCHES);
    if (!rc && gfits && fuse3 && pk != 3) {
      rc = chebhip::lap1d_multi_gather_try(d - 1, W->slab_plan.data() + 1, U, W->A.data(), W->pencil_plan, gsrc, W->TT, -1.0, st, &fused);
      if (fused) { exact = true; gathered = true; }
    }
    if (!rc && !fused) local_sweeps();
    if (!rc && gfits && !gathered) rc = chebhip::lap1d_gather_try(W->pencil_plan, gsrc, -1.0, W->TT, st, &gathered);        // TT = -L_0 (the peers' slabs)
    if (!rc && D->pencil > 0 && !gathered) {
      for (int s = 0; s < D->G; s++) {
        ps.p[s] = pv.ptr[s][0]; ps.s0[s] = D->s0[s]; ps.lq[s] = null ? lq : D->m0[s] * M1 * R; ps.pmax[s] = null ? (m0 > 0 ? m0 : 1) : D->m0[s]; ps.pitch[s] = 0;
      }
      ps.s0[D->G] = D->s0[D->G];
      const long wr = D->m1[r]; const long len = wr * R;
      unsigned gyp = (unsigned)((len + 2047) / 2048); if (gyp < 1) gyp = 1; if (gyp > 64) gyp = 64;
      const dim3 gridp((unsigned)D->M[0], gyp, (unsigned)nrhs);
      bool v2p = (((R & 1) == 0) || ((M1 & 1) == 0 && (D->s1[r] & 1) == 0 && (wr & 1) == 0)) && (nrhs == 1 || (pq & 1) == 0);
      for (int s = 0; s < D->G && v2p; s++) v2p = ((size_t)ps.p[s] & 15) == 0 && (nrhs == 1 || (ps.lq[s] & 1) == 0);
      if (v2p) hipLaunchKernelGGL((k_pull_pack<true>), gridp, dim3(256), 0, st, ps, D->G, M1, R, D->s1[r], wr, W->UT, pq);
      else hipLaunchKernelGGL((k_pull_pack<false>), gridp, dim3(256), 0, st, ps, D->G, M1, R, D->s1[r], wr, W->UT, pq);
      if (hipGetLastError() != hipSuccess) rc = chebhip_fail(CHEBHIP_ERR_DEVICE, "k_pull_pack launch failed");
      if (!rc) rc = cheb_apply_lap1d(W->pencil_plan, W->UT, nullptr, -1.0, W->TT, st);                                    // TT = -L_0 UT
    }
    if (!rc && push && D->pencil > 0 && !gathered) {                                                                      // ... and its rows to their owners
      PushDst pd;
      for (int s = 0; s < D->G; s++) { pd.p[s] = (double *)pv.ptr[s][1]; pd.s0[s] = D->s0[s]; pd.lq[s] = null ? lq : D->m0[s] * M1 * R; pd.pmax[s] = null ? (m0 > 0 ? m0 : 1) : D->m0[s]; }
      pd.s0[D->G] = D->s0[D->G];
      const long wr = D->m1[r]; const long len = wr * R;
      unsigned gyp = (unsigned)((len + 2047) / 2048); if (gyp < 1) gyp = 1; if (gyp > 64) gyp = 64;
      const dim3 gridp((unsigned)D->M[0], gyp, (unsigned)nrhs);
      bool v2p = (((R & 1) == 0) || ((M1 & 1) == 0 && (D->s1[r] & 1) == 0 && (wr & 1) == 0)) && (nrhs == 1 || (pq & 1) == 0);
      for (int s = 0; s < D->G && v2p; s++) v2p = ((size_t)pd.p[s] & 15) == 0 && (nrhs == 1 || (pd.lq[s] & 1) == 0);
      if (v2p) hipLaunchKernelGGL((k_push_unpack<true>), gridp, dim3(256), 0, st, pd, D->G, M1, R, D->s1[r], wr, (const double *)W->TT, pq);
      else hipLaunchKernelGGL((k_push_unpack<false>), gridp, dim3(256), 0, st, pd, D->G, M1, R, D->s1[r], wr, (const double *)W->TT, pq);
      if (hipGetLastError() != hipSuccess) rc = chebhip_fail(CHEBHIP_ERR_DEVICE, "k_push_unpack launch failed");
    }
    if (!rc) rc = chebhip::comm_mark(D->comm, 2, st);                                                                     // my reads of the peers' U end here
    post[0] = push ? nullptr : W->TT; post[1] = nullptr;
    // slot 1: "my TT is complete" / push: "my stores into your Tin have landed"; wait for the peers' slot 2: they have finished reading
    // my U (the caller may rewrite it after this call)
    if (!rc) rc = chebhip::comm_rendezvous(D->comm, post, 1, 1, 2, st, &pv);
    hipError_t e2 = (one_stream || fused) ? hipSuccess : hipStreamWaitEvent(st, D->ev_out, 0);
    if (rc) { chebhip::comm_abort(D->comm); return rc; }
    if (e1 != hipSuccess || e2 != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "chebhip_dist_mult: stream join failed");
    if (grid1) {
      for (int s = 0; s < D->G; s++) {
        const long m1s = null ? D->m1[r] : D->m1[s];
        ps.p[s] = pv.ptr[s][0]; ps.pitch[s] = m1s * R; ps.lq[s] = null ? pq : D->M[0] * D->m1[s] * R;
        if (push) { ps.p[s] = W->Tin + D->s1[s] * R; ps.pitch[s] = M1 * R; ps.lq[s] = lq; }         // the slab-shaped local array, segment s of a plane
      }
      APtrs A; A.n = exact ? d - 1 : 1; for (int k = 0; k < 9; k++) A.p[k] = k < A.n ? W->A[k] : nullptr;
      bool v2c = v2;
      for (int s = 0; s < D->G && v2c; s++) v2c = ((size_t)ps.p[s] & 15) == 0 && (ps.pitch[s] & 1) == 0 && (nrhs == 1 || (ps.lq[s] & 1) == 0);
      const long s0r = push ? 0 : D->s0[r];
      if (v2c) hipLaunchKernelGGL((k_pull_combine<true>), dim3(grid), dim3(256), 0, st, D->split, ps, s0r, M1, R, A, V, lq);
      else hipLaunchKernelGGL((k_pull_combine<false>), dim3(grid), dim3(256), 0, st, D->split, ps, s0r, M1, R, A, V, lq);
    }
    DHIPCHK(hipGetLastError());
    return chebhip::comm_mark(D->comm, 3, st);                                                                          // my reads of the peers' TT end here
  }

  // Message transports (RCCL, callbacks; LOCAL with dist_packed_exchange = 1).  The exchange chain is the critical path and stays on
  // the caller's stream.  With a chebhip_exchange_fn (the older callback contract moves every block, the own one included) everything
  // goes through the buffers; otherwise the own block bypasses them.
  local_sweeps();
  const int own = (D->xfn || chebhip::opt(chebhip::OPT_RCCL_SELF_MESSAGES)) ? -1 : r;     // (the option: one-rank smoke runs of the transport)
  double *own_in = W->UT + D->s0[r] * D->m1[r] * R;          // where the own block sits in the pencil: rows s0[r] .. s0[r+1]
  const double *own_out = W->TT + D->s0[r] * D->m1[r] * R;
  if (!rc && grid1) {
    if (v2) hipLaunchKernelGGL((k_pack<true>), dim3(grid), dim3(256), 0, st, D->split, m0, M1, R, U, W->sendbuf, own, own_in, lq, pq);
    else hipLaunchKernelGGL((k_pack<false>), dim3(grid), dim3(256), 0, st, D->split, m0, M1, R, U, W->sendbuf, own, own_in, lq, pq);
    if (hipGetLastError() != hipSuccess) rc = chebhip_fail(CHEBHIP_ERR_DEVICE, "k_pack launch failed");
  }
  if (!rc) rc = exchange(D, nrhs, W->sendbuf, lq, D->fwd_send.data(), W->UT, pq, D->fwd_recv.data(), own, st);          // lands as the pencil(s)
  if (!rc) rc = cheb_apply_lap1d(W->pencil_plan, W->UT, nullptr, -1.0, W->TT, st);                                      // TT = -L_0 UT
  if (!rc) rc = exchange(D, nrhs, W->TT, pq, D->fwd_recv.data(), W->recvbuf, lq, D->fwd_send.data(), own, st);           // pencil rows -> slab blocks
  // the caller's stream is rejoined with the side stream on every path, errors included
  hipError_t e2 = one_stream ? hipSuccess : hipStreamWaitEvent(st, D->ev_out, 0);
  if (rc) { chebhip::comm_abort(D->comm); return rc; }     // thread ranks waiting for this one fail at once instead of timing out
  if (e1 != hipSuccess || e2 != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "chebhip_dist_mult: stream join failed");
  APtrs A; A.n = exact ? d - 1 : 1; for (int k = 0; k < 9; k++) A.p[k] = k < A.n ? W->A[k] : nullptr;
  if (grid1) {
    if (v2) hipLaunchKernelGGL((k_combine<true>), dim3(grid), dim3(256), 0, st, D->split, m0, M1, R, (const double *)W->recvbuf, own, own_out, A, V, lq, pq);
    else hipLaunchKernelGGL((k_combine<false>), dim3(grid), dim3(256), 0, st, D->split, m0, M1, R, (const double *)W->recvbuf, own, own_out, A, V, lq, pq);
  }
  DHIPCHK(hipGetLastError());
  return 0;
}

extern "C" int chebhip_dist_mult(chebhip_dist *D, const double *U, double *V, void *stream) {
  if (!D || !U || !V) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (U == V) return chebhip_fail(CHEBHIP_ERR_ARG, "U and V must be distinct");
  return dist_mult(D, D->w[1], U, V, (hipStream_t)stream);
}

// The same matvec on nrhs vectors at once (U, V: nrhs consecutive local vectors): one launch per direction on the stacked slabs /
// pencils, ONE pack, ONE grouped exchange each way (nrhs (G - 1) messages in it) and ONE final sum -- the fixed cost of a launch of
// 256-point lines (10-15 us: matrix fetch, fill, drain) and of an RCCL launch (~25 us) is paid once per batch, not per vector.
// Callers: the independent right-hand sides of a block / s-step Krylov method, several Newton right-hand sides, the bench's pipelined
// steps.  Each vector's result equals chebhip_dist_mult's (same kernels, same order of operations per line).  Collective.
extern "C" int chebhip_dist_mult_batch(chebhip_dist *D, int nrhs, const double *U, double *V, void *stream) {
  if (!D || !U || !V) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (U == V) return chebhip_fail(CHEBHIP_ERR_ARG, "U and V must be distinct");
  if (nrhs < 1 || nrhs > 64) return chebhip_fail(CHEBHIP_ERR_ARG, "nrhs = %d must be in 1..64", nrhs);
  if (D->xfn && nrhs != 1) return chebhip_fail(CHEBHIP_ERR_ARG, "chebhip_dist_mult_batch needs a chebhip_comm transport (chebhip_dist_use_comm / _use_rccl)");
  chebhip_dist::Work *W = nullptr;
  int rc = dist_work(D, nrhs, &W); if (rc) return rc;
  return dist_mult(D, W, U, V, (hipStream_t)stream);
}
