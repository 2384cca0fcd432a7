// krylov.hip -- restarted flexible GMRES on device vectors: the role KSPSolve plays around the operator
// callbacks (KSPFGMRES, elliptic.C:181-185; KSPSchurVelocity inside StokesMatMultSchur, stokes.C:531).
//
// The operator and the (right, possibly varying) preconditioner are callbacks on device pointers with the
// signature of ell_op_mult / stokes_op_mult_vv, so those entry points can be passed directly.  All vectors
// stay in HBM, the Hessenberg matrix and its Givens rotations too; per iteration the host reads one double
// (the residual estimate, from pinned memory) and it does so one iteration late, so the device never idles.
//   w = A M v_j;  h = V^T w (classical Gram-Schmidt, one pass -- PETSc's default orthogonalisation);
//   w -= V h;  h_{j+1,j} = |w|;  Givens rotations and the triangular solve on the device (one thread).
// Reductions are two-stage with a fixed block order: results do not depend on scheduling.
// Round 5: ONE reduction per Gram-Schmidt step (k_gs_dots / k_gs_update below): w.w is taken in the same pass as V^T w and
// w.w - |h|^2 ~ h_{j+1,j}^2 supplies the scale of the new vector, so that the update and the normalisation are one pass over the
// basis and the step is two launches instead of three; the stored vector's exact norm comes out of that pass and is carried beside
// it (gs_coefficients), so nothing of a cycle's interior rests on the cancelling difference.  The ONE place that does: the last column a
// cycle can take (the end of the restart, or the iteration limit) runs no update pass, and its h_{j+1,j} IS sqrt(w.w - |h|^2), whose
// error is (orthogonality loss) x w.w -- after 30-60 vectors the residual estimate of that column can read low.  Convergence read off
// such a column is therefore never returned as it stands: the true residual b - A x is formed first (the top of the cycle loop) and
// decides.  Option "krylov_exact_norm" = 1: the three-launch step (every column's norm exact).
#include "../../include/chebhip.h"
#include "timers.h"
#include "sweep.h"
#include <hip/hip_runtime.h>
#include <cmath>
#include <new>
#include <vector>

int chebhip_fail(int code, const char *fmt, ...);   // chebhip.hip

#define KHIPCHK(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {

constexpr int RB = 256;      // reduction blocks per dot product
constexpr int RT = 256;      // threads per block

constexpr int ST = 1024;     // threads per block of the two kernels that stream the whole basis (16 waves per CU keep HBM busy;
                             // with 256 the orthogonalisation of a 256^3 solve ran at 3 TB/s and was 34 % of its device time)

template <int NT>
__device__ __forceinline__ double block_sum_t(double v, double *sh) {
  // wave reduction by DPP-free shuffles, then the wave sums through LDS in a fixed order
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) { for (int q = 0; q < NT / 64; q++) r += sh[q]; }
  __syncthreads();
  return r;   // valid in thread 0
}
__device__ __forceinline__ double block_sum(double v, double *sh) { return block_sum_t<RT>(v, sh); }

// part[kk][b] = sum over chunk b of V[kk][i] * w[i],  kk = blockIdx.y < k
__global__ __launch_bounds__(RT) void k_multidot(long n, const double *__restrict__ V, long ldv, const double *__restrict__ w,
                                                 double *__restrict__ part) {
  __shared__ double sh[RT / 64];
  const double *v = V + (long)blockIdx.y * ldv;
  double s = 0.0;
  for (long i = blockIdx.x * (long)RT + threadIdx.x; i < n; i += (long)RB * RT) s += v[i] * w[i];
  const double r = block_sum(s, sh);
  if (threadIdx.x == 0) part[(long)blockIdx.y * RB + blockIdx.x] = r;
}

// The same for k rows with w read once per group of KB rows instead of once per row (the orthogonalisation of restarted
// GMRES is the largest mover of bytes in a preconditioned solve at 256^3): block (b, g) forms the partial sums of the rows
// g KB .. g KB + KB - 1 over chunk b, in a fixed order (results do not depend on scheduling).
constexpr int KB = 8;
// V2: 16-byte loads (n even, every row and w 16-B aligned: ld is even, the basis allocation aligned).  A thread then carries two
// partial sums per row, over the even and the odd elements of its stride, added at the end -- a different but equally fixed order.
// (8-byte loads: the inner solves of the Stokes preconditioners, 2..5 rows of 50 MB at 128^3, ran this kernel at 3.8 TB/s.)
template <bool V2>
__global__ __launch_bounds__(ST) void k_multidot_grouped(long n, int k, const double *__restrict__ V, long ldv, const double *__restrict__ w,
                                                         double *__restrict__ part) {
  __shared__ double sh[ST / 64];
  const int kk0 = blockIdx.y * KB;
  const int cnt = k - kk0 < KB ? k - kk0 : KB;          // rows of this group (the inner solves of the Stokes preconditioner have 1..4)
  const double *v[KB];
#pragma unroll
  for (int q = 0; q < KB; q++) v[q] = V + (long)(kk0 + (q < cnt ? q : 0)) * ldv;
  double s[KB];
#pragma unroll
  for (int q = 0; q < KB; q++) s[q] = 0.0;
  if (V2) {
    double t[KB];
#pragma unroll
    for (int q = 0; q < KB; q++) t[q] = 0.0;
    const long n2 = n >> 1;
    if (cnt == KB) {
      for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n2; i += (long)RB * ST) {
        const double2 wi = ((const double2 *)w)[i];
#pragma unroll
        for (int q = 0; q < KB; q++) { const double2 a = ((const double2 *)v[q])[i]; s[q] += a.x * wi.x; t[q] += a.y * wi.y; }
      }
    } else {
      for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n2; i += (long)RB * ST) {
        const double2 wi = ((const double2 *)w)[i];
#pragma unroll
        for (int q = 0; q < KB; q++) if (q < cnt) { const double2 a = ((const double2 *)v[q])[i]; s[q] += a.x * wi.x; t[q] += a.y * wi.y; }
      }
    }
#pragma unroll
    for (int q = 0; q < KB; q++) s[q] += t[q];
  } else if (cnt == KB) {
    for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n; i += (long)RB * ST) {
      const double wi = w[i];
#pragma unroll
      for (int q = 0; q < KB; q++) s[q] += v[q][i] * wi;
    }
  } else {
    for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n; i += (long)RB * ST) {
      const double wi = w[i];
#pragma unroll
      for (int q = 0; q < KB; q++) if (q < cnt) s[q] += v[q][i] * wi;   // uniform: no load for a row that is not there
    }
  }
#pragma unroll
  for (int q = 0; q < KB; q++) {
    const double r = block_sum_t<ST>(s[q], sh);
    if (threadIdx.x == 0 && kk0 + q < k) part[(long)(kk0 + q) * RB + blockIdx.x] = r;
  }
}

// out[row] = sum_b part[row][b] (fixed order), optionally the square root; one block per row
__global__ __launch_bounds__(RT) void k_rows_finish(const double *__restrict__ part, double *__restrict__ out, int take_sqrt) {
  __shared__ double sh[RT / 64];
  const double r = block_sum(part[(long)blockIdx.x * RB + threadIdx.x], sh);
  if (threadIdx.x == 0) out[blockIdx.x] = take_sqrt ? sqrt(r) : r;
}

__global__ void k_sqrt1(double *__restrict__ p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = sqrt(p[0]); }

// h[kk] = sum_b dpart[kk][b] (every block forms the same sums in the same order; block 0 publishes them);
// w -= sum_{kk<k} h[kk] V[kk];  npart[b] = sum over chunk b of w^2
__global__ __launch_bounds__(ST) void k_orth_update(long n, int k, const double *__restrict__ V, long ldv,
                                                    const double *__restrict__ dpart, double *hcol,
                                                    double *__restrict__ w, double *__restrict__ npart, int store) {   // store = 0: only |w - V h|^2 is wanted
  __shared__ double sh[ST / 64];
  __shared__ double h[ST];
  for (int kk = threadIdx.x; kk < k; kk += ST) {
    double s = 0.0;
    if (dpart) { for (int q = 0; q < RB; q++) s += dpart[(long)kk * RB + q]; if (blockIdx.x == 0) hcol[kk] = s; }
    else s = hcol[kk];                            // several ranks: the sums were completed by the reduction callback
    h[kk] = s;
  }
  __syncthreads();
  double s = 0.0;
  for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n; i += (long)RB * ST) {
    double x = w[i];
    int kk = 0;
    for (; kk + 4 <= k; kk += 4) {                // four basis rows in flight per thread; the subtractions stay in row order
      const double a0 = V[(long)kk * ldv + i], a1 = V[(long)(kk + 1) * ldv + i], a2 = V[(long)(kk + 2) * ldv + i], a3 = V[(long)(kk + 3) * ldv + i];
      x -= h[kk] * a0; x -= h[kk + 1] * a1; x -= h[kk + 2] * a2; x -= h[kk + 3] * a3;
    }
    for (; kk < k; kk++) x -= h[kk] * V[(long)kk * ldv + i];
    if (store) w[i] = x;
    s += x * x;
  }
  const double r = block_sum_t<ST>(s, sh);
  if (threadIdx.x == 0) npart[blockIdx.x] = r;
}

// Column j of the Hessenberg matrix on the device and the normalisation of the new basis vector, ONE launch (round 4: two;
// the inner solves of the Stokes preconditioners are a few iterations of dependent 5-us kernels each):
//   every block: h_{j+1,j} = sqrt(sum npart) -- the same 256 partials summed in the same order by every block, so every block
//   holds the same value -- and w *= 1 / h_{j+1,j} on its share of w (scale = 0: the basis vector will not be used, skip);
//   block 0, thread 0: the previous Givens rotations, the new one, the rotated right-hand side G_{j+1} (a fresh copy: iterations
//   issued speculatively beyond the converged one must not disturb what the solve reads), the residual estimate for the host
//   (pinned memory).  A zero or non-finite h_{j+1,j} (breakdown: the column is dropped by the host) scales by 0.
__global__ __launch_bounds__(RT) void k_givens_scale(int j, int m, const double *__restrict__ npart, const double *__restrict__ hcol,
                                                     double *__restrict__ H, double *__restrict__ cs, double *__restrict__ sn,
                                                     double *__restrict__ G, double *__restrict__ res,
                                                     const double *__restrict__ nsq, long n, double *__restrict__ w, int scale) {
  __shared__ double sh[RT / 64];
  __shared__ double hn;
  const double ss = npart ? block_sum(npart[threadIdx.x], sh) : 0.0;
  if (threadIdx.x == 0) hn = sqrt(npart ? ss : nsq[0]);   // nsq: |w|^2 summed over the ranks by the reduction callback
  __syncthreads();
  const double hnext = hn;
  if (scale) {
    const double f = (hnext > 0.0 && hnext <= 1.7976931348623157e308) ? 1.0 / hnext : 0.0;
    for (long i = blockIdx.x * (long)RT + threadIdx.x; i < n; i += (long)gridDim.x * RT) w[i] *= f;
  }
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double *hc = H + (long)j * (m + 1);
  for (int i = 0; i <= j; i++) hc[i] = hcol[i];
  hc[j + 1] = hnext;
  for (int i = 0; i < j; i++) {
    const double t = cs[i] * hc[i] + sn[i] * hc[i + 1];
    hc[i + 1] = -sn[i] * hc[i] + cs[i] * hc[i + 1]; hc[i] = t;
  }
  const double den = hypot(hc[j], hc[j + 1]);
  const double *g0 = G + (long)j * (m + 2);
  double *g1 = G + (long)(j + 1) * (m + 2);
  for (int i = 0; i < j; i++) g1[i] = g0[i];
  if (den == 0.0 || !(den == den)) { res[j] = nan(""); return; }
  const double c = hc[j] / den, s_ = hc[j + 1] / den;
  cs[j] = c; sn[j] = s_;
  hc[j] = den; hc[j + 1] = 0.0;
  g1[j] = c * g0[j]; g1[j + 1] = -s_ * g0[j];
  res[j] = fabs(g1[j + 1]);
}

// ---- Gram-Schmidt step with one reduction (round 5) -------------------------------------------------------------------------------
// The Givens part of a step: column j of H (entries hcol[0..j] and hnext below the diagonal), the previous rotations, the new one, the
// rotated right-hand side G_{j+1} (a fresh copy: iterations issued speculatively beyond the converged one must not disturb what the
// solve reads), the residual estimate for the host (pinned memory).
__device__ void givens_column(int j, int m, const double *hcol, double hnext, double *H, double *cs, double *sn, double *G, double *res) {
  double *hc = H + (long)j * (m + 1);
  for (int i = 0; i <= j; i++) hc[i] = hcol[i];
  hc[j + 1] = hnext;
  for (int i = 0; i < j; i++) {
    const double t = cs[i] * hc[i] + sn[i] * hc[i + 1];
    hc[i + 1] = -sn[i] * hc[i] + cs[i] * hc[i + 1]; hc[i] = t;
  }
  const double den = hypot(hc[j], hc[j + 1]);
  const double *g0 = G + (long)j * (m + 2);
  double *g1 = G + (long)(j + 1) * (m + 2);
  for (int i = 0; i < j; i++) g1[i] = g0[i];
  if (den == 0.0 || !(den == den)) { res[j] = nan(""); return; }
  const double c = hc[j] / den, s_ = hc[j + 1] / den;
  cs[j] = c; sn[j] = s_;
  hc[j] = den; hc[j + 1] = 0.0;
  g1[j] = c * g0[j]; g1[j + 1] = -s_ * g0[j];
  res[j] = fabs(g1[j + 1]);
}
// One pass gives d[kk] = V[kk].w and w.w.  The stored basis vectors are only APPROXIMATELY normalised: V[kk] has the exactly known norm
// rn[kk] (rn[0] = 1), the orthonormal basis of the method is V[kk] / rn[kk].  So
//   h[kk] = d[kk] / rn[kk]                 (column j of H),        c[kk] = h[kk] / rn[kk]   (w - sum c[kk] V[kk] is the new direction),
//   e2 = w.w - |h|^2 ~ h_{j+1,j}^2         (exact for an orthonormal basis; in floating point it loses the digits by which it is smaller than w.w)
// and e2 is used for ONE thing: the factor f = 1 / sqrt(e2) by which the update pass scales the new vector so that its norm is about 1.  The
// update pass then sums the squares of what it stores: rn[j+1] exactly, and h_{j+1,j} = rn[j+1] / f exactly -- a cancelled e2 costs nothing
// but a basis vector whose stored norm is not close to 1 (any positive f gives the same orthonormal direction).  scal[1] = f.
__device__ void gs_coefficients(int k, double *hcol, double *coef, const double *rn, double *scal) {
  const double ww = hcol[k];
  double ssq = 0.0;
  for (int i = 0; i < k; i++) { const double h = hcol[i] / rn[i]; hcol[i] = h; coef[i] = h / rn[i]; ssq += h * h; }
  double e2 = ww - ssq;
  if (!(e2 >= 1.0e-12 * ww)) e2 = 1.0e-12 * ww;            // cancelled (or negative): any scale of the right order will do
  scal[0] = ww;
  scal[1] = (e2 > 0.0 && e2 <= 1.7976931348623157e308) ? 1.0 / sqrt(e2) : 0.0;      // w = 0 (breakdown) or NaN: scale by 0 as before
}
// ... and after the update pass: rn[j+1] and h_{j+1,j} from the squares of the stored vector (nsq), then the Givens step.
// est != 0: no update pass was run (the last iteration of a truncated solve: its vector is not used) -- h_{j+1,j} is the estimate sqrt(e2).
__device__ void gs_norm_and_givens(int j, int m, const double *hcol, const double *scal, double nsq, int est, double *rn,
                                   double *H, double *cs, double *sn, double *G, double *res) {
  const double f = scal[1];
  double hn;
  if (est) hn = f > 0.0 ? 1.0 / f : (scal[0] == 0.0 ? 0.0 : nan(""));
  else {
    const double r = sqrt(nsq);
    rn[j + 1] = (r > 0.0 && r <= 1.7976931348623157e308) ? r : 1.0;
    hn = f > 0.0 ? r / f : (scal[0] == 0.0 ? 0.0 : nan(""));
  }
  givens_column(j, m, hcol, hn, H, cs, sn, G, res);
}
// Block versions for the last workgroup of k_gs_dots / k_gs_update.  A serial finish is a chain of dependent global round trips (one per
// entry of the column: hcol, rn, cs, sn, G) on the critical path of every iteration; here every thread fetches one entry into LDS and
// thread 0 does the arithmetic out of LDS, in the same order as the serial versions (the same bits).  GSF = entries of a column + 2.
constexpr int GSF = RT + 4;
struct GsShared { double a[GSF], b[GSF], c[GSF], d[GSF]; };
__device__ void gs_coefficients_block(int k, double *hcol, double *coef, const double *rn, double *scal, GsShared &S) {
  for (int i = threadIdx.x; i <= k; i += blockDim.x) { S.a[i] = hcol[i]; if (i < k) S.b[i] = rn[i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += blockDim.x) { const double h = S.a[i] / S.b[i]; hcol[i] = h; coef[i] = h / S.b[i]; S.a[i] = h; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ww = S.a[k];
    double ssq = 0.0;
    for (int i = 0; i < k; i++) ssq += S.a[i] * S.a[i];
    double e2 = ww - ssq;
    if (!(e2 >= 1.0e-12 * ww)) e2 = 1.0e-12 * ww;
    scal[0] = ww;
    scal[1] = (e2 > 0.0 && e2 <= 1.7976931348623157e308) ? 1.0 / sqrt(e2) : 0.0;
    S.d[0] = ww; S.d[1] = scal[1];                         // (for a Givens step that follows in the same kernel)
  }
  __syncthreads();
}
// S.a[0..j] must hold column j of H before the rotations (h[kk]); ww / f in S.d[0..1] or read from scal
__device__ void gs_norm_and_givens_block(int j, int m, const double *hcol, bool col_in_lds, double ww, double f, double nsq, int est, double *rn,
                                         double *H, double *cs, double *sn, double *G, double *res, GsShared &S) {
  const double *g0 = G + (long)j * (m + 2);
  for (int i = threadIdx.x; i <= j; i += blockDim.x) {
    if (!col_in_lds) S.a[i] = hcol[i];
    S.c[i] = g0[i];
    if (i < j) { S.b[i] = cs[i]; S.d[i + 2] = sn[i]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double hn;
    if (est) hn = f > 0.0 ? 1.0 / f : (ww == 0.0 ? 0.0 : nan(""));
    else {
      const double r = sqrt(nsq);
      rn[j + 1] = (r > 0.0 && r <= 1.7976931348623157e308) ? r : 1.0;
      hn = f > 0.0 ? r / f : (ww == 0.0 ? 0.0 : nan(""));
    }
    S.a[j + 1] = hn;
    for (int i = 0; i < j; i++) {
      const double t = S.b[i] * S.a[i] + S.d[i + 2] * S.a[i + 1];
      S.a[i + 1] = -S.d[i + 2] * S.a[i] + S.b[i] * S.a[i + 1]; S.a[i] = t;
    }
    const double den = hypot(S.a[j], S.a[j + 1]);
    if (den == 0.0 || !(den == den)) { res[j] = nan(""); S.d[0] = 0.0; }
    else {
      const double c = S.a[j] / den, s_ = S.a[j + 1] / den;
      cs[j] = c; sn[j] = s_;
      S.a[j] = den; S.a[j + 1] = 0.0;
      const double gj = S.c[j];
      S.c[j] = c * gj; S.c[j + 1] = -s_ * gj;
      res[j] = fabs(S.c[j + 1]);
      S.d[0] = 1.0;
    }
  }
  __syncthreads();
  double *hc = H + (long)j * (m + 1), *g1 = G + (long)(j + 1) * (m + 2);
  const bool ok = S.d[0] != 0.0;
  for (int i = threadIdx.x; i <= j + 1; i += blockDim.x) {
    hc[i] = S.a[i];
    if (i < j || ok) g1[i] = S.c[i];                       // (breakdown: as the serial version, only the first j entries are copied)
  }
}
// The LAST block to arrive (device-scope ticket: no block ever waits for another) sums the 256 partials of `rows` rows in a fixed order
// (wave q of 16 takes rows q, q + 16, ..; a lane adds its four partials in order, then the wave tree) into out[0..rows-1].
// Every partial was written before its writer's ticket, and every ticket before this block's.
//
// The premises of this fence-free form -- change any of them and the ticket must become an
// __ATOMIC_ACQ_REL one (with the L2 write-back it implies):
//   (1) every partial is stored by a lane of WAVE 0, the wave of the ticket-taking thread 0: s_waitcnt is a per-wave counter, so
//       thread 0's wait covers exactly the stores of its own wave (k_gs_dots: threads 0 .. cnt-1 <= KB-1; k_gs_update: thread 0);
//   (2) gfx942 / gfx950 count stores in vmcnt (gfx10+ has a separate vscnt, which s_waitcnt(0) would not drain), and their agent-scope
//       (sc1) stores write through to memory while agent-scope loads bypass the non-coherent caches.
static_assert(KB <= 64, "k_gs_dots: the partials of a block must all be stored by lanes of wave 0 (threads 0 .. KB-1), see gs_last_block");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "krylov.hip: the fence-free last-block ticket is valid on gfx942 / gfx950 only (stores counted in vmcnt): use an acq_rel ticket on other targets"
#endif
__device__ __forceinline__ bool gs_last_block(int *ticket, int nblocks, int *sh_last) {
  if (threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);                    // wave 0's agent-scope stores of the partials have been acknowledged (premise 1)
    *sh_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1;
  }
  __syncthreads();
  return *sh_last != 0;
}
__device__ __forceinline__ void gs_row_sums(double *part, int rows, double *out) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto ld = [&](long i) { return __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  for (int r = wv; r < rows; r += ST / 64) {
    double a = ld((long)r * RB + lane);
    a += ld((long)r * RB + 64 + lane); a += ld((long)r * RB + 128 + lane); a += ld((long)r * RB + 192 + lane);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if (lane == 0) out[r] = a;
  }
  __syncthreads();
}
// part[row][b] = sum over chunk b of V[row][i] * w[i] for row < k, and of w[i]^2 for row == k (the row after the basis rows); the last
// block leaves the sums in hcol[0..k]; with `finish` (one rank: nothing left to reduce) it goes on to gs_coefficients, and with
// finish == 2 (no update pass will follow) to the Givens step with the estimated h_{j+1,j}.
template <bool V2>
__global__ __launch_bounds__(ST) void k_gs_dots(long n, int k, const double *__restrict__ V, long ldv, const double *__restrict__ w,
                                                double *part, int *ticket, double *hcol, double *coef, double *rn, int finish,
                                                int j, int m, double *scal, double *H, double *cs, double *sn, double *G, double *res) {
  __shared__ double sh[ST / 64];
  __shared__ int last;
  const int kk0 = blockIdx.y * KB, rows = k + 1;
  const int cnt = rows - kk0 < KB ? rows - kk0 : KB;
  const double *v[KB];
#pragma unroll
  for (int q = 0; q < KB; q++) { const int r = kk0 + (q < cnt ? q : 0); v[q] = r < k ? V + (long)r * ldv : w; }
  double s[KB];
#pragma unroll
  for (int q = 0; q < KB; q++) s[q] = 0.0;
  if (V2) {
    double t[KB];
#pragma unroll
    for (int q = 0; q < KB; q++) t[q] = 0.0;
    const long n2 = n >> 1;
    for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n2; i += (long)RB * ST) {
      const double2 wi = ((const double2 *)w)[i];
#pragma unroll
      for (int q = 0; q < KB; q++) if (q < cnt) { const double2 a = ((const double2 *)v[q])[i]; s[q] += a.x * wi.x; t[q] += a.y * wi.y; }
    }
#pragma unroll
    for (int q = 0; q < KB; q++) s[q] += t[q];
  } else {
    for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n; i += (long)RB * ST) {
      const double wi = w[i];
#pragma unroll
      for (int q = 0; q < KB; q++) if (q < cnt) s[q] += v[q][i] * wi;
    }
  }
  // the KB block sums with ONE barrier (wave trees, the 16 wave sums of every row through LDS, row q finished by thread q in the
  // order block_sum_t uses: the same bits) instead of KB reductions of two barriers each -- on the 6-MB vectors of a 64^3 Schur
  // solve the kernel is all latency
  {
    __shared__ double shq[KB][ST / 64];
    const int lane_ = threadIdx.x & 63, w_ = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < KB; q++) {
      double v = s[q];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane_ == 0) shq[q][w_] = v;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)cnt) {
      double r = 0.0;
      for (int w2 = 0; w2 < ST / 64; w2++) r += shq[threadIdx.x][w2];
      __hip_atomic_store(part + (long)(kk0 + threadIdx.x) * RB + blockIdx.x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                      // thread 0 takes the ticket after every row's store has been issued ...
    (void)sh;
  }
  if (!gs_last_block(ticket, (int)(gridDim.x * gridDim.y), &last)) return;
  gs_row_sums(part, rows, hcol);
  if (threadIdx.x == 0) *ticket = 0;
  if (finish) {
    __shared__ GsShared S;
    gs_coefficients_block(k, hcol, coef, rn, scal, S);
    if (finish == 2) gs_norm_and_givens_block(j, m, hcol, true, S.d[0], S.d[1], 0.0, 1, rn, H, cs, sn, G, res, S);
  }
}
// several ranks: hcol[0..k] / nsq have been summed over the ranks by the reduction callback
__global__ void k_gs_coefficients(int k, double *hcol, double *coef, const double *rn, double *scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) gs_coefficients(k, hcol, coef, rn, scal);
}
__global__ void k_gs_givens(int j, int m, const double *hcol, const double *scal, const double *nsq, int est, double *rn,
                            double *H, double *cs, double *sn, double *G, double *res) {
  if (threadIdx.x == 0 && blockIdx.x == 0) gs_norm_and_givens(j, m, hcol, scal, nsq ? nsq[0] : 0.0, est, rn, H, cs, sn, G, res);
}
// w = (w - sum_{kk<k} c[kk] V[kk]) * f: the update and the (approximate) normalisation in one pass, the subtractions in row order;
// npart[b] = the squares of what block b stored; the last block sums them (fixed order) and, with `finish`, runs the Givens step;
// otherwise (several ranks) it leaves the local sum in nsq.
template <bool V2>
__global__ __launch_bounds__(ST) void k_gs_update(long n, int k, const double *__restrict__ V, long ldv, const double *__restrict__ coef,
                                                  const double *scal, double *__restrict__ w, double *npart, int *ticket, double *nsq, int finish,
                                                  int j, int m, const double *hcol, double *rn, double *H, double *cs, double *sn, double *G, double *res) {
  __shared__ double h[ST];
  __shared__ double sh[ST / 64];
  __shared__ int last;
  for (int kk = threadIdx.x; kk < k; kk += ST) h[kk] = coef[kk];
  __syncthreads();
  const double f = scal[1];
  double s = 0.0;
  if (V2) {
    const long n2 = n >> 1;
    double s2 = 0.0;
    for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n2; i += (long)RB * ST) {
      double2 x = ((const double2 *)w)[i];
      int kk = 0;
      for (; kk + 4 <= k; kk += 4) {
        const double2 a0 = ((const double2 *)(V + (long)kk * ldv))[i], a1 = ((const double2 *)(V + (long)(kk + 1) * ldv))[i],
                      a2 = ((const double2 *)(V + (long)(kk + 2) * ldv))[i], a3 = ((const double2 *)(V + (long)(kk + 3) * ldv))[i];
        x.x -= h[kk] * a0.x; x.y -= h[kk] * a0.y; x.x -= h[kk + 1] * a1.x; x.y -= h[kk + 1] * a1.y;
        x.x -= h[kk + 2] * a2.x; x.y -= h[kk + 2] * a2.y; x.x -= h[kk + 3] * a3.x; x.y -= h[kk + 3] * a3.y;
      }
      for (; kk < k; kk++) { const double2 a = ((const double2 *)(V + (long)kk * ldv))[i]; x.x -= h[kk] * a.x; x.y -= h[kk] * a.y; }
      x.x *= f; x.y *= f;
      ((double2 *)w)[i] = x;
      s += x.x * x.x; s2 += x.y * x.y;
    }
    s += s2;
  } else {
    for (long i = blockIdx.x * (long)ST + threadIdx.x; i < n; i += (long)RB * ST) {
      double x = w[i];
      int kk = 0;
      for (; kk + 4 <= k; kk += 4) {
        const double a0 = V[(long)kk * ldv + i], a1 = V[(long)(kk + 1) * ldv + i], a2 = V[(long)(kk + 2) * ldv + i], a3 = V[(long)(kk + 3) * ldv + i];
        x -= h[kk] * a0; x -= h[kk + 1] * a1; x -= h[kk + 2] * a2; x -= h[kk + 3] * a3;
      }
      for (; kk < k; kk++) x -= h[kk] * V[(long)kk * ldv + i];
      x *= f;
      w[i] = x;
      s += x * x;
    }
  }
  const double r = block_sum_t<ST>(s, sh);
  if (threadIdx.x == 0) __hip_atomic_store(npart + blockIdx.x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!gs_last_block(ticket, (int)gridDim.x, &last)) return;
  gs_row_sums(npart, 1, nsq);
  if (threadIdx.x == 0) *ticket = 0;
  if (finish) {
    __shared__ GsShared S;
    gs_norm_and_givens_block(j, m, hcol, false, scal[0], scal[1], nsq[0], 0, rn, H, cs, sn, G, res, S);
  }
}

// y = R^{-1} g for the leading kk columns (R upper triangular, column-major with leading dimension m + 1)
__global__ void k_trisolve(int kk, int m, const double *__restrict__ H, const double *__restrict__ g, double *__restrict__ y) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = kk - 1; i >= 0; i--) {
    double s = g[i];
    for (int q = i + 1; q < kk; q++) s -= H[(long)q * (m + 1) + i] * y[q];
    y[i] = s / H[(long)i * (m + 1) + i];
  }
}

__global__ void k_set1(double *__restrict__ p, double v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }

// y = a * x        (a read from the host value)
__global__ void k_scale(long n, double a, const double *x, double *y) {   // x == y allowed
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = a * x[i];
}

// x = (fresh ? 0 : x) + sum_{kk<k} y[kk] Z[kk]      (fresh: x has not been written yet -- a zero initial guess that is never
// stored: 0 + a is a, so the sum has the bits of the accumulate on a cleared x)
__global__ void k_multiaxpy(long n, int k, const double *__restrict__ Z, long ldz, const double *__restrict__ y, double *__restrict__ x, int fresh) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double s = fresh ? 0.0 : x[i];
    for (int kk = 0; kk < k; kk++) s += y[kk] * Z[(long)kk * ldz + i];
    x[i] = s;
  }
}

// r = b - r
__global__ void k_residual(long n, const double *__restrict__ b, double *__restrict__ r) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) r[i] = b[i] - r[i];
}

inline unsigned pgrid(long n) { long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

}  // namespace

struct chebhip_fgmres {
  long n = 0, ld = 0;
  int m = 30;                       // restart (KSPGMRESSetRestart default 30)
  double rtol = 1e-5, atol = 1e-50; // KSP defaults
  int max_it = 10000;
  double *V = nullptr, *Z = nullptr, *part = nullptr, *npart = nullptr, *hcol = nullptr, *ydev = nullptr;
  double *H = nullptr, *cs = nullptr, *sn = nullptr, *G = nullptr;
  double *res = nullptr;            // pinned, written by the device: residual estimate after iteration j
  std::vector<hipEvent_t> ev;
  chebhip_reduce_fn reduce = nullptr;   // several ranks: sums device doubles over the ranks, in place, stream-ordered
  void *reduce_ctx = nullptr;
  double *nsq = nullptr;                // device scalar for the reduced |w|^2
  int *ticket = nullptr;                // k_gs_dots: blocks that have delivered their partial sums
  double *scal = nullptr;               // k_gs_dots -> k_gs_update: h_{j+1,j} and its reciprocal
  double *coef = nullptr, *rn = nullptr; // k_gs_dots -> k_gs_update: update coefficients; exact norms of the stored basis vectors
  bool exact = false;                   // option "krylov_exact_norm": the three-launch step with the explicit norm
  int its = 0, reason = 0;
  double rnorm = 0.0, rnorm0 = 0.0;
};

extern "C" int chebhip_fgmres_destroy(chebhip_fgmres *k) {
  if (!k) return 0;
  double *dev[] = {k->V, k->Z, k->part, k->npart, k->hcol, k->ydev, k->H, k->cs, k->sn, k->G, k->nsq, k->scal, k->coef, k->rn};
  for (double *p : dev) if (p) (void)hipFree(p);
  if (k->ticket) (void)hipFree(k->ticket);
  if (k->res) (void)hipHostFree(k->res);
  for (hipEvent_t e : k->ev) (void)hipEventDestroy(e);
  delete k;
  return 0;
}

extern "C" int chebhip_fgmres_create(long n, int restart, chebhip_fgmres **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (n < 0) return chebhip_fail(CHEBHIP_ERR_SIZE, "n = %ld but must be >= 0", n);     // 0: a rank without unknowns
  if (restart < 1 || restart > RT) return chebhip_fail(CHEBHIP_ERR_ARG, "restart = %d must be in 1..%d", restart, RT);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return chebhip_fail(CHEBHIP_ERR_DEVICE, "no usable HIP device; libchebhip has no CPU fallback");
  chebhip_fgmres *k = new (std::nothrow) chebhip_fgmres;
  if (!k) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  const int m = restart;
  k->n = n; k->m = m; k->ld = n > 0 ? ((n + 1) & ~1L) : 2;             // even leading dimension: every basis vector 16-B aligned
#define KC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { chebhip_fgmres_destroy(k); \
    return chebhip_fail(e_ == hipErrorOutOfMemory ? CHEBHIP_ERR_MEMORY : CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
  KC(hipMalloc((void **)&k->V, (size_t)(m + 1) * k->ld * sizeof(double)));
  KC(hipMalloc((void **)&k->Z, (size_t)m * k->ld * sizeof(double)));
  KC(hipMalloc((void **)&k->part, (size_t)(m + 2) * RB * sizeof(double)));
  KC(hipMalloc((void **)&k->ticket, sizeof(int)));
  KC(hipMemset(k->ticket, 0, sizeof(int)));
  KC(hipMalloc((void **)&k->scal, 4 * sizeof(double)));
  KC(hipMalloc((void **)&k->coef, (size_t)(m + 3) * sizeof(double)));
  KC(hipMalloc((void **)&k->rn, (size_t)(m + 3) * sizeof(double)));
  KC(hipMalloc((void **)&k->npart, (size_t)RB * sizeof(double)));
  KC(hipMalloc((void **)&k->hcol, (size_t)(m + 3) * sizeof(double)));
  KC(hipMalloc((void **)&k->ydev, (size_t)(m + 2) * sizeof(double)));
  KC(hipMalloc((void **)&k->H, (size_t)m * (m + 1) * sizeof(double)));
  KC(hipMalloc((void **)&k->cs, (size_t)m * sizeof(double)));
  KC(hipMalloc((void **)&k->sn, (size_t)m * sizeof(double)));
  KC(hipMalloc((void **)&k->G, (size_t)(m + 1) * (m + 2) * sizeof(double)));
  KC(hipMalloc((void **)&k->nsq, sizeof(double)));
  KC(hipHostMalloc((void **)&k->res, (size_t)(m + 2) * sizeof(double)));
  k->ev.assign(m, nullptr);
  for (int j = 0; j < m; j++) KC(hipEventCreateWithFlags(&k->ev[j], hipEventDisableTiming));
  KC(hipStreamSynchronize(nullptr));      // the ticket was cleared on the null stream, which a caller's non-blocking stream does not wait for
#undef KC
  *out = k;
  return 0;
}

extern "C" int chebhip_fgmres_set_tolerances(chebhip_fgmres *k, double rtol, double atol, int max_it) {
  if (!k) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL handle");
  if (!(rtol >= 0.0) || !(atol >= 0.0) || max_it < 0) return chebhip_fail(CHEBHIP_ERR_ARG, "negative tolerance or iteration limit");
  k->rtol = rtol; k->atol = atol; k->max_it = max_it;
  return 0;
}

// Vectors distributed over several ranks (each holds n local entries): every inner product is completed by
// `reduce`, which must sum `count` device doubles over the ranks in place, ordered on the given stream
// (ncclAllReduce).  NULL restores the single-rank behaviour.
extern "C" int chebhip_fgmres_set_reduce(chebhip_fgmres *k, chebhip_reduce_fn reduce, void *ctx) {
  if (!k) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL handle");
  k->reduce = reduce; k->reduce_ctx = ctx;
  return 0;
}

extern "C" int chebhip_fgmres_iterations(const chebhip_fgmres *k) { return k ? k->its : -1; }
extern "C" double chebhip_fgmres_residual(const chebhip_fgmres *k) { return k ? k->rnorm : -1.0; }
extern "C" int chebhip_fgmres_reason(const chebhip_fgmres *k) { return k ? k->reason : 0; }

// |v| on the device, result to the host (synchronises the stream)
static int dev_norm(chebhip_fgmres *k, const double *v, hipStream_t st, double *out) {
  hipLaunchKernelGGL(k_multidot, dim3(RB, 1), dim3(RT), 0, st, k->n, v, k->ld, v, k->npart);
  if (!k->reduce) hipLaunchKernelGGL(k_rows_finish, dim3(1), dim3(RT), 0, st, (const double *)k->npart, k->res + k->m, 1);
  else {
    hipLaunchKernelGGL(k_rows_finish, dim3(1), dim3(RT), 0, st, (const double *)k->npart, k->nsq, 0);
    int rc = k->reduce(k->reduce_ctx, k->nsq, 1, st); if (rc) return rc;
    hipLaunchKernelGGL(k_sqrt1, dim3(1), dim3(1), 0, st, k->nsq);
    KHIPCHK(hipMemcpyAsync(k->res + k->m, k->nsq, sizeof(double), hipMemcpyDeviceToHost, st));
  }
  KHIPCHK(hipStreamSynchronize(st));
  *out = k->res[k->m];
  return 0;
}

extern "C" int chebhip_fgmres_solve(chebhip_fgmres *k, chebhip_apply_fn A, void *actx, chebhip_apply_fn M, void *mctx,
                                    const double *b, double *x, int x_nonzero, void *stream) {
  if (!k || !A || ((!b || !x) && k->n > 0)) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  chebhip::StageTimer tm(CHEBHIP_STAGE_FGMRES_SOLVE, stream);
  hipStream_t st = (hipStream_t)stream;
  const long n = k->n, ld = k->ld;
  const int m = k->m;
  k->its = 0; k->reason = 0;
  k->exact = chebhip::opt(chebhip::OPT_KRYLOV_EXACT_NORM) != 0;
  double bnorm = 0.0;
  int rc = dev_norm(k, b, st, &bnorm); if (rc) return rc;
  const double tol = std::fmax(k->rtol * bnorm, k->atol);
  // A zero initial guess is not written: x stays untouched until the first update stores it (k_multiaxpy, fresh), or until the
  // solve ends without one -- clear_x on those paths.  One pass over x less per solve; the inner solves of the Stokes
  // preconditioners (stokes.C:328-341: 1..4 iterations each) are a few dozen vector passes in all.
  bool x_unset = !x_nonzero;
  auto clear_x = [&]() -> int {
    if (x_unset && n > 0) KHIPCHK(hipMemsetAsync(x, 0, (size_t)n * sizeof(double), st));
    x_unset = false; return 0;
  };
  bool first = true;
  for (;;) {
    // r = b - A x  into V[0]
    double *r = k->V;
    double beta = 0.0;
    const double *rsrc = r;
    if (first && !x_nonzero) { beta = bnorm; rsrc = b; }      // r = b: its norm is known, and V[0] = b / |b| is formed from b itself
    else {
      if ((rc = A(actx, x, r, st))) return rc;
      hipLaunchKernelGGL(k_residual, dim3(pgrid(n)), dim3(256), 0, st, n, b, r);
      if ((rc = dev_norm(k, r, st, &beta))) return rc;
    }
    if (first) k->rnorm0 = beta;
    first = false;
    k->rnorm = beta;
    if (!(beta == beta)) { k->reason = -9; return clear_x(); }                  // NaN: KSP_DIVERGED_NANORINF
    if (beta <= tol) { k->reason = beta <= k->atol ? 3 : 2; return clear_x(); } // KSP_CONVERGED_ATOL / RTOL
    if (k->its >= k->max_it) { k->reason = -3; return clear_x(); }              // KSP_DIVERGED_ITS
    hipLaunchKernelGGL(k_scale, dim3(pgrid(n)), dim3(256), 0, st, n, 1.0 / beta, rsrc, k->V);
    hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, st, k->G, beta);
    hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, st, k->rn, 1.0);          // V[0] is normalised exactly

    // One cycle.  Iteration j is enqueued BEFORE the host looks at the result of iteration j - 1: the device
    // never waits for the host.  If j - 1 turns out to have converged, iteration j was speculative; it only
    // wrote column j of H, G_{j+1} and V[j+1], none of which the update below reads.
    int kk = 0;            // accepted columns
    bool stop = false;
    auto examine = [&](int jj) {                                         // result of iteration jj (stream reached ev[jj])
      const double rj = k->res[jj];
      if (!(rj == rj)) { k->reason = -9; kk = jj; stop = true; return; } // breakdown: column jj is dropped
      kk = jj + 1; k->rnorm = rj;
      if (rj <= tol || k->its + kk >= k->max_it) stop = true;
    };
    int enq = 0;
    for (int j = 0; j < m && !stop; j++) {
      if (k->its + j >= k->max_it) break;
      const double *vj = k->V + (long)j * ld;
      const double *zj = vj;
      if (M) { double *z = k->Z + (long)j * ld; if ((rc = M(mctx, vj, z, st))) return rc; zj = z; }
      double *w = k->V + (long)(j + 1) * ld;
      // V[j+1] is a basis vector only if another iteration of this cycle can follow: not at the end of the cycle, not at the
      // iteration limit (the inner velocity solves stop there every time) -- then its normalisation pass is not run
      const int use_next = (j + 1 < m && k->its + j + 1 < k->max_it) ? 1 : 0;
      const unsigned gsgrid = use_next ? pgrid(n) : 1u;
      if ((rc = A(actx, zj, w, st))) return rc;
      if (!k->exact) {
        // V^T w and w.w in one pass; the update and the normalisation in one pass (see gs_coefficients)
        const int single = k->reduce ? 0 : 1;
        const dim3 dg(RB, (j + 1 + KB) / KB);
        const int fin = single ? (use_next ? 1 : 2) : 0;
        if ((n & 1) == 0) hipLaunchKernelGGL((k_gs_dots<true>), dg, dim3(ST), 0, st, n, j + 1, (const double *)k->V, ld, (const double *)w, k->part, k->ticket, k->hcol, k->coef, k->rn, fin,
                                             j, m, k->scal, k->H, k->cs, k->sn, k->G, k->res);
        else hipLaunchKernelGGL((k_gs_dots<false>), dg, dim3(ST), 0, st, n, j + 1, (const double *)k->V, ld, (const double *)w, k->part, k->ticket, k->hcol, k->coef, k->rn, fin,
                                j, m, k->scal, k->H, k->cs, k->sn, k->G, k->res);
        if (!single) {
          if ((rc = k->reduce(k->reduce_ctx, k->hcol, j + 2, st))) return rc;
          hipLaunchKernelGGL(k_gs_coefficients, dim3(1), dim3(1), 0, st, j + 1, k->hcol, k->coef, (const double *)k->rn, k->scal);
          if (!use_next) hipLaunchKernelGGL(k_gs_givens, dim3(1), dim3(1), 0, st, j, m, (const double *)k->hcol, (const double *)k->scal, (const double *)nullptr, 1, k->rn,
                                            k->H, k->cs, k->sn, k->G, k->res);
        }
        if (use_next) {             // (the last iteration of a truncated solve needs h_{j+1,j} only: no pass over the basis at all)
          if ((n & 1) == 0) hipLaunchKernelGGL((k_gs_update<true>), dim3(RB), dim3(ST), 0, st, n, j + 1, (const double *)k->V, ld, (const double *)k->coef, (const double *)k->scal, w,
                                               k->npart, k->ticket, k->nsq, single, j, m, (const double *)k->hcol, k->rn, k->H, k->cs, k->sn, k->G, k->res);
          else hipLaunchKernelGGL((k_gs_update<false>), dim3(RB), dim3(ST), 0, st, n, j + 1, (const double *)k->V, ld, (const double *)k->coef, (const double *)k->scal, w,
                                  k->npart, k->ticket, k->nsq, single, j, m, (const double *)k->hcol, k->rn, k->H, k->cs, k->sn, k->G, k->res);
          if (!single) {
            if ((rc = k->reduce(k->reduce_ctx, k->nsq, 1, st))) return rc;
            hipLaunchKernelGGL(k_gs_givens, dim3(1), dim3(1), 0, st, j, m, (const double *)k->hcol, (const double *)k->scal, (const double *)k->nsq, 0, k->rn,
                               k->H, k->cs, k->sn, k->G, k->res);
          }
        }
        KHIPCHK(hipGetLastError());
        KHIPCHK(hipEventRecord(k->ev[j], st));
        enq = j + 1;
        if (j >= 1) { KHIPCHK(hipEventSynchronize(k->ev[j - 1])); examine(j - 1); }
        continue;
      }
      if ((n & 1) == 0) hipLaunchKernelGGL((k_multidot_grouped<true>), dim3(RB, (j + KB) / KB), dim3(ST), 0, st, n, j + 1, (const double *)k->V, ld, (const double *)w, k->part);
      else hipLaunchKernelGGL((k_multidot_grouped<false>), dim3(RB, (j + KB) / KB), dim3(ST), 0, st, n, j + 1, (const double *)k->V, ld, (const double *)w, k->part);
      if (!k->reduce) {
        hipLaunchKernelGGL(k_orth_update, dim3(RB), dim3(ST), 0, st, n, j + 1, (const double *)k->V, ld, (const double *)k->part, k->hcol, w, k->npart, use_next);
        hipLaunchKernelGGL(k_givens_scale, dim3(gsgrid), dim3(RT), 0, st, j, m, (const double *)k->npart, (const double *)k->hcol, k->H, k->cs, k->sn, k->G, k->res,
                           (const double *)nullptr, n, w, use_next);
      } else {            // several ranks: local sums -> all-reduce -> update; the same for |w|^2
        hipLaunchKernelGGL(k_rows_finish, dim3(j + 1), dim3(RT), 0, st, (const double *)k->part, k->hcol, 0);
        if ((rc = k->reduce(k->reduce_ctx, k->hcol, j + 1, st))) return rc;
        hipLaunchKernelGGL(k_orth_update, dim3(RB), dim3(ST), 0, st, n, j + 1, (const double *)k->V, ld, (const double *)nullptr, k->hcol, w, k->npart, use_next);
        hipLaunchKernelGGL(k_rows_finish, dim3(1), dim3(RT), 0, st, (const double *)k->npart, k->nsq, 0);
        if ((rc = k->reduce(k->reduce_ctx, k->nsq, 1, st))) return rc;
        hipLaunchKernelGGL(k_givens_scale, dim3(gsgrid), dim3(RT), 0, st, j, m, (const double *)nullptr, (const double *)k->hcol, k->H, k->cs, k->sn, k->G, k->res,
                           (const double *)k->nsq, n, w, use_next);
      }
      KHIPCHK(hipEventRecord(k->ev[j], st));
      enq = j + 1;
      if (j >= 1) { KHIPCHK(hipEventSynchronize(k->ev[j - 1])); examine(j - 1); }
    }
    if (!stop && enq > 0) { KHIPCHK(hipEventSynchronize(k->ev[enq - 1])); examine(enq - 1); }
    // was the last accepted column one whose h_{j+1,j} is the one-pass estimate (no update pass: `use_next` = 0 above)?
    const bool last_est = !k->exact && kk > 0 && !(kk < m && k->its + kk < k->max_it);
    k->its += kk;
    // y = R^{-1} g (leading kk columns, right-hand side G_kk), x += Z y
    if (kk > 0) {
      hipLaunchKernelGGL(k_trisolve, dim3(1), dim3(1), 0, st, kk, m, (const double *)k->H, (const double *)(k->G + (long)kk * (m + 2)), k->ydev);
      hipLaunchKernelGGL(k_multiaxpy, dim3(pgrid(n)), dim3(256), 0, st, n, kk, (const double *)(M ? k->Z : k->V), ld, (const double *)k->ydev, x, x_unset ? 1 : 0);
      x_unset = false;
    }
    KHIPCHK(hipStreamSynchronize(st));      // a speculative iteration may still be running: drain before V is reused
    if (k->reason == -9) return clear_x();
    // convergence read off an estimated column is checked on the true residual (top of the loop: converged, iteration limit, or on)
    if (k->rnorm <= tol && !last_est) { k->reason = k->rnorm <= k->atol ? 3 : 2; return clear_x(); }
    if (k->its >= k->max_it && !(k->rnorm <= tol)) { k->reason = -3; return clear_x(); }
    // otherwise restart: the true residual is recomputed at the top
  }
}
