// stokes.hip -- the Stokes operator callbacks (stokes.C:499-758) over the sweep kernels:
// StokesMatMult / VV / PV / VP / Schur and StokesFunction with the linear and power-law rheologies,
// -boundary 0 (all-Dirichlet velocity).  The ChebMults of the reference (DV[i]: rank d+1 tensor with the d
// components innermost, stokes.C:284-290; DP[i]: scalar rank d) run as multi-job sweep launches on
// component-major work vectors, the vector passes between them as the node-local kernels below.  Routes
// (DESIGN.md 4.3): the general viscous block -- gather, x / y gradient launch, k_st_pfaces, k_st_zfused16 (z
// gradient + node loop + z divergence in one launch, the pressure subtracted from the diagonal stress), x / y
// divergence launch, scatter -- on 3-D grids with 68..128-point contiguous lines; separate passes elsewhere;
// -eta/2 (sum D_j D_j v + grad div v) without a node loop for a uniform viscosity; slab-mode handles whose
// dimension-0 work goes through the driver's callback (slabx.hip).
#include "../../include/chebhip.h"
#include "sweep.h"
#include "ops.h"
#include "timers.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace chebhip;

#define SHIPCHK(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

static inline unsigned sgrid(long n) { long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
static inline unsigned pgrid(long nlines) { long g = (nlines + 31) / 32; return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }
#define GS_LOOP(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// Work-vector layout.  The reference keeps the d components of a node interleaved (rank d+1 plans with the
// components innermost, stokes.C:284-290); the global vectors at the ABI keep that layout.  Inside the
// operator the component index is OUTERMOST (field k occupies [k*N, (k+1)*N)): DV[i] is then DP[i] over d
// stacked scalar fields -- contiguous lines for the last grid dimension, 16-byte accesses everywhere -- and a
// component is a contiguous scalar field (no VecStrideGather/Scatter copies, stokes.C:585,613).
//
// xL <- velocity part, pL <- pressure part of a global vector (node stride gs, pressure offset go); zero or
// Dirichlet values on the boundary: VecZeroEntries + scatterGV/VL (+ scatterDL) + scatterGP,
// stokes.C:505-510,575-582,605-608,634-637,695-699.  xL or pL may be null.
// cs: stride between the components of a node in src -- 1 for the reference's node-major vectors, I (with gs = 1) for the
// component-major velocity vectors of the block preconditioners' inner solves (saddle.hip)
template <int D>
__global__ void k_st_local(long N, int gs, int go, const int *__restrict__ ixL, const double *__restrict__ src,
                           const double *__restrict__ dirloc, double *__restrict__ xL, double *__restrict__ pL, long cs) {
  GS_LOOP(l, N) {
    const int n = ixL[l];
    const double *s = src + (long)(n >= 0 ? n : 0) * gs;
    if (xL) {
#pragma unroll
      for (int k = 0; k < D; k++) xL[k * N + l] = n >= 0 ? s[k * cs] : (dirloc ? dirloc[k * N + l] : 0.0);
    }
    if (pL) pL[l] = n >= 0 ? s[go] : 0.0;
  }
}
// The same for the full 3-D vector (node stride 4, pressure last, 16-byte aligned): the 32 bytes of a node come as two
// 16-byte loads, and a thread keeps UN nodes in flight (index loads first, then the node loads, then the stores) -- the
// one-node version is a chain of two dependent memory round trips per node and ran at 3.7 TB/s at 128^3.
template <int UN>
__global__ __launch_bounds__(256) void k_st_local4(long N, const int *__restrict__ ixL, const double *__restrict__ src,
                                                   const double *__restrict__ dirloc, double *__restrict__ xL, double *__restrict__ pL) {
  const long T = (long)gridDim.x * blockDim.x;
  for (long l0 = blockIdx.x * (long)blockDim.x + threadIdx.x; l0 < N; l0 += UN * T) {
    int n[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) { const long l = l0 + u * T; n[u] = l < N ? ixL[l] : -1; }
    double2 a[UN], b[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const double2 *s2 = (const double2 *)(src + 4L * (n[u] >= 0 ? n[u] : 0));
      a[u] = s2[0]; b[u] = s2[1];
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long l = l0 + u * T;
      if (l >= N) continue;
      const bool in = n[u] >= 0;
      xL[l] = in ? a[u].x : (dirloc ? dirloc[l] : 0.0);
      xL[N + l] = in ? a[u].y : (dirloc ? dirloc[N + l] : 0.0);
      xL[2 * N + l] = in ? b[u].x : (dirloc ? dirloc[2 * N + l] : 0.0);
      pL[l] = in ? b[u].y : 0.0;
    }
  }
}

// Interior indices of the node pair (2t, 2t + 1) of a serial 3-D grid with an even last extent, by arithmetic instead of from the table
// ixL (4 B/node of every gather and scatter: 1.2 % of the bytes of a 128^3 callback): n(i, j, k) = ((i-1) M1 + (j-1)) M2 + (k-1) for
// interior nodes (BlockIt order, stokes.C:791-879), -1 on the boundary.  P2 == 0: read the table (slab handles, odd extents, d = 2).
struct StGrid { unsigned P0, P1, P2; };
__device__ __forceinline__ int2 st_pair_ix(const int *__restrict__ ixL, long t, const StGrid g) {
  if (g.P2 == 0) return ((const int2 *)ixL)[t];
  const unsigned l = (unsigned)(2 * t), q = l / g.P2, k = l - q * g.P2;          // k even: the pair shares its line
  const unsigned i = q / g.P1, j = q - i * g.P1;
  if (i == 0 || i == g.P0 - 1 || j == 0 || j == g.P1 - 1) return make_int2(-1, -1);
  const int base = (int)(((i - 1) * (g.P1 - 2) + (j - 1)) * (g.P2 - 2)) - 1;
  return make_int2(k == 0 ? -1 : base + (int)k, k + 2 == g.P2 ? -1 : base + (int)k + 1);
}

// ... and on node PAIRS (N even): the two nodes (l, l+1) of a pair are neighbours in every local field, so each field
// leaves as one 16-byte store per pair instead of two 8-byte ones (a CU's store path moves 8-byte accesses at ~0.6x the
// rate of 16-byte ones, sweep_vec.hip); UN pairs in flight per thread.
template <int UN>
__global__ __launch_bounds__(256) void k_st_local4p(long N, const int *__restrict__ ixL, const double *__restrict__ src,
                                                    const double *__restrict__ dirloc, double *__restrict__ xL, double *__restrict__ pL, const StGrid sg) {
  const long T = (long)gridDim.x * blockDim.x, half = N >> 1;
  for (long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x; t0 < half; t0 += UN * T) {
    int2 n[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) { const long t = t0 + u * T; n[u] = t < half ? st_pair_ix(ixL, t, sg) : make_int2(-1, -1); }
    double2 a[UN][2], b[UN][2];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const double2 *s0 = (const double2 *)(src + 4L * (n[u].x >= 0 ? n[u].x : 0)), *s1 = (const double2 *)(src + 4L * (n[u].y >= 0 ? n[u].y : 0));
      a[u][0] = s0[0]; b[u][0] = s0[1]; a[u][1] = s1[0]; b[u][1] = s1[1];
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long t = t0 + u * T;
      if (t >= half) continue;
      const bool i0 = n[u].x >= 0, i1 = n[u].y >= 0;
      double2 d0 = make_double2(0.0, 0.0), d1 = d0, d2v = d0;
      if (dirloc && !(i0 && i1)) { d0 = ((const double2 *)dirloc)[t]; d1 = ((const double2 *)(dirloc + N))[t]; d2v = ((const double2 *)(dirloc + 2 * N))[t]; }
      ((double2 *)xL)[t] = make_double2(i0 ? a[u][0].x : d0.x, i1 ? a[u][1].x : d0.y);
      ((double2 *)(xL + N))[t] = make_double2(i0 ? a[u][0].y : d1.x, i1 ? a[u][1].y : d1.y);
      ((double2 *)(xL + 2 * N))[t] = make_double2(i0 ? b[u][0].x : d2v.x, i1 ? b[u][1].x : d2v.y);
      ((double2 *)pL)[t] = make_double2(i0 ? b[u][0].y : 0.0, i1 ? b[u][1].y : 0.0);
    }
  }
}

// Boundary pressure of one family of grid lines (StokesPressureReduceOrder, stokes.C:1029-1080): the two
// end values of a line become the degree-(len-3) polynomial through its interior values evaluated at the
// ends.  The reference builds a Neville table per line (util.C:129-144, O(len^2)); the same linear
// functional is applied here as two dot products with precomputed Lagrange weights.
// 256 threads = 8 parts x 32 lines: neighbouring threads take neighbouring lines (coalesced for the y and x
// families, whose lines are strided), each part sums a fixed slice of the line and the 8 partial sums are
// added in a fixed order through LDS -- deterministic, and 8x the parallelism of one thread per line.
__global__ __launch_bounds__(256) void k_st_preduce(double *__restrict__ pres, long na, long a0, long sa, long nb, long b0,
                                                    long sb, long se, int len, const double *__restrict__ w0,
                                                    const double *__restrict__ w1) {
  __shared__ double s0[8][32], s1[8][32];
  const int tb = threadIdx.x & 31, part = threadIdx.x >> 5;
  const long nlines = na * nb;
  const int m = len - 2, chunk = (m + 7) / 8;
  for (long g = blockIdx.x; g * 32 < nlines; g += gridDim.x) {
    const long t = g * 32 + tb;
    double f0 = 0.0, f1 = 0.0;
    double *line = nullptr;
    if (t < nlines) {
      const long a = t / nb, b = t - a * nb;
      line = pres + (a + a0) * sa + (b + b0) * sb;
      const int j0 = part * chunk, j1 = (j0 + chunk < m) ? j0 + chunk : m;
      for (int j = j0; j < j1; j++) { const double v = line[(long)(j + 1) * se]; f0 += w0[j] * v; f1 += w1[j] * v; }
    }
    s0[part][tb] = f0; s1[part][tb] = f1;
    __syncthreads();
    if (part == 0 && t < nlines) {
      double r0 = 0.0, r1 = 0.0;
#pragma unroll
      for (int q = 0; q < 8; q++) { r0 += s0[q][tb]; r1 += s1[q][tb]; }
      line[0] = r0; line[(long)(len - 1) * se] = r1;
    }
    __syncthreads();
  }
}

// The same for lines that are contiguous in memory (the last grid dimension): one wave per line, lanes along
// the line (coalesced), the 64 partial sums combined by a fixed shuffle tree -- deterministic.
__global__ __launch_bounds__(256) void k_st_preduce_contig(double *__restrict__ pres, long na, long a0, long sa, long nb, long b0,
                                                           long sb, int len, const double *__restrict__ w0,
                                                           const double *__restrict__ w1) {
  const int lane = threadIdx.x & 63;
  const long nlines = na * nb;
  const int m = len - 2;
  for (long t = blockIdx.x * 4L + (threadIdx.x >> 6); t < nlines; t += (long)gridDim.x * 4) {
    const long a = t / nb, b = t - a * nb;
    double *line = pres + (a + a0) * sa + (b + b0) * sb;
    double f0 = 0.0, f1 = 0.0;
    for (int j = lane; j < m; j += 64) { const double v = line[j + 1]; f0 += w0[j] * v; f1 += w1[j] * v; }
    for (int o = 32; o > 0; o >>= 1) { f0 += __shfl_down(f0, o, 64); f1 += __shfl_down(f1, o, 64); }
    if (lane == 0) { line[0] = f0; line[len - 1] = f1; }
  }
}

// The end values of every INTERIOR grid line of a 3-D pressure field along each of the three directions -- the face-interior nodes -- by
// extrapolation along the line, in ONE launch (blockIdx.y = direction; the three families write disjoint nodes, edges and corners
// stay as they are: they only end lines that lie in the boundary, whose results nobody reads).  This is what the fused pressure
// route needs of StokesPressureReduceOrder (st_viscous_jacobian_zfused): -D_k tau_kk + DP_k p = -D_k (tau_kk - p_ext).
// z lines: one wave per line, lanes along it; x and y lines: 32 neighbouring lines per block, 8 parts each, fixed-order sums.
__global__ __launch_bounds__(256) void k_st_pfaces(double *__restrict__ pres, int P0, int P1, int P2,
                                                   const double *__restrict__ w0x, const double *__restrict__ w1x,
                                                   const double *__restrict__ w0y, const double *__restrict__ w1y,
                                                   const double *__restrict__ w0z, const double *__restrict__ w1z) {
  const int dir = blockIdx.y;
  if (dir == 2) {
    const int lane = threadIdx.x & 63, m = P2 - 2;
    const long nb = P1 - 2, nlines = (long)(P0 - 2) * nb;
    for (long t = blockIdx.x * 4L + (threadIdx.x >> 6); t < nlines; t += (long)gridDim.x * 4) {
      const long a = t / nb, b = t - a * nb;
      double *line = pres + ((a + 1) * P1 + (b + 1)) * (long)P2;
      double f0 = 0.0, f1 = 0.0;
      for (int j = lane; j < m; j += 64) { const double v = line[j + 1]; f0 += w0z[j] * v; f1 += w1z[j] * v; }
      for (int o = 32; o > 0; o >>= 1) { f0 += __shfl_down(f0, o, 64); f1 += __shfl_down(f1, o, 64); }
      if (lane == 0) { line[0] = f0; line[P2 - 1] = f1; }
    }
    return;
  }
  __shared__ double s0[8][32], s1[8][32];
  const int tb = threadIdx.x & 31, part = threadIdx.x >> 5;
  // dir 1: lines (a, :, b), a in 1..P0-2, b in 1..P2-2, element stride P2;  dir 0: lines (:, a, b), a in 1..P1-2, b in 1..P2-2, stride P1 P2
  const long nb = P2 - 2, na = dir == 1 ? P0 - 2 : P1 - 2, nlines = na * nb;
  const long sa = dir == 1 ? (long)P1 * P2 : (long)P2, se = dir == 1 ? (long)P2 : (long)P1 * P2;
  const int len = dir == 1 ? P1 : P0, m = len - 2, chunk = (m + 7) / 8;
  const double *w0 = dir == 1 ? w0y : w0x, *w1 = dir == 1 ? w1y : w1x;
  for (long g = blockIdx.x; g * 32 < nlines; g += gridDim.x) {
    const long t = g * 32 + tb;
    double f0 = 0.0, f1 = 0.0;
    double *line = nullptr;
    if (t < nlines) {
      const long a = t / nb, b = t - a * nb;
      line = pres + (a + 1) * sa + (b + 1);
      const int j0 = part * chunk, j1 = (j0 + chunk < m) ? j0 + chunk : m;
      for (int j = j0; j < j1; j++) { const double v = line[(long)(j + 1) * se]; f0 += w0[j] * v; f1 += w1[j] * v; }
    }
    s0[part][tb] = f0; s1[part][tb] = f1;
    __syncthreads();
    if (part == 0 && t < nlines) {
      double r0 = 0.0, r1 = 0.0;
#pragma unroll
      for (int q = 0; q < 8; q++) { r0 += s0[q][tb]; r1 += s1[q][tb]; }
      line[0] = r0; line[(long)(len - 1) * se] = r1;
    }
    __syncthreads();
  }
}

// Node loop of StokesMatMultVV, stokes.C:647-662.  V[j], S[j]: d stacked fields (component k of direction j).
// DETA = false when deta is identically zero (linear rheology): the deta * S0 * z term vanishes and S0 is not
// read.  The trace of the velocity gradient is the divergence StokesMatMult needs for the pressure rows
// (DV[i] restricted to component i IS DP[i] on that component, stokes.C:583-591), so it is written here
// (div may be null) instead of being recomputed by d more sweeps.
template <int D, bool DETA>
__global__ void k_st_node_vv(long N, double *__restrict__ V0, double *__restrict__ V1, double *__restrict__ V2,
                             const double *__restrict__ S0, const double *__restrict__ S1, const double *__restrict__ S2,
                             const double *__restrict__ eta, const double *__restrict__ deta, double *__restrict__ div) {
  double *V[3] = {V0, V1, V2};
  const double *S[3] = {S0, S1, S2};
  GS_LOOP(i, N) {
    double g[D][D], strain[D][D], S0v[D][D], z = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
      for (int k = 0; k < D; k++) g[j][k] = V[j][k * N + i];
    // the stored strain S0 is symmetric (StokesFunction writes the symmetrised gradient, stokes.C:718-722): the upper
    // triangle is read, D (D - 1) / 2 fewer loads per node
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
      for (int k = j; k < D; k++) { S0v[j][k] = DETA ? S[j][k * N + i] : 0.0; S0v[k][j] = S0v[j][k]; }
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
      for (int k = 0; k < D; k++) { strain[j][k] = 0.5 * (g[j][k] + g[k][j]); z += strain[j][k] * S0v[j][k]; }
    const double e = eta[i], de = DETA ? deta[i] : 0.0;
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
      for (int k = 0; k < D; k++) V[j][k * N + i] = DETA ? e * strain[j][k] + de * S0v[j][k] * z : e * strain[j][k];
    if (div) { double t = g[0][0] + g[1][1]; if (D == 3) t += g[D - 1][D - 1]; div[i] = t; }
  }
}

// Node loop of StokesFunction, stokes.C:710-725, with the rheology inlined (stokes.C:1920-1944).
template <int D>
__global__ void k_st_node_fn(long N, double *__restrict__ S0, double *__restrict__ S1, double *__restrict__ S2,
                             double *__restrict__ V0, double *__restrict__ V1, double *__restrict__ V2,
                             double *__restrict__ eta, double *__restrict__ deta, double *__restrict__ div,
                             int kind, double hardness, double expo, double eps, double gamma0) {
  double *V[3] = {V0, V1, V2};
  double *S[3] = {S0, S1, S2};
  GS_LOOP(i, N) {
    double g[D][D], s[D][D], gamma = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
      for (int k = 0; k < D; k++) g[j][k] = S[j][k * N + i];
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
      for (int k = 0; k < D; k++) { s[j][k] = 0.5 * (g[j][k] + g[k][j]); gamma += 0.5 * (s[j][k] * s[j][k]); }
    double e = 1.0, de = 0.0;
    if (kind == 1) {
      const double p = (1.0 - expo) / (2.0 * expo);
      // one pow: q^(p-1) = q^p / q (the second pow was half of this kernel's time at 128^3; the quotient differs from
      // it by an ulp, far inside the 1e-10 bar)
      const double q = eps + gamma / gamma0, qp = pow(q, p);
      e = hardness * qp;
      de = (fabs(expo) > 1.0e-5) ? hardness * p / gamma0 * (qp / q) : 0.0;
    }
    eta[i] = e; deta[i] = de;
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
      for (int k = 0; k < D; k++) { V[j][k * N + i] = e * s[j][k]; S[j][k * N + i] = s[j][k]; }
    if (div) { double t = g[0][0] + g[1][1]; if (D == 3) t += g[D - 1][D - 1]; div[i] = t; }    // stokes.C:746
  }
}

// The two node loops for d = 3 on node PAIRS (N even; every array is an allocation of the handle, 16-byte aligned): each
// field is read and written 16 bytes at a time.  Same arithmetic per node as k_st_node_vv / k_st_node_fn.
__device__ __forceinline__ double &comp(double2 &v, int q) { return q ? v.y : v.x; }
// SYM: the stress goes to the 6 slots of T (stokes_op::T) instead of overwriting the 9 gradient fields.
template <bool DETA, bool SYM>
__global__ __launch_bounds__(256) void k_st_node_vv_pair(long N, double *__restrict__ V0, double *__restrict__ V1, double *__restrict__ V2,
                                                         const double *__restrict__ S0, const double *__restrict__ S1, const double *__restrict__ S2,
                                                         const double *__restrict__ eta, const double *__restrict__ deta, double *__restrict__ div,
                                                         double *__restrict__ T) {
  double *V[3] = {V0, V1, V2};
  const double *S[3] = {S0, S1, S2};
  const long half = N >> 1;
  GS_LOOP(t, half) {
    double2 g[3][3], s0[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) g[j][k] = ((const double2 *)(V[j] + k * N))[t];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = j; k < 3; k++) { s0[j][k] = DETA ? ((const double2 *)(S[j] + k * N))[t] : make_double2(0.0, 0.0); s0[k][j] = s0[j][k]; }
    double2 e = ((const double2 *)eta)[t], de = DETA ? ((const double2 *)deta)[t] : make_double2(0.0, 0.0);
    double2 out[3][3], dv;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      double strain[3][3], z = 0.0;
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) { strain[j][k] = 0.5 * (comp(g[j][k], q) + comp(g[k][j], q)); z += strain[j][k] * comp(s0[j][k], q); }
      const double eq = comp(e, q), deq = comp(de, q);
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) comp(out[j][k], q) = DETA ? eq * strain[j][k] + deq * comp(s0[j][k], q) * z : eq * strain[j][k];
      double tr = comp(g[0][0], q) + comp(g[1][1], q); tr += comp(g[2][2], q);
      comp(dv, q) = tr;
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (SYM) { if (j <= k) ((double2 *)(T + (long)(j + k + j * k) * N))[t] = out[j][k]; }
        else ((double2 *)(V[j] + k * N))[t] = out[j][k];
      }
    if (div) ((double2 *)div)[t] = dv;
  }
}

template <bool SYM>
__global__ __launch_bounds__(256) void k_st_node_fn_pair(long N, double *__restrict__ S0, double *__restrict__ S1, double *__restrict__ S2,
                                                         double *__restrict__ V0, double *__restrict__ V1, double *__restrict__ V2,
                                                         double *__restrict__ eta, double *__restrict__ deta, double *__restrict__ div,
                                                         int kind, double hardness, double expo, double eps, double gamma0, double *__restrict__ T) {
  double *V[3] = {V0, V1, V2};
  double *S[3] = {S0, S1, S2};
  const long half = N >> 1;
  GS_LOOP(t, half) {
    double2 g[3][3], sv[3][3], tv[3][3], e2, de2, dv;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) g[j][k] = ((const double2 *)(S[j] + k * N))[t];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      double s[3][3], gamma = 0.0;
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) { s[j][k] = 0.5 * (comp(g[j][k], q) + comp(g[k][j], q)); gamma += 0.5 * (s[j][k] * s[j][k]); }
      double e = 1.0, de = 0.0;
      if (kind == 1) {
        const double p = (1.0 - expo) / (2.0 * expo);
        const double qq = eps + gamma / gamma0, qp = pow(qq, p);
        e = hardness * qp;
        de = (fabs(expo) > 1.0e-5) ? hardness * p / gamma0 * (qp / qq) : 0.0;
      }
      comp(e2, q) = e; comp(de2, q) = de;
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) { comp(tv[j][k], q) = e * s[j][k]; comp(sv[j][k], q) = s[j][k]; }
      double tr = comp(g[0][0], q) + comp(g[1][1], q); tr += comp(g[2][2], q);
      comp(dv, q) = tr;
    }
    ((double2 *)eta)[t] = e2; ((double2 *)deta)[t] = de2;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (SYM) { if (j <= k) { ((double2 *)(T + (long)(j + k + j * k) * N))[t] = tv[j][k]; ((double2 *)(S[j] + k * N))[t] = sv[j][k]; } }
        else { ((double2 *)(V[j] + k * N))[t] = tv[j][k]; ((double2 *)(S[j] + k * N))[t] = sv[j][k]; }
      }
    if (div) ((double2 *)div)[t] = dv;
  }
}

// Final scatter: velocity rows = yL (+ grad p), pressure rows = div v, minus force
// (scatterLV/VG, VecAXPY stokes.C:513-517,750-756).  Any of yL / gp0 / p2 may be null.
template <int D>
__global__ void k_st_out(long N, int gs, const int *__restrict__ ixL, const double *__restrict__ yL,
                         const double *__restrict__ yL1, const double *__restrict__ yL2,
                         const double *__restrict__ gp0, const double *__restrict__ gp1, const double *__restrict__ gp2,
                         const double *__restrict__ p2, int po, const double *__restrict__ force, double *__restrict__ out,
                         const double *__restrict__ G, long cs,         // G (may be null): one more velocity term (d fields), added after yL2
                         const double *__restrict__ p2b = nullptr, const double *__restrict__ p2c = nullptr) {   // pressure rows (p2 + p2b) + p2c
  GS_LOOP(l, N) {                                                       // cs: component stride of out / force (see k_st_local)
    const int n = ixL[l];
    if (n < 0) continue;
    const long o = (long)n * gs;
    if (yL || gp0) {
      double v[D];
#pragma unroll
      for (int k = 0; k < D; k++) {
        // yL (+ yL1 + yL2): the terms -DV[j] V[j] of stokes.C:668-671 / :737-740, summed in the order j = 0, 1, 2
        double t = yL ? yL[k * N + l] : 0.0;
        if (yL1) t = t + yL1[k * N + l];
        if (yL2) t = t + yL2[k * N + l];
        if (G) t = t + G[k * N + l];
        v[k] = t;
      }
      if (gp0) {
        const double g0 = gp0[l], g1 = gp1[l], g2 = (D == 3) ? gp2[l] : 0.0;
        if (yL) { v[0] += 1.0 * g0; v[1] += 1.0 * g1; if (D == 3) v[D - 1] += 1.0 * g2; }
        else { v[0] = g0; v[1] = g1; if (D == 3) v[D - 1] = g2; }
      }
#pragma unroll
      for (int k = 0; k < D; k++) { if (force) v[k] += -1.0 * force[o + k * cs]; out[o + k * cs] = v[k]; }
    }
    if (p2) { double w = p2[l]; if (p2b) w = w + p2b[l]; if (p2c) w = w + p2c[l]; if (force) w += -1.0 * force[o + po]; out[o + po] = w; }
  }
}

// The final scatter of StokesMatMult / StokesFunction in 3-D (all terms present, 16-byte aligned output): UN nodes in
// flight per thread, the node's four values leave as two 16-byte stores.  Same sums in the same order as k_st_out.
template <int UN>
__global__ __launch_bounds__(256) void k_st_out4(long N, const int *__restrict__ ixL, const double *__restrict__ yL,
                                                 const double *__restrict__ yL1, const double *__restrict__ yL2,
                                                 const double *__restrict__ gp0, const double *__restrict__ gp1, const double *__restrict__ gp2,
                                                 const double *__restrict__ p2, const double *__restrict__ force, double *__restrict__ out,
                                                 const double *__restrict__ G) {                // G (may be null): as in k_st_out
  const long T = (long)gridDim.x * blockDim.x;
  for (long l0 = blockIdx.x * (long)blockDim.x + threadIdx.x; l0 < N; l0 += UN * T) {
    int n[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) { const long l = l0 + u * T; n[u] = l < N ? ixL[l] : -1; }
    double v[UN][4];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long l = (l0 + u * T < N) ? l0 + u * T : 0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        double t = yL[k * N + l];
        t = t + yL1[k * N + l];
        t = t + yL2[k * N + l];
        if (G) t = t + G[k * N + l];
        v[u][k] = t;
      }
      v[u][0] += 1.0 * gp0[l]; v[u][1] += 1.0 * gp1[l]; v[u][2] += 1.0 * gp2[l];
      v[u][3] = p2[l];
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      if (n[u] < 0) continue;
      double2 *o2 = (double2 *)(out + 4L * n[u]);
      if (force) {
        const double2 f0 = ((const double2 *)(force + 4L * n[u]))[0], f1 = ((const double2 *)(force + 4L * n[u]))[1];
        v[u][0] += -1.0 * f0.x; v[u][1] += -1.0 * f0.y; v[u][2] += -1.0 * f1.x; v[u][3] += -1.0 * f1.y;
      }
      o2[0] = make_double2(v[u][0], v[u][1]); o2[1] = make_double2(v[u][2], v[u][3]);
    }
  }
}

// ... and on node PAIRS (N even): every term array is read 16 bytes at a time.  Same sums in the same order.
template <int UN>
__global__ __launch_bounds__(256) void k_st_out4p(long N, const int *__restrict__ ixL, const double *__restrict__ yL,
                                                  const double *__restrict__ yL1, const double *__restrict__ yL2,
                                                  const double *__restrict__ gp0, const double *__restrict__ gp1, const double *__restrict__ gp2,
                                                  const double *__restrict__ p2, const double *__restrict__ force, double *__restrict__ out,
                                                  const double *__restrict__ G, const double *__restrict__ p2b, const double *__restrict__ p2c, const StGrid sg) {
  // p2b, p2c (may be null): the pressure rows are (p2 + p2b) + p2c -- the divergence as the sum of its three terms
  const long T = (long)gridDim.x * blockDim.x, half = N >> 1;
  const double *gp[3] = {gp0, gp1, gp2};
  for (long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x; t0 < half; t0 += UN * T) {
    int2 n[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) { const long t = t0 + u * T; n[u] = t < half ? st_pair_ix(ixL, t, sg) : make_int2(-1, -1); }
    double2 v[UN][4];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long t = (t0 + u * T < half) ? t0 + u * T : 0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        double2 s = ((const double2 *)(yL + k * N))[t];
        const double2 b = ((const double2 *)(yL1 + k * N))[t], c = ((const double2 *)(yL2 + k * N))[t];
        s.x = s.x + b.x; s.y = s.y + b.y;
        s.x = s.x + c.x; s.y = s.y + c.y;
        if (G) { const double2 g = ((const double2 *)(G + k * N))[t]; s.x = s.x + g.x; s.y = s.y + g.y; }
        if (gp0) { const double2 q = ((const double2 *)gp[k])[t]; s.x += 1.0 * q.x; s.y += 1.0 * q.y; }      // (null: grad p is inside the y terms)
        v[u][k] = s;
      }
      v[u][3] = ((const double2 *)p2)[t];
      if (p2b) { const double2 b = ((const double2 *)p2b)[t], c = ((const double2 *)p2c)[t]; v[u][3].x = (v[u][3].x + b.x) + c.x; v[u][3].y = (v[u][3].y + b.y) + c.y; }
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int nn = h ? n[u].y : n[u].x;
        if (nn < 0) continue;
        double w0 = h ? v[u][0].y : v[u][0].x, w1 = h ? v[u][1].y : v[u][1].x, w2 = h ? v[u][2].y : v[u][2].x, w3 = h ? v[u][3].y : v[u][3].x;
        double2 *o2 = (double2 *)(out + 4L * nn);
        if (force) {
          const double2 f0 = ((const double2 *)(force + 4L * nn))[0], f1 = ((const double2 *)(force + 4L * nn))[1];
          w0 += -1.0 * f0.x; w1 += -1.0 * f0.y; w2 += -1.0 * f1.x; w3 += -1.0 * f1.y;
        }
        o2[0] = make_double2(w0, w1); o2[1] = make_double2(w2, w3);
      }
    }
  }
}

// Gather / scatter of the d = 3 velocity blocks on COMPONENT-MAJOR vectors (component c of interior node n at c I + n), on node
// pairs: the full-grid side moves 16 bytes per access, the interior side 8 bytes with neighbouring lanes on neighbouring
// addresses (an interior run of the last dimension is contiguous in both).  N even.  The same sums in the same order as k_st_out.
template <int UN>
__global__ __launch_bounds__(256) void k_st_local_cm3p(long N, long I, const int *__restrict__ ixL, const double *__restrict__ src, double *__restrict__ xL, const StGrid sg) {
  const long T = (long)gridDim.x * blockDim.x, half = N >> 1;
  for (long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x; t0 < half; t0 += UN * T) {
    int2 n[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) { const long t = t0 + u * T; n[u] = t < half ? st_pair_ix(ixL, t, sg) : make_int2(-1, -1); }
    double2 v[UN][3];
#pragma unroll
    for (int u = 0; u < UN; u++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        v[u][c].x = n[u].x >= 0 ? src[c * I + n[u].x] : 0.0;
        v[u][c].y = n[u].y >= 0 ? src[c * I + n[u].y] : 0.0;
      }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long t = t0 + u * T;
      if (t >= half) continue;
#pragma unroll
      for (int c = 0; c < 3; c++) ((double2 *)(xL + c * N))[t] = v[u][c];
    }
  }
}
struct StTerms3 { const double *p[4][3]; int n; };      // up to four terms of three component fields each, summed in order
template <int UN>
__global__ __launch_bounds__(256) void k_st_out_cm3p(long N, long I, const int *__restrict__ ixL, StTerms3 tm, double *__restrict__ out, const StGrid sg) {
  const long T = (long)gridDim.x * blockDim.x, half = N >> 1;
  for (long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x; t0 < half; t0 += UN * T) {
    int2 n[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) { const long t = t0 + u * T; n[u] = t < half ? st_pair_ix(ixL, t, sg) : make_int2(-1, -1); }
    double2 v[UN][3];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long t = (t0 + u * T < half) ? t0 + u * T : 0;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double2 s = ((const double2 *)tm.p[0][c])[t];
        for (int q = 1; q < tm.n; q++) { const double2 b = ((const double2 *)tm.p[q][c])[t]; s.x = s.x + b.x; s.y = s.y + b.y; }
        v[u][c] = s;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; u++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (n[u].x >= 0) out[c * I + n[u].x] = v[u][c].x;
        if (n[u].y >= 0) out[c * I + n[u].y] = v[u][c].y;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// k_st_zfused16: the z direction of the viscous block of StokesMatMultVV in ONE launch (d = 3, contiguous lines of 68 .. 128 points,
// P % 4 == 0).  The separate-pass route computes V[2] = D_z xL (3 fields), runs the node loop over all nine gradient fields, and takes
// -D_z of the three stress fields tau_z. -- the gradient and the stress along z each cross HBM twice.  Here a tile of 16 z-lines holds
// all three components of its nodes, so
//   A  v_c lines -> LDS, parity-split (as the sweep kernels do)
//   B  G_z,c = D_z v_c on the matrix cores (v_mfma_f64_16x16x4, even / odd halves), written to LDS in node order
//   C  the node loop of stokes.C:647-662 in LOADER layout -- one thread per (line, point pair + mirror pair), so that every operand
//      (G_x, G_y, S0, eta, eta') comes as coalesced 16-byte pieces -- with G_z from LDS; tau_x., tau_y. overwrite G_x, G_y in place as
//      in the separate route, the trace goes to `div`, tau_z. goes back to LDS, parity-split
//   D  -D_z tau_z,c on the matrix cores -> yz
// 216 B/node instead of 312 for the three kernels it replaces (z third of the gradient launch, node loop, z third of the divergence
// launch); the matrix work is < 10 % of its time at P = 128.  The phases of a workgroup do not overlap each other, so two workgroups
// of 256 threads share a CU (one waits for memory while the other computes): 115 us = 3.9 TB/s at 128^3 against 137 us for the three
// kernels (rocprofv3, same run) -- the dependent rounds of loads of a tile (v; then the operands of slot 0, of slot 1) leave the memory
// system under-occupied compared with a streaming node loop (6 TB/s); more rounds in flight need registers the node loop's 22
// 16-byte operands per slot already take (252 VGPRs).  Built and lost, one process each: requesting the next tile's v lines one tile
// ahead, under stage D (MatVV 230 -> 241 us); the node loop as four rounds with a rolling window of two rounds of operands in flight
// (spills 38 VGPRs: MatVV 229 -> 235 us, StokesFunction 296 -> 322; without the staging of G_z, images written after one more barrier:
// 26 spilled in the eta' variant, 229 -> 240 / 296 -> 302).  Same arithmetic in the same order as the separate route: the same bits (tests).
typedef double zf_v4 __attribute__((ext_vector_type(4)));
struct ZfParams {
  int P, H; unsigned nlines, ntiles; long N;
  const double *xL; double *Vx, *Vy;
  const double *S0, *S1, *S2, *eta, *deta;
  double *div, *yz;
  const double *fragE, *fragO;
  // StokesFunction (MODE 2): Vx / Vy are strain[0] / strain[1] (gradient in, symmetrised upper triangle out), Sz = strain[2];
  // eta_w / deta_w and the six-slot stress T are written; the rheology of stokes.C:1920-1944
  double *Sz, *eta_w, *deta_w, *T;
  int kind; double hardness, expo, eps, gamma0;
  // pL (may be null): the pressure with its face values extrapolated (k_st_pfaces) -- the stress leaves as tau - p I, so that the
  // three divergence sweeps deliver -div tau + grad p and neither the pressure-gradient sweeps nor their term in the scatter exist
  const double *pL;
};
// Image row pitch HP + ZF_PAD doubles.  With an ODD pad the MFMA operand reads (lane (l16, kq) reads points kq + 4k, kq + 4k + 4 of line
// l16: ds_read2_b64) are free of bank conflicts, with pitch = 2 mod 32 each is a 2-way conflict (tools/lds_probe.hip,
// profiles/r06_lds_probe.txt).  Measured here (tools/ldspad_ab.sh, profiles/r06_lds_probe.txt): the callback does not change (255.8 / 257.2
// against 256.9 / 255.6 us at 128^3) -- the launch is a byte stream, the reads sit under it -- while the 8-byte park of an odd pitch costs the
// FOLD variants four spilled VGPRs at their 256-register limit.  Shipped: 2 (ZF_PAD = 1 builds and passes the suite).
#ifndef ZF_PAD
#define ZF_PAD 2
#endif
constexpr int ZF_KS = 16, ZF_LDJ = 4 * ZF_KS + ZF_PAD, ZF_NT = 16, ZF_PG = 130;   // k-steps; image row pitch; lines per tile; row pitch of G_z
__device__ __forceinline__ void zf_put2(double *dst, double2 v) {
  if (ZF_LDJ % 2 == 0) *(double2 *)dst = v; else { dst[0] = v.x; dst[1] = v.y; }
}
// MODE 0 / 1: the node loop of StokesMatMultVV without / with the eta' S0 z term (k_st_node_vv_pair); 2: the node loop of
// StokesFunction with the six-component storage (k_st_node_fn_pair<true>: rheology, eta, eta', symmetrised strain as state)
template <int MODE, bool FOLD>
__global__ __launch_bounds__(256, 2) void k_st_zfused16(const ZfParams p) {
  constexpr bool DETA = MODE == 1;
  // Two workgroups of 256 threads per CU (out of phase with each other: one waits for memory while the other computes), 50.7 KB of
  // LDS each: the parity-split images E[3], O[3]; G_z in node order ALIASES them (it lives between the end of the stage-1 chains and
  // the moment every thread has taken its own values into registers).
  constexpr int IMG = ZF_NT * ZF_LDJ, GPL = ZF_NT * ZF_PG;
  static_assert(3 * GPL <= 6 * IMG, "G_z must fit in the image space");
  __shared__ __attribute__((aligned(16))) double sI[6 * IMG];
  __shared__ __attribute__((aligned(16))) double sP[FOLD ? GPL : 2];   // FOLD: the tile's pressure in node order (16.6 KB)
  const int tid = threadIdx.x, lane = tid & 63, mt = tid >> 6;         // wave = m-tile; every wave runs the three fields
  const int kq = lane >> 4, l16 = lane & 15;
  const int P = p.P, H = p.H, nn = P - 1;
  const long N = p.N;
  const int frag = l16 * ZF_LDJ + kq;
  const int oi = mt * 16 + l16;                                        // output point of this lane (and its mirror nn - oi)
  // the two loader / node slots of this thread: line sl, points (sj, sj + 1) and their mirrors (nn - sj - 1, nn - sj)
  int sl[2], sj[2];
#pragma unroll
  for (int u = 0; u < 2; u++) { const int id = tid + 256 * u; sl[u] = id >> 5; sj[u] = 2 * (id & 31); }
  // (the matrix fragments are fetched per stage from L2 -- 64 KB per workgroup and stage -- rather than held across the node loop,
  // whose operands need the registers)
  const double *fragE = p.fragE, *fragO = p.fragO;
  auto chains = [&](zf_v4 (&ce)[3], zf_v4 (&co)[3]) {
    asm volatile("" : "+s"(fragE), "+s"(fragO));                       // no reuse of the previous stage's fragment registers
    double ae[ZF_KS], ao[ZF_KS];
#pragma unroll
    for (int s = 0; s < ZF_KS; s++) { ae[s] = fragE[((long)(mt * ZF_KS + s)) * 64 + lane]; ao[s] = fragO[((long)(mt * ZF_KS + s)) * 64 + lane]; }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      ce[c] = zf_v4{0.0, 0.0, 0.0, 0.0}; co[c] = zf_v4{0.0, 0.0, 0.0, 0.0};
      const double *fE = sI + c * IMG + frag, *fO = sI + (3 + c) * IMG + frag;
#pragma unroll
      for (int s = 0; s < ZF_KS; s++) {
        ce[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fE[4 * s], ae[s], ce[c], 0, 0, 0);
        co[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fO[4 * s], ao[s], co[c], 0, 0, 0);
      }
    }
  };
  for (unsigned tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    bool live[2]; long a[2], m[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const unsigned gl = tile * ZF_NT + sl[u];
      live[u] = sj[u] < H && gl < p.nlines;                            // H even: the pair is in or out together
      a[u] = (long)gl * P + sj[u]; m[u] = (long)gl * P + (nn - sj[u] - 1);
    }
    // ---- A: v_c -> parity-split images (dead slots write the zero padding)
    {
      double2 rj[2][3], rm[2][3];
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          rj[u][c] = make_double2(0.0, 0.0); rm[u][c] = rj[u][c];
          if (live[u]) { rj[u][c] = *(const double2 *)(p.xL + c * N + a[u]); rm[u][c] = *(const double2 *)(p.xL + c * N + m[u]); }
        }
      double2 pj[2], pm[2];
      if (FOLD) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
          pj[u] = make_double2(0.0, 0.0); pm[u] = pj[u];
          if (live[u]) { pj[u] = *(const double2 *)(p.pL + a[u]); pm[u] = *(const double2 *)(p.pL + m[u]); }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          zf_put2(sI + c * IMG + sl[u] * ZF_LDJ + sj[u], make_double2(rj[u][c].x + rm[u][c].y, rj[u][c].y + rm[u][c].x));
          zf_put2(sI + (3 + c) * IMG + sl[u] * ZF_LDJ + sj[u], make_double2(rj[u][c].x - rm[u][c].y, rj[u][c].y - rm[u][c].x));
        }
      if (FOLD) {
#pragma unroll
        for (int u = 0; u < 2; u++)
          if (live[u]) {                                               // (a dead slot's mirror index would land on a live slot's points)
            *(double2 *)(sP + sl[u] * ZF_PG + sj[u]) = pj[u];
            *(double2 *)(sP + sl[u] * ZF_PG + (nn - sj[u] - 1)) = pm[u];
          }
      }
    }
    __syncthreads();
    // ---- B: G_z,c = D_z v_c, to LDS in node order (over the images, once every wave has finished its chains)
    {
      zf_v4 ce[3], co[3];
      chains(ce, co);
      __syncthreads();
      if (oi < H) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            double *row = sI + c * GPL + (4 * r + kq) * ZF_PG;
            row[oi] = ce[c][r] + co[c][r]; row[nn - oi] = co[c][r] - ce[c][r];
          }
      }
    }
    __syncthreads();
    // ---- C: node loop on the two slots' four nodes each; G_z first into registers (its space becomes the tau_z images)
    {
      double2 gz[2][3][2];
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
          gz[u][k][0] = *(const double2 *)(sI + k * GPL + sl[u] * ZF_PG + sj[u]);
          gz[u][k][1] = *(const double2 *)(sI + k * GPL + sl[u] * ZF_PG + (nn - sj[u] - 1));
        }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; u++) {
        double2 tz[3][2];                                              // tau_z,k of the pair / of the mirror pair
#pragma unroll
        for (int k = 0; k < 3; k++) { tz[k][0] = make_double2(0.0, 0.0); tz[k][1] = tz[k][0]; }
        if (live[u] && MODE == 2) {
          double2 g[3][3][2];
          const long off[2] = {a[u], m[u]};
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
              g[0][k][h] = *(const double2 *)(p.Vx + k * N + off[h]);
              g[1][k][h] = *(const double2 *)(p.Vy + k * N + off[h]);
              g[2][k][h] = gz[u][k][h];
            }
#pragma unroll
          for (int h = 0; h < 2; h++) {
            double2 sv[3][3], tv[3][3], e2, de2, dv;
#pragma unroll
            for (int q = 0; q < 2; q++) {
              double sn[3][3], gamma = 0.0;
#pragma unroll
              for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 3; k++) { sn[j][k] = 0.5 * (comp(g[j][k][h], q) + comp(g[k][j][h], q)); gamma += 0.5 * (sn[j][k] * sn[j][k]); }
              double e = 1.0, de = 0.0;
              if (p.kind == 1) {
                const double pw = (1.0 - p.expo) / (2.0 * p.expo);
                const double qq = p.eps + gamma / p.gamma0, qp = pow(qq, pw);
                e = p.hardness * qp;
                de = (fabs(p.expo) > 1.0e-5) ? p.hardness * pw / p.gamma0 * (qp / qq) : 0.0;
              }
              comp(e2, q) = e; comp(de2, q) = de;
#pragma unroll
              for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 3; k++) { comp(tv[j][k], q) = e * sn[j][k]; comp(sv[j][k], q) = sn[j][k]; }
              double tr = comp(g[0][0][h], q) + comp(g[1][1][h], q); tr += comp(g[2][2][h], q);
              comp(dv, q) = tr;
            }
            *(double2 *)(p.eta_w + off[h]) = e2; *(double2 *)(p.deta_w + off[h]) = de2;
            // state: the symmetrised strain in its upper triangle; stress: slot j + k + j k of T for j <= k, but (2,2): tau_zz is used
            // on chip only
#pragma unroll
            for (int k = 0; k < 3; k++) *(double2 *)(p.Vx + k * N + off[h]) = sv[0][k];
            *(double2 *)(p.Vy + 1 * N + off[h]) = sv[1][1]; *(double2 *)(p.Vy + 2 * N + off[h]) = sv[1][2];
            *(double2 *)(p.Sz + 2 * N + off[h]) = sv[2][2];
            if (FOLD) {                                                // the stress the divergence sweeps read: tau - p I
              const double2 pp = *(const double2 *)(sP + sl[u] * ZF_PG + (h ? nn - sj[u] - 1 : sj[u]));
#pragma unroll
              for (int k = 0; k < 3; k++) { tv[k][k].x -= pp.x; tv[k][k].y -= pp.y; }
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
              for (int k = j; k < 3; k++) *(double2 *)(p.T + (long)(j + k + j * k) * N + off[h]) = tv[j][k];
#pragma unroll
            for (int k = 0; k < 3; k++) tz[k][h] = tv[2][k];
            if (p.div) *(double2 *)(p.div + off[h]) = dv;
          }
        }
        if (live[u] && MODE != 2) {
          double2 g[3][3][2], s0[3][3][2], e[2], de[2];
          const long off[2] = {a[u], m[u]};
          const double *S[3] = {p.S0, p.S1, p.S2};
#pragma unroll
          for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
              g[0][k][h] = *(const double2 *)(p.Vx + k * N + off[h]);
              g[1][k][h] = *(const double2 *)(p.Vy + k * N + off[h]);
              g[2][k][h] = gz[u][k][h];
            }
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
              for (int k = j; k < 3; k++) { s0[j][k][h] = DETA ? *(const double2 *)(S[j] + k * N + off[h]) : make_double2(0.0, 0.0); s0[k][j][h] = s0[j][k][h]; }
            e[h] = *(const double2 *)(p.eta + off[h]);
            de[h] = DETA ? *(const double2 *)(p.deta + off[h]) : make_double2(0.0, 0.0);
          }
#pragma unroll
          for (int h = 0; h < 2; h++) {
            double2 out[3][3], dv;
#pragma unroll
            for (int q = 0; q < 2; q++) {
              double strain[3][3], z = 0.0;
#pragma unroll
              for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 3; k++) { strain[j][k] = 0.5 * (comp(g[j][k][h], q) + comp(g[k][j][h], q)); z += strain[j][k] * comp(s0[j][k][h], q); }
              const double eq = comp(e[h], q), deq = comp(de[h], q);
#pragma unroll
              for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 3; k++) comp(out[j][k], q) = DETA ? eq * strain[j][k] + deq * comp(s0[j][k][h], q) * z : eq * strain[j][k];
              double tr = comp(g[0][0][h], q) + comp(g[1][1][h], q); tr += comp(g[2][2][h], q);
              comp(dv, q) = tr;
            }
            if (FOLD) {                                                // tau - p I
              const double2 pp = *(const double2 *)(sP + sl[u] * ZF_PG + (h ? nn - sj[u] - 1 : sj[u]));
#pragma unroll
              for (int k = 0; k < 3; k++) { out[k][k].x -= pp.x; out[k][k].y -= pp.y; }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
              *(double2 *)(p.Vx + k * N + off[h]) = out[0][k];
              *(double2 *)(p.Vy + k * N + off[h]) = out[1][k];
              tz[k][h] = out[2][k];
            }
            if (p.div) *(double2 *)(p.div + off[h]) = dv;
          }
        }
        // tau_z,k parity-split: the pair is (t_j, t_{j+1}), the mirror pair (t_{nn-j-1}, t_{nn-j})
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double2 tp = tz[k][0], tm = tz[k][1];
          zf_put2(sI + k * IMG + sl[u] * ZF_LDJ + sj[u], make_double2(tp.x + tm.y, tp.y + tm.x));
          zf_put2(sI + (3 + k) * IMG + sl[u] * ZF_LDJ + sj[u], make_double2(tp.x - tm.y, tp.y - tm.x));
        }
      }
    }
    __syncthreads();
    // ---- D: yz_c = -D_z tau_z,c
    {
      zf_v4 ce[3], co[3];
      chains(ce, co);
      if (oi < H) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const unsigned ol = tile * ZF_NT + 4 * r + kq;
            if (ol < p.nlines) {
              double *row = p.yz + (long)c * N + (long)ol * P;
              row[oi] = -1.0 * (ce[c][r] + co[c][r]); row[nn - oi] = -1.0 * (co[c][r] - ce[c][r]);
            }
          }
      }
    }
    __syncthreads();                                                   // the images are rewritten by the next tile's phase A
  }
}

__global__ void k_st_fill(long n, double v, double *__restrict__ a) { GS_LOOP(i, n) a[i] = v; }

// ---------------------------------------------------------------------------------------------
struct stokes_op {
  int d = 0;
  std::vector<int> dims;
  long N = 0, I = 0;
  std::map<int, DiffMat> mats;
  // D with the end-point extrapolation of StokesPressureReduceOrder folded in, per extent (every extent of 3 .. 256 points;
  // in slab mode the matrix of dimension 0 is applied on the pencils): gp[i] = matsP * pL needs no extrapolation pass -- see st_pressure_gradient
  std::map<int, DiffMat> matsP; bool pext = false;
  // Uniform viscosity with eta' = 0 (the state after create and after a StokesFunction with the linear rheology): the viscous
  // block of the Jacobian is -eta/2 (sum_j D_j D_j v + grad div v) -- see st_viscous_uniform.  matsDD: D D per extent.
  std::map<int, DiffMat> matsDD; bool uniform_ok = false, eta_uniform = true; double eta_value = 1.0;
  std::vector<unsigned> innerP, ncolsP, innerV, ncolsV;      // DP[i] / DV[i] geometry
  int *ixL = nullptr;
  double *xL = nullptr, *yL = nullptr;                       // workV[0], workV[1]
  // StokesFunction with the linear rheology on the uniform-viscosity route (st_viscous_uniform) does not run the node loop
  // that would leave the symmetrised strain as state (stokes.C:722): it keeps ITS local vector (velocity with Dirichlet
  // values) in xF instead -- the later callbacks overwrite xL -- and whoever reads the strain (the state accessors, the VTK
  // writer, a Jacobian apply after eta' has been set by hand) first rebuilds it from xF: st_sync_strain.
  double *xF = nullptr; bool strain_stale = false;
  double *yLx[3] = {nullptr, nullptr, nullptr};              // serial handles: terms 1, 2 of the stress divergence (summed in the final scatter)
  double *V[3] = {nullptr, nullptr, nullptr};                // workV[2..]
  // d = 3 serial handles on lines of more than 64 points: the stress is symmetric, so the node loops write its 6 distinct
  // components only, into slots {0, 1, 2, 3, 5, 8} of T (9 N doubles): tau_jk = tau_kj sits at slot j + k + j k, chosen so
  // that the three fields the divergence sweep of direction j reads -- (j,0), (j,1), (j,2) -- are slots j, 2j+1, 3j+2: an
  // arithmetic progression of stride j + 1, which the sweep kernel takes as spaced-out input fields (sweep.h).  Likewise
  // the node loop of StokesFunction symmetrises c->strain in its upper triangle only (entries j <= k); the state
  // accessors rebuild the full tensor.  48 B/node less written per StokesFunction (128^3: 325 -> 306 us).  StokesMatMult
  // keeps its in-place node loop over all 9 fields: writing the 6 components elsewhere measured 13 us SLOWER there (the
  // gradient fields it has just read are the lines the Infinity Cache still holds; tools/stokes_ab.py).
  bool sym = false;
  double *T = nullptr;
  double *strain[3] = {nullptr, nullptr, nullptr};           // c->strain[]
  double *eta = nullptr, *deta = nullptr;
  double *pL = nullptr, *p2 = nullptr, *gp[3] = {nullptr, nullptr, nullptr};   // workP[]
  double *dirloc = nullptr, *force = nullptr;
  std::vector<double *> w0, w1;                              // pressure extrapolation weights per dim
  // StokesMatMultSchur: work vectors vG0, vG1 (stokes.C:530-532) and the built-in inner solver
  double *sv0 = nullptr, *sv1 = nullptr;
  chebhip_fgmres *inner = nullptr;
  int in_restart = 30, in_maxit = 10000, inner_its = 0;      // KSP defaults
  chebhip_reduce_fn in_reduce = nullptr;                     // slab mode: completes the inner solver's inner products
  void *in_reduce_ctx = nullptr;
  double in_rtol = 1e-5, in_atol = 1e-50;
  // the pressure-gradient chain (extrapolation + d scalar sweeps) is independent of the viscous chain between the
  // gather and the final scatter: it runs on a second stream (small grids leave most CUs idle per launch)
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // slab mode (multi-GPU): the handle owns the planes [lo, lo + dims[0]) of a grid whose dimension 0 has
  // gP0 points; everything along dimension 0 (sweeps, x-line pressure extrapolation) goes through `dim0`
  bool slab = false;
  // slab mode on a direct transport (slabx.hip): independent local sweeps parked here by a callback ride in the launch of the dimension-0
  // sweep it calls next (stokes_pencil_gather_try), so that the d directions of a gradient / divergence are ONE launch; what the driver
  // did not take is launched by st_flush_pending
  int npend = 0; const DiffMat *pend_m[8] = {}; SweepParams pend_sp[8] = {};
  int gP0 = 0, lo = 0;
  stokes_dim0_fn dim0 = nullptr;
  void *dim0_ctx = nullptr;
  bool deta_nonzero = false;                                 // deta == 0 everywhere: the node loop skips S0
  int rh_kind = 0; double rh_hard = 1.0, rh_expo = 1.0, rh_eps = 1.0, rh_g0 = 1.0;   // stokes.C:403
};

// Work arrays of one handle.  (Round 4 tried starting every array at a different multiple of 4352 bytes of its allocation, against
// a suspected HBM channel / bank alignment of the ~30 arrays a node loop streams: 128^3 power-law StokesMatMult 287 -> 305 us,
// StokesFunction 305 -> 310 us in an A/B in one process -- the 16-MiB-aligned allocations are the better placement.  Removed.)
static int st_alloc(double **p, size_t n) { SHIPCHK(hipMalloc((void **)p, n * sizeof(double))); SHIPCHK(hipMemset(*p, 0, n * sizeof(double))); return 0; }
static void st_free(double *p) { if (p) (void)hipFree(p); }

extern "C" int stokes_op_destroy(stokes_op *op) {
  if (!op) return 0;
  for (auto &kv : op->mats) diffmat_destroy(&kv.second);
  for (auto &kv : op->matsP) diffmat_destroy(&kv.second);
  for (auto &kv : op->matsDD) diffmat_destroy(&kv.second);
  if (op->slab) { op->pL = nullptr; op->gp[0] = nullptr; }      // parts of xL / V[0] / strain[0] (st_create)
  double *all[] = {op->xL, op->yL, op->V[0], op->V[1], op->V[2], op->strain[0], op->strain[1], op->strain[2], op->eta, op->deta,
                   op->pL, op->p2, op->gp[0], op->gp[1], op->gp[2], op->yLx[1], op->yLx[2], op->T, op->xF};
  for (double *p : all) st_free(p);
  st_free(op->dirloc); st_free(op->force);
  st_free(op->sv0); st_free(op->sv1);
  if (op->inner) chebhip_fgmres_destroy(op->inner);
  if (op->aux) (void)hipStreamDestroy(op->aux);
  if (op->ev_fork) (void)hipEventDestroy(op->ev_fork);
  if (op->ev_join) (void)hipEventDestroy(op->ev_join);
  for (double *p : op->w0) if (p) (void)hipFree(p);
  for (double *p : op->w1) if (p) (void)hipFree(p);
  if (op->ixL) (void)hipFree(op->ixL);
  delete op;
  return 0;
}

static bool st_zfused_ok(stokes_op *op);
// boundary node of the GLOBAL grid?  ind: local multi-index (dimension 0 is offset by op->lo in slab mode)
static inline bool st_is_bdy(const stokes_op *op, const int *ind) {
  const int g0 = ind[0] + op->lo;
  if (g0 == 0 || g0 == op->gP0 - 1) return true;
  for (int j = 1; j < op->d; j++) if (ind[j] == 0 || ind[j] == op->dims[j] - 1) return true;
  return false;
}

static int st_create(int d, const int *gdims, int lo, int hi, stokes_dim0_fn dim0, void *dim0_ctx, stokes_op **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!gdims || d < 2 || d > 3) return chebhip_fail(CHEBHIP_ERR_DIMS, "d = %d: Stokes needs d = 2 or 3 (stokes.C:1036)", d);
  const bool slab = dim0 != nullptr;
  for (int k = 0; k < d; k++) {
    if (gdims[k] < 3) return chebhip_fail(CHEBHIP_ERR_SIZE, "dims[%d] = %d but must be >= 3", k, gdims[k]);
    if (gdims[k] > 4096) return chebhip_fail(CHEBHIP_ERR_ARG, "dims[%d] = %d: at most 4096 points per line", k, gdims[k]);
  }
  if (slab && !(0 <= lo && lo < hi && hi <= gdims[0])) return chebhip_fail(CHEBHIP_ERR_ARG, "slab planes [%d, %d) outside 0..%d", lo, hi, gdims[0]);
  std::vector<int> dims(gdims, gdims + d);
  if (slab) dims[0] = hi - lo; else { lo = 0; hi = gdims[0]; }
  long N = 1;
  for (int k = 0; k < d; k++) {
    N *= dims[k];
    if (N * d > 0x7fffffffL) return chebhip_fail(CHEBHIP_ERR_DIMS, "tensor of more than 2^31-1 values");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return chebhip_fail(CHEBHIP_ERR_DEVICE, "no usable HIP device; libchebhip has no CPU fallback");
  stokes_op *op = new (std::nothrow) stokes_op;
  if (!op) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  op->d = d; op->dims = dims; op->N = N;
  op->slab = slab; op->gP0 = gdims[0]; op->lo = lo; op->dim0 = dim0; op->dim0_ctx = dim0_ctx;
#define OPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { stokes_op_destroy(op); \
    return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
#define OPRC(expr) do { int rc_ = (expr); if (rc_) { stokes_op_destroy(op); return rc_; } } while (0)
  for (int k = 0; k < d; k++)      // dimension 0: the global extent (in slab mode it is applied on pencils)
    if (!op->mats.count(gdims[k])) { DiffMat m; OPCHK(diffmat_create(gdims[k], &m)); op->mats[gdims[k]] = m; }
  op->pext = !opt(OPT_PRESSURE_PASSES);                     // "pressure_passes": the three extrapolation passes of the reference (A/B)
  for (int k = 0; k < d; k++) op->pext = op->pext && gdims[k] >= 3 && op->mats[gdims[k]].KS != 0;
  if (op->pext)
    for (int k = 0; k < d; k++)
      if (!op->matsP.count(gdims[k])) { DiffMat m; OPCHK(diffmat_create_pext(gdims[k], &m)); op->matsP[gdims[k]] = m; }
  op->uniform_ok = op->pext && !slab && d >= 2 && !opt(OPT_GENERAL_VISCOUS);        // "general_viscous": A/B
  if (op->uniform_ok)
    for (int k = 0; k < d; k++)
      if (!op->matsDD.count(gdims[k])) { DiffMat m; OPCHK(diffmat_create_dd(gdims[k], &m)); op->matsDD[gdims[k]] = m; }
  {  // ixLP of StokesSetupDomain (stokes.C:791-879): interior index or -1, BlockIt order (of this slab)
    std::vector<int> ixL((size_t)N), ind(d, 0);
    long g = 0;
    for (long l = 0; l < N; l++) {
      ixL[l] = st_is_bdy(op, ind.data()) ? -1 : (int)g++;
      for (int j = d - 1; j >= 0; j--) { if (++ind[j] < dims[j]) break; ind[j] = 0; }
    }
    op->I = g;
    OPCHK(hipMalloc((void **)&op->ixL, (size_t)N * sizeof(int)));
    OPCHK(hipMemcpy(op->ixL, ixL.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice));
  }
  op->innerP.resize(d); op->ncolsP.resize(d); op->innerV.resize(d); op->ncolsV.resize(d);
  for (int k = 0; k < d; k++) {
    unsigned in = 1; for (int r = k + 1; r < d; r++) in *= dims[r];
    op->innerP[k] = in; op->ncolsP[k] = (unsigned)(N / dims[k]);
    op->innerV[k] = in; op->ncolsV[k] = (unsigned)(N / dims[k]) * d;          // DV[k]: the same lines for d stacked fields
  }
  const size_t nd = (size_t)N * d;
  // Slab mode: the gradient along dimension 0 and the pressure gradient along it share ONE round trip to pencils (d + 1 stacked
  // fields, see st_gradient_and_pressure_gradient_slab): pL sits right behind xL, and gp[0] is the (d+1)-th field of whichever
  // array receives the gradient (V[0] in StokesMatMult, strain[0] in StokesFunction) -- set per call, not allocated.
  const size_t nd1 = slab ? nd + (size_t)N : nd;
  struct Req { double **p; size_t n; };
  std::vector<Req> reqs;
  auto want = [&](double **p, size_t n) { reqs.push_back({p, n}); };
  want(&op->xL, nd1); want(&op->yL, nd);
  if (op->uniform_ok) want(&op->xF, nd);
  for (int j = 1; j < d; j++) want(&op->yLx[j], nd);
  for (int j = 0; j < d; j++) {
    want(&op->V[j], j == 0 ? nd1 : nd); want(&op->strain[j], j == 0 ? nd1 : nd);
    if (!(slab && j == 0)) want(&op->gp[j], (size_t)N);
  }
  want(&op->eta, (size_t)N); want(&op->deta, (size_t)N);
  // (the spaced-out input fields of the six-component storage exist in the 16-byte kernels only: not with "general_kernels")
  if (!slab && d == 3 && (N & 1) == 0 && (size_t)N * 9 * 8 < 0x38000000ull && !opt(OPT_FULL_STRESS) && !opt(OPT_GENERAL_KERNELS)) {
    bool ok = true;
    for (int k = 0; k < d; k++) ok = ok && op->mats[dims[k]].KS >= 16 && (dims[k] & 1) == 0;       // the long-line 16-byte kernel, both tilings
    const int nt_last = op->mats[dims[d - 1]].KS == 16 ? 64 : 32;                                   // lines per tile of the contiguous direction
    ok = ok && ((N / dims[d - 1]) % nt_last) == 0;
    if (ok) { want(&op->T, (size_t)N * 9); op->sym = true; }
  }
  if (!slab) want(&op->pL, (size_t)N);
  want(&op->p2, (size_t)N);
  // One hipMalloc per array.  (Round 5 measured the alternative -- ONE allocation with every array at a chosen offset: contiguous, with
  // 2..34-MiB gaps, with per-array shifts of 256 B .. 2 MiB, and with exactly the offsets separate allocations get -- because handles of
  // one process differ by 6 % and keep their speed for life (128^3 power-law StokesMatMult 275 / 283 / 291 us; tools/placement_probe.py).
  // Every arena layout ran at 279..290 us, the separate allocations of a fresh process at 275..278: the virtual layout is not the lever,
  // the physical blocks behind each allocation are (the difference vanishes when a profiler serialises the launches, i.e. it lives in
  // what one launch leaves in the caches for the next).  profiles/r05_placement*.txt, r05_arena_sweep*.txt; DESIGN_history.md.)
  for (const Req &r : reqs) OPRC(st_alloc(r.p, r.n));
  if (slab) { op->gp[0] = op->V[0] + nd; op->pL = op->xL + nd; }
  hipLaunchKernelGGL(k_st_fill, dim3(sgrid(N)), dim3(256), 0, nullptr, N, 1.0, op->eta);
  // Lagrange weights of the interior nodes x_1..x_{P-2} at x_0 and x_{P-1} (the polyInterp functional)
  op->w0.assign(d, nullptr); op->w1.assign(d, nullptr);
  for (int k = 0; k < d; k++) {
    const int P = gdims[k], m = P - 2;
    std::vector<long double> x(P);
    for (int i = 0; i < P; i++) x[i] = cosl(3.14159265358979323846264338327950288L * i / (P - 1));
    std::vector<double> a(m), b(m);
    for (int j = 1; j <= m; j++) {
      long double l0 = 1.0L, l1 = 1.0L;
      for (int q = 1; q <= m; q++) if (q != j) { l0 *= (x[0] - x[q]) / (x[j] - x[q]); l1 *= (x[P - 1] - x[q]) / (x[j] - x[q]); }
      a[j - 1] = (double)l0; b[j - 1] = (double)l1;
    }
    OPCHK(hipMalloc((void **)&op->w0[k], (size_t)m * sizeof(double)));
    OPCHK(hipMalloc((void **)&op->w1[k], (size_t)m * sizeof(double)));
    OPCHK(hipMemcpy(op->w0[k], a.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice));
    OPCHK(hipMemcpy(op->w1[k], b.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice));
  }
  {
    // The pressure chain on a second stream pays on large grids (128^3 StokesMatMult 298 against 325 us, 120^3 260 against
    // 275); below about 100^3 the fork / join events cost more than the overlap gives (64^3: 70.5 against 66.5 us, and
    // 62 us with the one-launch gradients of the one-stream path; 96^3 .. 112^3: a tie), so smaller grids stay on one stream.
    // Round 5: where the fused-z route runs (st_zfused_ok) the pressure-gradient sweeps are jobs of its first launch instead, and
    // the second stream is only made with option "stokes_pressure_stream" = 1 (A/B)
    op->N = N;
    const bool z1 = st_zfused_ok(op) && !opt(OPT_STOKES_PRESSURE_STREAM);
    if (!opt(OPT_STOKES_SINGLE_STREAM) && !slab && N >= 1200000 && !z1) {         // "stokes_single_stream": read when the handle is created
      OPCHK(hipStreamCreateWithFlags(&op->aux, hipStreamNonBlocking));
      OPCHK(hipEventCreateWithFlags(&op->ev_fork, hipEventDisableTiming));
      OPCHK(hipEventCreateWithFlags(&op->ev_join, hipEventDisableTiming));
    }
  }
  OPCHK(hipDeviceSynchronize());
#undef OPCHK
#undef OPRC
  *out = op;
  return 0;
}

// Diagnostic (tools/placement_probe.py; not in the header): device addresses of the work arrays in request order
extern "C" int chebhip_debug_stokes_arrays(stokes_op *op, int cap, unsigned long long *addr) {
  if (!op || !addr) return -1;
  double *all[] = {op->xL, op->yL, op->xF, op->yLx[1], op->yLx[2], op->V[0], op->strain[0], op->gp[0], op->V[1], op->strain[1], op->gp[1],
                   op->V[2], op->strain[2], op->gp[2], op->eta, op->deta, op->T, op->pL, op->p2};
  int n = 0;
  for (double *p : all) if (n < cap) addr[n++] = (unsigned long long)(uintptr_t)p;
  return n;
}

extern "C" int stokes_op_create(int d, const int *dims, stokes_op **out) { return st_create(d, dims, 0, 0, nullptr, nullptr, out); }

// Slab of the planes [lo, hi) of grid dimension 0 (multi-GPU).  Vector layouts are those of the serial
// operator restricted to the slab: the global vectors hold the slab's interior nodes, the Dirichlet vector its
// boundary nodes, both in BlockIt order (contiguous pieces of the serial vectors, dimension 0 being outermost).
extern "C" int stokes_op_create_slab(int d, const int *dims, int lo, int hi, stokes_dim0_fn dim0, void *dim0_ctx, stokes_op **out) {
  if (!dim0) return chebhip_fail(CHEBHIP_ERR_ARG, "slab mode needs the dimension-0 callback");
  return st_create(d, dims, lo, hi, dim0, dim0_ctx, out);
}


int stokes_op_fd_view(stokes_op *op, chebhip::FdView *v) {
  if (!op || !v) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (op->slab) return chebhip_fail(CHEBHIP_ERR_ARG, "slab-mode handle: the preconditioner comes from chebhip_dist_stokes_pc");
  return stokes_op_fd_view_any(op, v, nullptr);
}
int stokes_op_fd_view_any(stokes_op *op, chebhip::FdView *v, int *gP0) {
  if (!op || !v) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (gP0) *gP0 = op->gP0;
  v->d = op->d; v->dims = op->dims.data(); v->N = op->N; v->G = op->I; v->ixL = op->ixL;
  v->eta = op->eta; v->deta = op->deta;                   // StokesPCSetUp0 reads eta only (stokes.C:1217-1222)
  return 0;
}

extern "C" long stokes_op_size(const stokes_op *op, int which) {
  if (!op) return -1;
  switch (which) {
    case 0: return op->N;
    case 1: return op->I;
    case 2: return op->I * op->d;
    case 3: return op->I;
    case 4: return op->I * (op->d + 1);
    case 5: return (op->N - op->I) * op->d;
    default: return -1;
  }
}

extern "C" int stokes_op_set_rheology(stokes_op *op, int kind, double hardness, double exponent, double eps, double gamma0) {
  if (!op) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL handle");
  if (kind != 0 && kind != 1) return chebhip_fail(CHEBHIP_ERR_ARG, "rheology %d not implemented (stokes.C:470-480)", kind);
  op->rh_kind = kind; op->rh_hard = hardness; op->rh_expo = exponent; op->rh_eps = eps; op->rh_g0 = gamma0;
  return 0;
}

extern "C" int stokes_op_set_dirichlet(stokes_op *op, const double *values) {
  if (!op || !values) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  const int d = op->d;
  std::vector<double> loc((size_t)op->N * d, 0.0);
  std::vector<int> ind(d, 0);
  long dd = 0;
  for (long l = 0; l < op->N; l++) {                      // ixDL order: boundary nodes in BlockIt order, d values each
    if (st_is_bdy(op, ind.data())) for (int k = 0; k < d; k++) loc[(size_t)k * op->N + l] = values[dd++];
    for (int j = d - 1; j >= 0; j--) { if (++ind[j] < op->dims[j]) break; ind[j] = 0; }
  }
  if (!op->dirloc) SHIPCHK(hipMalloc((void **)&op->dirloc, loc.size() * sizeof(double)));
  SHIPCHK(hipMemcpy(op->dirloc, loc.data(), loc.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int stokes_op_set_force(stokes_op *op, const double *force) {
  if (!op || !force) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  const size_t g = (size_t)op->I * (op->d + 1);
  if (!op->force) SHIPCHK(hipMalloc((void **)&op->force, (g ? g : 1) * sizeof(double)));
  SHIPCHK(hipMemcpy(op->force, force, g * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

// ---- building blocks ------------------------------------------------------------------------
// Diagnostic builds only (-DCHEBHIP_DIAG: `make diag` -> tools/libchebhip_diag.so, tools/stokes_ablate.py; the shipped library has
// neither the symbol nor the tests): launches of the 128^3 callbacks left out one at a time -- results are wrong, the timings give
// each launch's MARGINAL cost in the pipelined callback (a profiler serialises the launches and both streams).
// bit 0 gather, 1 x/y gradient, 2 fused z launch, 3 x/y divergence, 4 scatter, 5 pressure chain
#ifdef CHEBHIP_DIAG
static int g_st_ablate = 0;
extern "C" void chebhip_debug_stokes_ablate(int mask) { g_st_ablate = mask; }
#define ST_ABL(bit) (g_st_ablate & (1 << (bit)))
#else
#define ST_ABL(bit) 0
#endif
// DP[k] (scalar field) or DV[k] (d stacked fields: same lines, d times as many)
static int sweep_plain(stokes_op *op, bool vec, int k, const double *x, double *y, int out_mode, const double *acc,
                       double alpha, hipStream_t st, bool pext = false) {       // pext: the pressure matrices (stokes_op::matsP)
  if (op->slab && k == 0)       // lines along dimension 0 cross the slabs: transposes and the pencil sweep are the driver's
    return op->dim0(op->dim0_ctx, 0, vec ? op->d : 1, x, out_mode == OUT_ACC ? acc : nullptr, alpha, y, st);
  SweepParams sp = {};
  sp.ncols = vec ? op->ncolsV[k] : op->ncolsP[k];
  sp.inner = op->innerP[k];
  sp.in0 = x; sp.in_mode = IN_PLAIN; sp.out = y; sp.out_mode = out_mode; sp.acc = acc; sp.alpha = alpha;
  SHIPCHK(sweep_launch(pext ? op->matsP[op->dims[k]] : op->mats[op->dims[k]], sp, st));
  return 0;
}

#define ST_D(KERNEL, ...) do { if (d == 2) hipLaunchKernelGGL((KERNEL<2>), dim3(sgrid(op->N)), dim3(256), 0, st, op->N, __VA_ARGS__); \
                               else hipLaunchKernelGGL((KERNEL<3>), dim3(sgrid(op->N)), dim3(256), 0, st, op->N, __VA_ARGS__); } while (0)

// the grid of st_pair_ix: serial 3-D handles with an even last extent (option "general_kernels": the table)
static inline StGrid st_grid(const stokes_op *op) {
  StGrid g = {0u, 0u, 0u};
  if (op->d == 3 && !op->slab && (op->dims[2] & 1) == 0 && !opt(OPT_GENERAL_KERNELS)) { g.P0 = (unsigned)op->dims[0]; g.P1 = (unsigned)op->dims[1]; g.P2 = (unsigned)op->dims[2]; }
  return g;
}
// xL / pL <- global vector (either may be null)
static inline unsigned ugrid(long n, int un) { long g = (n + 256L * un - 1) / (256L * un); return (unsigned)(g < 1 ? 1 : g); }
static inline bool st_al16(const void *q) { return ((size_t)q & 15) == 0; }
static void st_local(stokes_op *op, int gs, int go, const double *src, const double *dirloc, double *xL, double *pL, hipStream_t st, long cs = 1) {
  const int d = op->d;
  // (k_st_local4 loads node 0 of src for boundary nodes too: not for a slab without unknowns, whose src may be NULL)
  if (d == 3 && gs == 4 && go == 3 && xL && pL && op->I > 0 && st_al16(src)) {
    if ((op->N & 1) == 0 && st_al16(xL) && st_al16(pL) && (!dirloc || st_al16(dirloc)))
      // (one pair per thread, as the scatter: 128^3 StokesMatMult 272.5 -> 271.1 us, StokesFunction 288.0 -> 285.5 us against two)
      hipLaunchKernelGGL((k_st_local4p<1>), dim3(ugrid(op->N >> 1, 1)), dim3(256), 0, st, op->N, (const int *)op->ixL, src, dirloc, xL, pL, st_grid(op));
    else
      hipLaunchKernelGGL((k_st_local4<4>), dim3(ugrid(op->N, 4)), dim3(256), 0, st, op->N, (const int *)op->ixL, src, dirloc, xL, pL);
    return;
  }
  ST_D(k_st_local, gs, go, (const int *)op->ixL, src, dirloc, xL, pL, cs);
}
// final scatter with every term present (StokesMatMult, StokesFunction)
// y0, y1, y2 (+ G): the velocity terms, summed in this order (the general path: yL, yLx[1], yLx[2]; y1, y2, G may be null)
static inline bool st_out_pairs(const stokes_op *op, const double *y0, const double *y1, const double *y2, const double *G, const double *force, const double *out) {
  return op->d == 3 && y1 && y2 && st_al16(out) && (!force || st_al16(force)) && (op->N & 1) == 0 && st_al16(y0) && st_al16(y1) && st_al16(y2) && (!G || st_al16(G));
}
static void st_out_full(stokes_op *op, const double *force, double *out, hipStream_t st, const double *y0 = nullptr, const double *y1 = nullptr,
                        const double *y2 = nullptr, const double *G = nullptr, const double *p3 = nullptr, bool no_gp = false) {      // p3: three stacked terms of the pressure rows instead of p2
  const int d = op->d;                                      // no_gp: grad p is inside the y terms (the folded pressure route: pair kernel only)
  if (p3) {                                                 // (the caller has checked st_out_pairs)
    // one pair per thread here too (round 6; rounds 4-5: two): at 64^3 the launch is 256 workgroups either way
#ifndef ST_OUT_P3_UN
#define ST_OUT_P3_UN 1
#endif
    hipLaunchKernelGGL((k_st_out4p<ST_OUT_P3_UN>), dim3(ugrid(op->N >> 1, ST_OUT_P3_UN)), dim3(256), 0, st, op->N, (const int *)op->ixL, y0, y1, y2,
                       (const double *)op->gp[0], (const double *)op->gp[1], (const double *)op->gp[2], p3, force, out, G, p3 + op->N, p3 + 2 * op->N, st_grid(op));
    return;
  }
  if (!y0) { y0 = op->yL; y1 = op->yLx[1]; y2 = op->yLx[2]; }
  if (d == 3 && y1 && y2 && st_al16(out) && (!force || st_al16(force))) {
    if ((op->N & 1) == 0 && st_al16(y0) && st_al16(y1) && st_al16(y2) && (!G || st_al16(G))) {     // (the handle's own arrays: always)
      // (one pair per thread: twice the waves in flight of the two-pair form for the same loads per CU -- 128^3 StokesMatMult 282.4 ->
      // 279.2 us, StokesFunction 296.1 -> 292.7 us in one process; four pairs per thread: 286.6 / 294.8)
      hipLaunchKernelGGL((k_st_out4p<1>), dim3(ugrid(op->N >> 1, 1)), dim3(256), 0, st, op->N, (const int *)op->ixL, y0, y1, y2,
                         (const double *)(no_gp ? nullptr : op->gp[0]), (const double *)op->gp[1], (const double *)op->gp[2], (const double *)op->p2, force, out, G,
                         (const double *)nullptr, (const double *)nullptr, st_grid(op));
    }
    else
      hipLaunchKernelGGL((k_st_out4<4>), dim3(ugrid(op->N, 4)), dim3(256), 0, st, op->N, (const int *)op->ixL, y0, y1, y2,
                         (const double *)op->gp[0], (const double *)op->gp[1], (const double *)op->gp[2], (const double *)op->p2, force, out, G);
    return;
  }
  ST_D(k_st_out, d + 1, (const int *)op->ixL, y0, y1, y2,
       (const double *)op->gp[0], (const double *)op->gp[1], (const double *)op->gp[2], (const double *)op->p2, d, force, out, G, 1L);
}

// d independent plain sweeps y[k] = alpha * D_k x[k] (DV: vec, d stacked fields; DP: scalar) as ONE launch where the
// kernels allow it (sweep_launch_multi), else one launch each.  Serial handles only.
static int st_flush_pending(stokes_op *op, hipStream_t st) {
  if (op->npend <= 0) return 0;
  const int n = op->npend; op->npend = 0;
  SHIPCHK(sweep_launch_multi(n, op->pend_m, op->pend_sp, st));
  return 0;
}
static int sweeps_multi(stokes_op *op, bool vec, int k0, const double *const *x, double *const *y, double alpha, hipStream_t st, bool spaced = false,
                        bool pext = false, int k1 = -1, bool park = false) {   // pext: the matrices with the end-point extrapolation of the pressure folded in; directions k0 .. k1 - 1; park: see stokes_op::npend
  const DiffMat *m[3]; SweepParams sp[3];
  int n = 0;
  if (k1 < 0) k1 = op->d;
  for (int k = k0; k < k1; k++, n++) {
    sp[n] = SweepParams{};
    sp[n].ncols = vec ? op->ncolsV[k] : op->ncolsP[k]; sp[n].inner = op->innerP[k];
    if (spaced && k > 0) {                                  // job k: its d fields are k N doubles further apart than dense stacking (stokes_op::T)
      sp[n].in_fblocks = (unsigned)(op->innerP[k] >= 16 ? op->N / ((long)op->dims[k] * op->innerP[k]) : op->N / op->dims[k]);
      sp[n].in_fskip = (unsigned)((long)k * op->N);
    }
    sp[n].in0 = x[k]; sp[n].in_mode = IN_PLAIN; sp[n].out = y[k]; sp[n].out_mode = OUT_STORE; sp[n].alpha = alpha;
    m[n] = pext ? &op->matsP[op->dims[k]] : &op->mats[op->dims[k]];
  }
  if (park && op->npend + n <= 8) { for (int j = 0; j < n; j++) { op->pend_m[op->npend] = m[j]; op->pend_sp[op->npend++] = sp[j]; } return 0; }
  SHIPCHK(sweep_launch_multi(n, m, sp, st));
  return 0;
}

// yL (+ yLx[1] + yLx[2]) = -sum_j DV[j] V[j]   (stokes.C:668-671, 737-740): one launch, the sum is taken by the final scatter
// in the order j = 0, 1, 2 (slab mode: the sweep along dimension 0 is the driver's, the others one launch)
static int st_div_stress(stokes_op *op, hipStream_t st, bool from_T = false) {
  double *y[3] = {op->yL, op->yLx[1], op->yLx[2]};
  if (op->slab) {
    // dimension 0 through the driver (pencils), the local directions as ONE launch into arrays of their own; the final scatter
    // adds the terms in the order j = 0, 1, 2 as on one GPU (the same bits).  (Round 3 ran an accumulating chain of d launches here:
    // at 8 ranks every launch of a 128^3 problem is a 64^3-sized, latency-bound one.)
    // (the local jobs are parked first: on a direct transport the driver's dimension-0 launch takes them along)
    const double *x[3] = {op->V[0], op->V[1], op->V[2]};
    int rc = sweeps_multi(op, true, 1, x, y, -1.0, st, false, false, -1, true); if (rc) return rc;
    rc = sweep_plain(op, true, 0, op->V[0], op->yL, OUT_STORE, nullptr, -1.0, st);
    const int rc2 = st_flush_pending(op, st);
    return rc ? rc : rc2;
  }
  if (from_T) {         // the fields (j,0), (j,1), (j,2) of direction j are slots j, 2j+1, 3j+2 of T: base j N, one field every (j+1) N
    const double *x[3] = {op->T, op->T + op->N, op->T + 2 * op->N};
    return sweeps_multi(op, true, 0, x, y, -1.0, st, true);
  }
  const double *x[3] = {op->V[0], op->V[1], op->V[2]};
  return sweeps_multi(op, true, 0, x, y, -1.0, st);
}

// V[j] = DV[j] xL (stokes.C:639) / strain[j] = DV[j] xL (:701)
static int st_gradient(stokes_op *op, double *const *out, hipStream_t st) {
  const double *x[3] = {op->xL, op->xL, op->xL};
  if (op->slab) {                                           // dimension 0 on pencils (the driver), the local directions in one launch
    int rc = sweeps_multi(op, true, 1, x, out, 1.0, st, false, false, -1, true); if (rc) return rc;
    rc = sweep_plain(op, true, 0, op->xL, out[0], OUT_STORE, nullptr, 1.0, st);
    const int rc2 = st_flush_pending(op, st);
    return rc ? rc : rc2;
  }
  return sweeps_multi(op, true, 0, x, out, 1.0, st);
}

// One stream (small grids, see st_create): the d gradient sweeps out[j] = DV[j] xL and the d pressure-gradient sweeps
// gp[i] = DP[i] pL of a callback are 2d independent plain sweeps of the same two local vectors -- ONE launch (64^3: one
// dependent launch less on a chain that is all launch latency).  Without op->pext, pL must hold its extrapolated boundary values.
static int st_gradient_and_pressure_gradient(stokes_op *op, double *const *out, hipStream_t st) {
  const int d = op->d;
  const DiffMat *m[6]; SweepParams sp[6];
  int n = 0;
  for (int pass = 0; pass < 2; pass++)
    for (int k = 0; k < d; k++, n++) {
      const bool vec = pass == 0;
      sp[n] = SweepParams{};
      sp[n].ncols = vec ? op->ncolsV[k] : op->ncolsP[k]; sp[n].inner = op->innerP[k];
      sp[n].in0 = vec ? op->xL : op->pL; sp[n].in_mode = IN_PLAIN; sp[n].out = vec ? out[k] : op->gp[k]; sp[n].out_mode = OUT_STORE; sp[n].alpha = 1.0;
      m[n] = (!vec && op->pext) ? &op->matsP[op->dims[k]] : &op->mats[op->dims[k]];
    }
  SHIPCHK(sweep_launch_multi(n, m, sp, st));
  return 0;
}
static inline bool st_one_launch_gradients(const stokes_op *op) { return !op->aux && !op->slab; }

static void st_pressure_extrapolate(stokes_op *op, double *pL, hipStream_t st);
// Slab mode: out[j] = DV[j] xL and gp[i] = DP[i] pL with ONE round trip to pencils for everything along dimension 0 (callback
// kind 2: d velocity fields differentiated with D, the pressure field with its end-point extrapolation) and ONE launch for the
// 2 (d - 1) local sweeps.  Round 3 made two round trips (gradient, pressure): 4 of the 6 exchanges of a callback, now 2 of 4.
static int st_gradient_and_pressure_gradient_slab(stokes_op *op, double *const *out, hipStream_t st) {
  const int d = op->d; const long N = op->N;
  if (!op->pext) st_pressure_extrapolate(op, op->pL, st);      // z and y lines (the x lines: on the pencils, stokes_op_pencil_pressure)
  op->gp[0] = out[0] + (size_t)d * N;                          // the (d+1)-th field of the array that receives the gradient
  // the local gradient and pressure-gradient sweeps: parked, so that on a direct transport the driver's dimension-0 launch takes them
  // along (all 2d sweeps of the callback's first half are then ONE launch); otherwise launched right after it
  op->npend = 0;
  for (int pass = 0; pass < 2; pass++)
    for (int k = 1; k < d; k++) {
      const bool vec = pass == 0;
      SweepParams &q = op->pend_sp[op->npend];
      q = SweepParams{};
      q.ncols = vec ? op->ncolsV[k] : op->ncolsP[k]; q.inner = op->innerP[k];
      q.in0 = vec ? op->xL : op->pL; q.in_mode = IN_PLAIN; q.out = vec ? out[k] : op->gp[k]; q.out_mode = OUT_STORE; q.alpha = 1.0;
      op->pend_m[op->npend++] = (!vec && op->pext) ? &op->matsP[op->dims[k]] : &op->mats[op->dims[k]];
    }
  const int rc = op->dim0(op->dim0_ctx, 2, d + 1, op->xL, nullptr, 1.0, out[0], st);
  const int rc2 = st_flush_pending(op, st);
  return rc ? rc : rc2;
}

// Uniform viscosity, eta' = 0 (linear rheology, stokes.C:470-474; the state after create): the viscous block of the Jacobian,
//   -sum_j D_j eta (D_j v_c + D_c v_j) / 2 = -eta/2 ( sum_j D_j D_j v_c + D_c div v ),   div v = sum_j D_j v_j
// (sweeps along different directions commute), needs no node loop and no second set of d^2 sweeps: ONE launch of the d
// second-derivative sweeps V[j] = -eta/2 (D D)_j xL (d fields each), the d sweeps of the trace t_j = D_j v_j and, for
// StokesMatMult, the d pressure-gradient sweeps; a pointwise sum div v = sum t_j (also the pressure rows); one launch of the
// d sweeps G_c = -eta/2 D_c div v.  The scatter adds V[0] + V[1] + V[2] + G (+ grad p).  15 field sweeps instead of 18 and a
// 3-array sum instead of the 27-array node loop.  Same result to rounding (tests: against the general path and the oracle).
__global__ void k_st_sum_fields(long N, int d, const double *__restrict__ t, double *__restrict__ out) {
  GS_LOOP(i, N) { double v = t[i] + t[N + i]; if (d == 3) v = v + t[2 * N + i]; out[i] = v; }
}
static inline bool st_uniform(const stokes_op *op) { return op->uniform_ok && op->eta_uniform && !op->deta_nonzero; }
static int st_viscous_uniform(stokes_op *op, bool with_pressure, hipStream_t st, const double *xloc = nullptr, double eta_value = 0.0, bool *split_div = nullptr) {
  const int d = op->d; const long N = op->N;
  if (!xloc) { xloc = op->xL; eta_value = op->eta_value; }             // StokesFunction: its own local vector (xF), eta = 1
  const double a = -0.5 * eta_value;
  const DiffMat *m[9]; SweepParams sp[9];
  int n = 0;
  auto job = [&](const DiffMat &mat, bool vec, int k, const double *in, double *out, double alpha) {
    sp[n] = SweepParams{};
    sp[n].ncols = vec ? op->ncolsV[k] : op->ncolsP[k]; sp[n].inner = op->innerP[k];
    sp[n].in0 = in; sp[n].in_mode = IN_PLAIN; sp[n].out = out; sp[n].out_mode = OUT_STORE; sp[n].alpha = alpha;
    m[n++] = &mat;
  };
  for (int k = 0; k < d; k++) job(op->matsDD[op->dims[k]], true, k, xloc, op->V[k], a);
  for (int k = 0; k < d; k++) job(op->mats[op->dims[k]], false, k, xloc + (size_t)k * N, op->yL + (size_t)k * N, 1.0);
  if (with_pressure) for (int k = 0; k < d; k++) job(op->matsP[op->dims[k]], false, k, op->pL, op->gp[k], 1.0);
  SHIPCHK(sweep_launch_multi(n, m, sp, st));
  // Lines of at most 64 points (the launch-bound sizes): the sweeps of grad div v read div v = (t_0 + t_1) + t_2 from its three
  // terms as they load (IN_SUM3: the same sum in the same order), so the pointwise pass between the two sweep launches -- one
  // more dependent launch on a chain that is all launch latency -- is not run; *split_div tells the caller that p2 was not formed.
  if (split_div) {
    *split_div = false;
    if (d == 3 && !opt(OPT_SEPARATE_LAUNCHES) && !opt(OPT_GENERAL_KERNELS)) {
      n = 0;
      for (int c = 0; c < d; c++) {
        job(op->mats[op->dims[c]], false, c, op->yL, op->yLx[1] + (size_t)c * N, a);
        sp[n - 1].in_mode = IN_SUM3; sp[n - 1].in1 = op->yL + N; sp[n - 1].in2 = op->yL + 2 * N;
      }
      bool done = false;
      SHIPCHK(sweep_launch_multi_try(n, m, sp, st, &done));
      if (done) { *split_div = true; return 0; }
    }
  }
  hipLaunchKernelGGL(k_st_sum_fields, dim3(sgrid(N)), dim3(256), 0, st, N, d, (const double *)op->yL, op->p2);
  n = 0;
  for (int c = 0; c < d; c++) job(op->mats[op->dims[c]], false, c, op->p2, op->yLx[1] + (size_t)c * N, a);
  SHIPCHK(sweep_launch_multi(n, m, sp, st));
  return 0;
}

// The strain state of a StokesFunction that took the uniform-viscosity route (stokes_op::xF), rebuilt on demand:
// strain[j] = DV[j] xF (stokes.C:701), symmetrised per node (:711-716, :722).  eta = 1, eta' = 0 were written by that call.
template <int D>
__global__ void k_st_symmetrise(long N, double *__restrict__ S0, double *__restrict__ S1, double *__restrict__ S2) {
  double *S[3] = {S0, S1, S2};
  GS_LOOP(i, N) {
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
      for (int k = j + 1; k < D; k++) { const double s = 0.5 * (S[j][k * N + i] + S[k][j * N + i]); S[j][k * N + i] = s; S[k][j * N + i] = s; }
  }
}
static int st_sync_strain(stokes_op *op, hipStream_t st) {
  if (!op->strain_stale) return 0;
  const double *x[3] = {op->xF, op->xF, op->xF};
  int rc = sweeps_multi(op, true, 0, x, op->strain, 1.0, st); if (rc) return rc;
  const int d = op->d;
  ST_D(k_st_symmetrise, op->strain[0], op->strain[1], op->strain[2]);
  SHIPCHK(hipGetLastError());
  op->strain_stale = false;
  return 0;
}

// viscous part of StokesMatMultVV on xL: V[j] = DV[j] xL, node loop, yL = -sum DV[j] V[j]; div (may be null)
// receives the trace of the gradient = StokesDivergence of the same xL.  have_gradient: V already holds DV[j] xL.
// The z direction in one launch (k_st_zfused16): d = 3 on one GPU, contiguous lines of 68 .. 128 points (KS = 16), P % 4 == 0.
// Option "stokes_z_separate" = 1 keeps the separate-pass route (A/B; the two give the same bits).
static bool st_zfused_ok(stokes_op *op) {
  if (op->d != 3 || op->slab || (op->N & 1) || opt(OPT_STOKES_Z_SEPARATE) || opt(OPT_GENERAL_KERNELS) || opt(OPT_SEPARATE_LAUNCHES)) return false;
  const int P = op->dims[2];
  // From 14 400 lines on (120^2): below, the tiles of 16 lines x 3 fields are too few to balance over 2 x 256 workgroups and the
  // launch merely ties with the three it replaces (one process, MatVV separate / fused: 80^3 92 / 94 us, 96^3 126 / 123, 112^3 175 / 175,
  // 128^3 254 / 230)
  return P > 64 && P <= 128 && (P % 4) == 0 && op->mats[P].KS == ZF_KS && op->mats[P].sym == 0 && op->N / P >= 14400 && op->N / P < 0x7fffffffL / ZF_NT;
}
static int st_zfused_launch(stokes_op *op, int mode, ZfParams zp, hipStream_t st) {
  zp.P = op->dims[2]; zp.H = zp.P / 2; zp.N = op->N; zp.nlines = (unsigned)(op->N / zp.P); zp.ntiles = (zp.nlines + ZF_NT - 1) / ZF_NT;
  zp.yz = op->yLx[2];
  const DiffMat &m = op->mats[zp.P];
  zp.fragE = m.fragE; zp.fragO = m.fragO;
  hipError_t cu_err; const int ncu = sweep_num_cus(&cu_err); SHIPCHK(cu_err);
  const unsigned grid = zp.ntiles < 2u * (unsigned)ncu ? zp.ntiles : 2u * (unsigned)ncu;      // two workgroups per CU
  if (zp.pL) {
    if (mode == 2) hipLaunchKernelGGL((k_st_zfused16<2, true>), dim3(grid), dim3(256), 0, st, zp);
    else if (mode == 1) hipLaunchKernelGGL((k_st_zfused16<1, true>), dim3(grid), dim3(256), 0, st, zp);
    else hipLaunchKernelGGL((k_st_zfused16<0, true>), dim3(grid), dim3(256), 0, st, zp);
  } else {
    if (mode == 2) hipLaunchKernelGGL((k_st_zfused16<2, false>), dim3(grid), dim3(256), 0, st, zp);
    else if (mode == 1) hipLaunchKernelGGL((k_st_zfused16<1, false>), dim3(grid), dim3(256), 0, st, zp);
    else hipLaunchKernelGGL((k_st_zfused16<0, false>), dim3(grid), dim3(256), 0, st, zp);
  }
  SHIPCHK(hipGetLastError());
  return 0;
}
// Round 5: the pressure inside the stress.  -sum_j D_j tau_jk + DP_k p = -(sum_{j != k} D_j tau_jk + D_k (tau_kk - p_ext)) with p_ext the
// pressure whose line-end values are extrapolated (StokesPressureReduceOrder, stokes.C:1029-1080; DP_k of stokes.C:609-614 is D_k on
// that field): k_st_pfaces fills the face-interior nodes of pL in one small launch, the fused z launch subtracts p from the diagonal
// stress as it leaves the node loop, and the three divergence sweeps deliver the velocity rows of MatVV + MatVP at once -- the three
// pressure-gradient sweeps (48 B/node, 25 us at 128^3) and their term in the scatter (24 B/node) are gone for one more 8-byte read.
// Same operator to rounding (sums in another order; observed <= 1e-14 against the separate route).  Where the fused-z route runs
// with 16-byte-aligned even grids (the pair scatter); option "stokes_pressure_sweeps" = 1 keeps the separate sweeps (A/B).
static bool st_fold_pressure(stokes_op *op, const double *out, const double *force) {
  // (the scatter without a grad p term is the pair kernel: a result or force vector that is only 8-byte aligned keeps the sweeps)
  return !op->aux && !op->slab && op->d == 3 && (op->N & 1) == 0 && !opt(OPT_STOKES_PRESSURE_SWEEPS) && st_zfused_ok(op) &&
         st_al16(out) && (!force || st_al16(force));
}
static int st_pressure_faces(stokes_op *op, hipStream_t st) {
  const int P0 = op->dims[0], P1 = op->dims[1], P2 = op->dims[2];
  long lines = (long)(P0 - 2) * (P1 - 2); if ((long)(P0 - 2) * (P2 - 2) > lines) lines = (long)(P0 - 2) * (P2 - 2); if ((long)(P1 - 2) * (P2 - 2) > lines) lines = (long)(P1 - 2) * (P2 - 2);
  long g = (lines + 3) / 4; if (g > 4096) g = 4096; if (g < 1) g = 1;
  hipLaunchKernelGGL(k_st_pfaces, dim3((unsigned)g, 3), dim3(256), 0, st, op->pL, P0, P1, P2, (const double *)op->w0[0], (const double *)op->w1[0],
                     (const double *)op->w0[1], (const double *)op->w1[1], (const double *)op->w0[2], (const double *)op->w1[2]);
  SHIPCHK(hipGetLastError());
  return 0;
}
// The x / y gradient of the fused-z route, out[0] = D_x xL, out[1] = D_y xL (3 fields each), and -- with_pressure -- the three
// pressure-gradient sweeps gp[i] = DP[i] pL as jobs of the SAME launch: five independent plain sweeps of the two local vectors.
// (The pressure chain on a second stream, rounds 2-4, costs the callback 32 us for 100 MB: tools/stokes_ablate.py,
// profiles/r05_stokes_ablate.txt -- fork, join and a launch that competes with the viscous chain for the memory system rather
// than filling idle CUs.)  Option "stokes_pressure_stream" = 1 (read when the handle is created) keeps the second stream (A/B).
static int st_xy_gradient(stokes_op *op, double *const *out, bool with_pressure, hipStream_t st) {
  const int d = op->d;
  const DiffMat *m[5]; SweepParams sp[5];
  int n = 0;
  for (int k = 0; k < 2; k++, n++) {
    sp[n] = SweepParams{};
    sp[n].ncols = op->ncolsV[k]; sp[n].inner = op->innerP[k];
    sp[n].in0 = op->xL; sp[n].in_mode = IN_PLAIN; sp[n].out = out[k]; sp[n].out_mode = OUT_STORE; sp[n].alpha = 1.0;
    m[n] = &op->mats[op->dims[k]];
  }
  if (with_pressure)
    for (int k = 0; k < d; k++, n++) {
      sp[n] = SweepParams{};
      sp[n].ncols = op->ncolsP[k]; sp[n].inner = op->innerP[k];
      sp[n].in0 = op->pL; sp[n].in_mode = IN_PLAIN; sp[n].out = op->gp[k]; sp[n].out_mode = OUT_STORE; sp[n].alpha = 1.0;
      m[n] = op->pext ? &op->matsP[op->dims[k]] : &op->mats[op->dims[k]];
    }
  SHIPCHK(sweep_launch_multi(n, m, sp, st));
  return 0;
}
static int st_viscous_jacobian_zfused(stokes_op *op, double *div, hipStream_t st, bool with_pressure = false, bool fold = false) {
  int rc = ST_ABL(1) ? 0 : st_xy_gradient(op, op->V, with_pressure && !fold, st); if (rc) return rc;   // V[0] = D_x xL, V[1] = D_y xL (+ gp[])
  if (fold && (rc = st_pressure_faces(op, st))) return rc;
  ZfParams zp = {};
  zp.pL = fold ? op->pL : nullptr;
  zp.xL = op->xL; zp.Vx = op->V[0]; zp.Vy = op->V[1];
  zp.S0 = op->strain[0]; zp.S1 = op->strain[1]; zp.S2 = op->strain[2]; zp.eta = op->eta; zp.deta = op->deta;
  zp.div = div;
  if (!ST_ABL(2) && (rc = st_zfused_launch(op, op->deta_nonzero ? 1 : 0, zp, st))) return rc;
  double *y[3] = {op->yL, op->yLx[1], op->yLx[2]};
  const double *t[3] = {op->V[0], op->V[1], op->V[2]};
  if (ST_ABL(3)) return 0;
  return sweeps_multi(op, true, 0, t, y, -1.0, st, false, false, 2);                                     // yL = -D_x tau_x., yLx[1] = -D_y tau_y.
}
static int st_viscous_jacobian(stokes_op *op, double *div, hipStream_t st, bool have_gradient = false) {
  const int d = op->d;
  if (!have_gradient && st_zfused_ok(op)) return st_viscous_jacobian_zfused(op, div, st);
  if (!have_gradient) { int rc = st_gradient(op, op->V, st); if (rc) return rc; }                                               // :639
#define NODE_VV(D_, DETA_) hipLaunchKernelGGL((k_st_node_vv<D_, DETA_>), dim3(sgrid(op->N)), dim3(256), 0, st, op->N, op->V[0], op->V[1], op->V[2], \
    (const double *)op->strain[0], (const double *)op->strain[1], (const double *)op->strain[2], (const double *)op->eta, (const double *)op->deta, div)
#define NODE_VV_PAIR(DETA_, SYM_) hipLaunchKernelGGL((k_st_node_vv_pair<DETA_, SYM_>), dim3(sgrid(op->N >> 1)), dim3(256), 0, st, op->N, op->V[0], op->V[1], op->V[2], \
    (const double *)op->strain[0], (const double *)op->strain[1], (const double *)op->strain[2], (const double *)op->eta, (const double *)op->deta, div, op->T)
  if (d == 2) { if (op->deta_nonzero) NODE_VV(2, true); else NODE_VV(2, false); }
  else if ((op->N & 1) == 0) { if (op->deta_nonzero) NODE_VV_PAIR(true, false); else NODE_VV_PAIR(false, false); }
  else        { if (op->deta_nonzero) NODE_VV(3, true); else NODE_VV(3, false); }
#undef NODE_VV_PAIR
#undef NODE_VV
  return st_div_stress(op, st);
}

// p2 = sum_i DP[i] (component i of xL)   (StokesDivergence, stokes.C:583-591); a component is a contiguous field
static int st_divergence(stokes_op *op, hipStream_t st) {
  for (int i = 0; i < op->d; i++) {
    int rc = sweep_plain(op, false, i, op->xL + (size_t)i * op->N, op->p2, i == 0 ? OUT_STORE : OUT_ACC, op->p2, 1.0, st);
    if (rc) return rc;
  }
  return 0;
}

// boundary values of pL by extrapolation from the interior (StokesPressureReduceOrder, stokes.C:1029-1080); in slab
// mode the x lines are left to the driver (see st_pressure_gradient)
static void st_pressure_extrapolate(stokes_op *op, double *pL, hipStream_t st) {
  const int d = op->d;
  const long m = op->dims[0], n = op->dims[1], p = (d == 2) ? 1 : op->dims[2];
  const long i_lo = (op->lo == 0) ? 1 : 0, ni = m - i_lo;
  if (p > 1 && ni > 0) {
    long g = (ni * (n - 1) + 3) / 4; if (g > 8192) g = 8192; if (g < 1) g = 1;
    hipLaunchKernelGGL(k_st_preduce_contig, dim3((unsigned)g), dim3(256), 0, st, pL, ni, i_lo, n * p, n - 1, 1L, p,
                       (int)p, (const double *)op->w0[2], (const double *)op->w1[2]);
  }
  if (ni > 0)
    hipLaunchKernelGGL(k_st_preduce, dim3(pgrid(ni * p)), dim3(256), 0, st, pL, ni, i_lo, n * p, p, 0L, 1L, p,
                       (int)n, (const double *)op->w0[1], (const double *)op->w1[1]);
  if (!op->slab)
    hipLaunchKernelGGL(k_st_preduce, dim3(pgrid(n * p)), dim3(256), 0, st, pL, n, 0L, p, p, 0L, 1L, n * p,
                       (int)m, (const double *)op->w0[0], (const double *)op->w1[0]);
}

// pL (interior filled, boundary zero) -> boundary extrapolation -> gp[i] = DP[i] pL   (stokes.C:609-614)
static int st_pressure_gradient(stokes_op *op, hipStream_t st) {
  // Lines of at most 256 points: gp[i] at an interior node needs the end values of ITS line along i only, and those are a
  // linear functional of the line's interior values (the other directions' extrapolations touch boundary lines, whose
  // gradients the scatter never reads): the extrapolation is part of the matrix (diffmat_create_pext) and the three passes
  // of StokesPressureReduceOrder are not run at all.  64^3 StokesMatMult: 62.7 -> 47.8 us.
  if (!op->pext) st_pressure_extrapolate(op, op->pL, st);      // z lines, y lines (and, on one GPU, x lines): stokes.C:1043-1074
  if (op->slab) {
    // x lines cross the slabs: their extrapolation (stokes.C:1064-1074) and DP[0] happen on pencils, in the driver
    // (stokes_op_pencil_pressure).  Without pext, the end planes of pL the x pass would have filled only feed DP[1], DP[2]
    // on those planes, which the final scatter never reads.
    int rc = op->dim0(op->dim0_ctx, 1, 1, op->pL, nullptr, 1.0, op->gp[0], st); if (rc) return rc;
    const double *x[3] = {op->pL, op->pL, op->pL};
    return sweeps_multi(op, false, 1, x, op->gp, 1.0, st, false, op->pext);
  }
  const double *x[3] = {op->pL, op->pL, op->pL};
  return sweeps_multi(op, false, 0, x, op->gp, 1.0, st, false, op->pext);
}

// pressure chain on the second stream, between the gather (already enqueued on st) and the final scatter
static int st_pressure_gradient_forked(stokes_op *op, hipStream_t st) {
  if (!op->aux) return st_pressure_gradient(op, st);
  SHIPCHK(hipEventRecord(op->ev_fork, st));
  SHIPCHK(hipStreamWaitEvent(op->aux, op->ev_fork, 0));
  return st_pressure_gradient(op, op->aux);
}
static int st_join(stokes_op *op, hipStream_t st) {
  if (!op->aux) return 0;
  SHIPCHK(hipEventRecord(op->ev_join, op->aux));
  SHIPCHK(hipStreamWaitEvent(st, op->ev_join, 0));
  return 0;
}

#define ST_OUT(...) ST_D(k_st_out, __VA_ARGS__)
#define ARGCHK(c) do { if (!(c)) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument"); } while (0)
// a vector of the handle may be NULL when it is empty (a slab that owns only boundary planes has no unknowns)
#define VEC_OK(ptr) ((ptr) != nullptr || op->I == 0)
#define CDP(x) ((const double *)(x))

// The three blocks on their own.  cm: the velocity vector is component-major (component c of interior node n at c I + n) instead
// of the reference's node-major layout (n d + c): the layout of the block preconditioners' inner Krylov solves (saddle.hip), whose
// vectors are then d stacked scalar fields for MatVVPC's line transforms as well -- no (de)interleaving pass per iteration.
// component-major d = 3 vectors on node pairs (k_st_local_cm3p / k_st_out_cm3p): N even, the handle's arrays are 16-B aligned
static inline bool st_cm_pairs(const stokes_op *op, bool cm) { return cm && op->d == 3 && (op->N & 1) == 0 && op->I > 0; }
static void st_local_cm(stokes_op *op, const double *v_cm, hipStream_t st) {
  hipLaunchKernelGGL((k_st_local_cm3p<1>), dim3(ugrid(op->N >> 1, 1)), dim3(256), 0, st, op->N, op->I, (const int *)op->ixL, v_cm, op->xL, st_grid(op));
}
static void st_out_cm(stokes_op *op, int nterms, const double *t0, const double *t1, const double *t2, const double *t3, double *out_cm, hipStream_t st,
                      bool separate_fields = false) {      // separate_fields: t0, t1, t2 are the three component fields of ONE term (gp[])
  StTerms3 tm = {};
  const long N = op->N;
  if (separate_fields) { tm.n = 1; tm.p[0][0] = t0; tm.p[0][1] = t1; tm.p[0][2] = t2; }
  else {
    const double *t[4] = {t0, t1, t2, t3};
    tm.n = nterms;
    for (int q = 0; q < nterms; q++) for (int c = 0; c < 3; c++) tm.p[q][c] = t[q] + (size_t)c * N;
  }
  hipLaunchKernelGGL((k_st_out_cm3p<2>), dim3(ugrid(op->N >> 1, 2)), dim3(256), 0, st, op->N, op->I, (const int *)op->ixL, tm, out_cm, st_grid(op));
}

static int st_mult_vv(stokes_op *op, const double *vG, double *out, hipStream_t st, bool cm) {
  const int d = op->d, gs = cm ? 1 : d; const long cs = cm ? op->I : 1;
  const bool pairs = st_cm_pairs(op, cm);
  if (pairs) st_local_cm(op, vG, st); else
  st_local(op, gs, 0, vG, nullptr, op->xL, nullptr, st, cs);
  if (st_uniform(op)) {
    bool split = false;                                    // (no pressure rows here: whether div v was formed as p2 does not matter)
    int rc = st_viscous_uniform(op, false, st, nullptr, 0.0, &split); if (rc) return rc;
    if (pairs) { st_out_cm(op, 4, op->V[0], op->V[1], op->V[2], op->yLx[1], out, st); SHIPCHK(hipGetLastError()); return 0; }
    ST_OUT(gs, (const int *)op->ixL, CDP(op->V[0]), CDP(op->V[1]), CDP(op->V[2]), CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(nullptr), 0, CDP(nullptr), out, CDP(op->yLx[1]), cs);
    SHIPCHK(hipGetLastError());
    return 0;
  }
  int rc = st_viscous_jacobian(op, nullptr, st); if (rc) return rc;
  if (pairs) { st_out_cm(op, 3, op->yL, op->yLx[1], op->yLx[2], nullptr, out, st); SHIPCHK(hipGetLastError()); return 0; }
  ST_OUT(gs, (const int *)op->ixL, CDP(op->yL), CDP(op->yLx[1]), CDP(op->yLx[2]), CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(nullptr), 0, CDP(nullptr), out, CDP(nullptr), cs);
  SHIPCHK(hipGetLastError());
  return 0;
}
static int st_mult_pv(stokes_op *op, const double *vG, double *pout, hipStream_t st, bool cm) {
  const int d = op->d;
  if (st_cm_pairs(op, cm)) st_local_cm(op, vG, st); else
  st_local(op, cm ? 1 : d, 0, vG, nullptr, op->xL, nullptr, st, cm ? op->I : 1);
  if (!op->slab && d >= 2) {
    // the d terms D_i v_i as ONE launch of d jobs into arrays of their own (yL is free here), summed by the scatter in the order of
    // the accumulating chain, (t_0 + t_1) + t_2: the same bits, two dependent launches less (128^3: 77 -> 57 us)
    const DiffMat *m[3]; SweepParams sp[3];
    for (int k = 0; k < d; k++) {
      sp[k] = SweepParams{};
      sp[k].ncols = op->ncolsP[k]; sp[k].inner = op->innerP[k];
      sp[k].in0 = op->xL + (size_t)k * op->N; sp[k].in_mode = IN_PLAIN; sp[k].out = op->yL + (size_t)k * op->N; sp[k].out_mode = OUT_STORE; sp[k].alpha = 1.0;
      m[k] = &op->mats[op->dims[k]];
    }
    bool done = false;
    SHIPCHK(sweep_launch_multi_try(d, m, sp, st, &done));
    if (done) {
      ST_OUT(1, (const int *)op->ixL, CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(op->yL), 0, CDP(nullptr), pout, CDP(nullptr), 1L,
             CDP(op->yL + op->N), CDP(d == 3 ? op->yL + 2 * op->N : nullptr));
      SHIPCHK(hipGetLastError());
      return 0;
    }
  }
  int rc = st_divergence(op, st); if (rc) return rc;
  ST_OUT(1, (const int *)op->ixL, CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(op->p2), 0, CDP(nullptr), pout, CDP(nullptr), 1L);
  SHIPCHK(hipGetLastError());
  return 0;
}
static int st_mult_vp(stokes_op *op, const double *pG, double *vout, hipStream_t st, bool cm) {
  const int d = op->d;
  st_local(op, 1, 0, pG, nullptr, nullptr, op->pL, st);
  if (op->slab) op->gp[0] = op->V[0] + (size_t)op->d * op->N;
  int rc = st_pressure_gradient(op, st); if (rc) return rc;
  if (st_cm_pairs(op, cm)) { st_out_cm(op, 1, op->gp[0], op->gp[1], op->gp[2], nullptr, vout, st, true); SHIPCHK(hipGetLastError()); return 0; }
  ST_OUT(cm ? 1 : d, (const int *)op->ixL, CDP(nullptr), CDP(nullptr), CDP(nullptr), CDP(op->gp[0]), CDP(op->gp[1]), CDP(op->gp[2]), CDP(nullptr), 0, CDP(nullptr), vout, CDP(nullptr),
         cm ? op->I : 1L);
  SHIPCHK(hipGetLastError());
  return 0;
}

extern "C" int stokes_op_mult_vv(stokes_op *op, const double *vG, double *out, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(vG) && VEC_OK(out));
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_MULT_VV, stream);
  return st_mult_vv(op, vG, out, (hipStream_t)stream, false);
}
extern "C" int stokes_op_mult_pv(stokes_op *op, const double *vG, double *pout, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(vG) && VEC_OK(pout));
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_MULT_PV, stream);
  return st_mult_pv(op, vG, pout, (hipStream_t)stream, false);
}
extern "C" int stokes_op_mult_vp(stokes_op *op, const double *pG, double *vout, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(pG) && VEC_OK(vout));
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_MULT_VP, stream);
  return st_mult_vp(op, pG, vout, (hipStream_t)stream, false);
}
// ... on component-major velocity vectors (see st_mult_vv)
extern "C" int stokes_op_mult_vv_cm(stokes_op *op, const double *v_cm, double *out_cm, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(v_cm) && VEC_OK(out_cm));
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_MULT_VV, stream);
  return st_mult_vv(op, v_cm, out_cm, (hipStream_t)stream, true);
}
extern "C" int stokes_op_mult_pv_cm(stokes_op *op, const double *v_cm, double *pout, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(v_cm) && VEC_OK(pout));
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_MULT_PV, stream);
  return st_mult_pv(op, v_cm, pout, (hipStream_t)stream, true);
}
extern "C" int stokes_op_mult_vp_cm(stokes_op *op, const double *pG, double *vout_cm, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(pG) && VEC_OK(vout_cm));
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_MULT_VP, stream);
  return st_mult_vp(op, pG, vout_cm, (hipStream_t)stream, true);
}

extern "C" int stokes_op_mult(stokes_op *op, const double *xG, double *yG, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(xG) && VEC_OK(yG));
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_MULT, stream);
  hipStream_t st = (hipStream_t)stream;
  const int d = op->d;
  // scatterGV + scatterVL (zero boundary) and scatterGP (:505-510) in one pass over xG; the same xL serves
  // MatVV (:508) and MatPV (:509), whose result is the trace written by the node loop
  if (!ST_ABL(0)) st_local(op, d + 1, d, xG, nullptr, op->xL, op->pL, st);
  int rc;
  if (st_uniform(op)) {
    bool split = false;
    const bool pairs = st_out_pairs(op, op->V[0], op->V[1], op->V[2], op->yLx[1], nullptr, yG);
    if ((rc = st_viscous_uniform(op, true, st, nullptr, 0.0, pairs ? &split : nullptr))) return rc;
    st_out_full(op, nullptr, yG, st, op->V[0], op->V[1], op->V[2], op->yLx[1], split ? op->yL : nullptr);
    SHIPCHK(hipGetLastError());
    return 0;
  }
  if (op->slab) {
    if ((rc = st_gradient_and_pressure_gradient_slab(op, op->V, st))) return rc;                                                 // MatVP (:512) + :639
    if ((rc = st_viscous_jacobian(op, op->p2, st, true))) return rc;
  } else if (st_fold_pressure(op, yG, nullptr)) {             // the fused-z route with the pressure inside the stress
    if ((rc = st_viscous_jacobian_zfused(op, op->p2, st, false, true))) return rc;                                               // MatVP (:512) + MatVV
    if (!ST_ABL(4)) st_out_full(op, nullptr, yG, st, nullptr, nullptr, nullptr, nullptr, nullptr, true);
    SHIPCHK(hipGetLastError());
    return 0;
  } else if (!op->aux && st_zfused_ok(op)) {                  // the fused-z route with the pressure-gradient sweeps inside its first launch
    if (!op->pext) st_pressure_extrapolate(op, op->pL, st);
    if ((rc = st_viscous_jacobian_zfused(op, op->p2, st, !ST_ABL(5)))) return rc;                                                // MatVP (:512) + MatVV
  } else if (st_one_launch_gradients(op)) {
    if (!op->pext) st_pressure_extrapolate(op, op->pL, st);
    if ((rc = st_gradient_and_pressure_gradient(op, op->V, st))) return rc;                                                      // MatVP (:512) + :639
    if ((rc = st_viscous_jacobian(op, op->p2, st, true))) return rc;
  } else {
    if (!ST_ABL(5) && (rc = st_pressure_gradient_forked(op, st))) return rc;                                                     // MatVP (:512)
    if ((rc = st_viscous_jacobian(op, op->p2, st))) return rc;
  }
  if (!ST_ABL(5) && (rc = st_join(op, st))) return rc;
  if (!ST_ABL(4)) st_out_full(op, nullptr, yG, st);
  SHIPCHK(hipGetLastError());
  return 0;
}

extern "C" int stokes_op_function(stokes_op *op, const double *xG, double *yG, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(xG) && VEC_OK(yG));
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_FUNCTION, stream);
  hipStream_t st = (hipStream_t)stream;
  const int d = op->d;
  if (op->rh_kind == 0 && op->uniform_ok) {
    // Linear rheology (stokes.C:1920-1926: eta = 1, eta' = 0): the residual's viscous part is -1/2 (sum_j D_j D_j v + grad div v)
    // of the velocity WITH its Dirichlet values (sweeps along different directions commute on the full grid), so the node
    // loop and the second set of d^2 sweeps are not needed (st_viscous_uniform); the symmetrised strain it would leave as
    // state is rebuilt from xF by whoever asks for it (st_sync_strain).  Option "general_viscous" keeps the general route.
    st_local(op, d + 1, d, xG, op->dirloc, op->xF, op->pL, st);
    if (!(op->eta_uniform && op->eta_value == 1.0)) hipLaunchKernelGGL(k_st_fill, dim3(sgrid(op->N)), dim3(256), 0, st, op->N, 1.0, op->eta);
    if (op->deta_nonzero) hipLaunchKernelGGL(k_st_fill, dim3(sgrid(op->N)), dim3(256), 0, st, op->N, 0.0, op->deta);
    op->eta_uniform = true; op->eta_value = 1.0; op->deta_nonzero = false; op->strain_stale = true;
    bool split = false;
    const bool pairs = st_out_pairs(op, op->V[0], op->V[1], op->V[2], op->yLx[1], op->force, yG);
    int rc = st_viscous_uniform(op, true, st, op->xF, 1.0, pairs ? &split : nullptr); if (rc) return rc;
    st_out_full(op, op->force, yG, st, op->V[0], op->V[1], op->V[2], op->yLx[1], split ? op->yL : nullptr);                        // :750-756
    SHIPCHK(hipGetLastError());
    return 0;
  }
  op->strain_stale = false;                               // the node loop below leaves the strain as state
  // xL = velocity with Dirichlet values (stokes.C:691-699); it also feeds StokesDivergence(withDirichlet) (:746)
  if (!ST_ABL(0)) st_local(op, d + 1, d, xG, op->dirloc, op->xL, op->pL, st);
  if (!op->slab && op->sym && st_zfused_ok(op)) {
    // the z direction in one launch (k_st_zfused16, MODE 2): gradient along x, y -> strain[0], strain[1]; the fused launch leaves eta,
    // eta', the symmetrised strain (upper triangle) and the stress slots the x / y divergence reads, and returns -D_z tau_z. in yLx[2]
    const bool fold = st_fold_pressure(op, yG, op->force);
    if (op->aux && !ST_ABL(5)) { int rc = st_pressure_gradient_forked(op, st); if (rc) return rc; }                             // :747 (second stream: A/B)
    if (!op->aux && !op->pext && !fold) st_pressure_extrapolate(op, op->pL, st);
    int rc = ST_ABL(1) ? 0 : st_xy_gradient(op, op->strain, !op->aux && !ST_ABL(5) && !fold, st); if (rc) return rc;             // :701 (x, y), :747
    if (fold && (rc = st_pressure_faces(op, st))) return rc;
    ZfParams zp = {};
    zp.xL = op->xL; zp.Vx = op->strain[0]; zp.Vy = op->strain[1]; zp.Sz = op->strain[2];
    zp.eta_w = op->eta; zp.deta_w = op->deta; zp.T = op->T; zp.div = op->p2;
    zp.kind = op->rh_kind; zp.hardness = op->rh_hard; zp.expo = op->rh_expo; zp.eps = op->rh_eps; zp.gamma0 = op->rh_g0;
    zp.pL = fold ? op->pL : nullptr;
    if (!ST_ABL(2) && (rc = st_zfused_launch(op, 2, zp, st))) return rc;
    op->deta_nonzero = (op->rh_kind == 1);
    op->eta_uniform = (op->rh_kind == 0); op->eta_value = 1.0;
    double *y[3] = {op->yL, op->yLx[1], op->yLx[2]};
    const double *t[3] = {op->T, op->T + op->N, op->T + 2 * op->N};
    if (!ST_ABL(3) && (rc = sweeps_multi(op, true, 0, t, y, -1.0, st, true, false, 2))) return rc;                               // :737-740 (x, y)
    if (!ST_ABL(5) && (rc = st_join(op, st))) return rc;
    if (!ST_ABL(4)) st_out_full(op, op->force, yG, st, nullptr, nullptr, nullptr, nullptr, nullptr, fold);                          // :750-756
    SHIPCHK(hipGetLastError());
    return 0;
  }
  if (op->slab) {
    int rc = st_gradient_and_pressure_gradient_slab(op, op->strain, st); if (rc) return rc;                                      // :747, :701
  } else if (st_one_launch_gradients(op)) {
    if (!op->pext) st_pressure_extrapolate(op, op->pL, st);
    int rc = st_gradient_and_pressure_gradient(op, op->strain, st); if (rc) return rc;                                           // :747, :701
  } else {
    { int rc = st_pressure_gradient_forked(op, st); if (rc) return rc; }                                                         // :747
    { int rc = st_gradient(op, op->strain, st); if (rc) return rc; }                                                              // :701
  }
  if (op->sym)
    hipLaunchKernelGGL((k_st_node_fn_pair<true>), dim3(sgrid(op->N >> 1)), dim3(256), 0, st, op->N, op->strain[0], op->strain[1], op->strain[2],
                       op->V[0], op->V[1], op->V[2], op->eta, op->deta, op->p2, op->rh_kind, op->rh_hard, op->rh_expo, op->rh_eps, op->rh_g0, op->T);
  else if (d == 3 && (op->N & 1) == 0)
    hipLaunchKernelGGL((k_st_node_fn_pair<false>), dim3(sgrid(op->N >> 1)), dim3(256), 0, st, op->N, op->strain[0], op->strain[1], op->strain[2],
                       op->V[0], op->V[1], op->V[2], op->eta, op->deta, op->p2, op->rh_kind, op->rh_hard, op->rh_expo, op->rh_eps, op->rh_g0, op->T);
  else
    ST_D(k_st_node_fn, op->strain[0], op->strain[1], op->strain[2], op->V[0], op->V[1], op->V[2], op->eta, op->deta, op->p2,
         op->rh_kind, op->rh_hard, op->rh_expo, op->rh_eps, op->rh_g0);
  op->deta_nonzero = (op->rh_kind == 1);
  op->eta_uniform = (op->rh_kind == 0); op->eta_value = 1.0;                      // linear rheology: eta = 1, eta' = 0 (k_st_node_fn)
  int rc = st_div_stress(op, st, op->sym); if (rc) return rc;                                                                    // :737-740
  if ((rc = st_join(op, st))) return rc;
  st_out_full(op, op->force, yG, st);                                                                                              // :750-756
  SHIPCHK(hipGetLastError());
  return 0;
}

// ---- StokesMatMultSchur (stokes.C:523-535): y = -PV * solve(VV, VP x) ------------------------------
__global__ void k_st_neg(long n, double *__restrict__ a) { GS_LOOP(i, n) a[i] = -1.0 * a[i]; }   // VecScale(yG,-1), :533

static int st_vv_apply(void *ctx, const double *x, double *y, void *stream) { return stokes_op_mult_vv((stokes_op *)ctx, x, y, stream); }

extern "C" int stokes_op_set_inner_solver(stokes_op *op, int restart, double rtol, double atol, int max_it) {
  ARGCHK(op);
  if (restart < 1 || restart > 256 || !(rtol >= 0.0) || !(atol >= 0.0) || max_it < 0)
    return chebhip_fail(CHEBHIP_ERR_ARG, "restart must be in 1..256, tolerances and max_it non-negative");
  if (op->inner && restart != op->in_restart) { chebhip_fgmres_destroy(op->inner); op->inner = nullptr; }
  op->in_restart = restart; op->in_rtol = rtol; op->in_atol = atol; op->in_maxit = max_it;
  return 0;
}

extern "C" int stokes_op_inner_iterations(const stokes_op *op) { return op ? op->inner_its : -1; }

// Slab mode: the velocity vectors of the built-in inner solve are distributed; see chebhip_fgmres_set_reduce.
extern "C" int stokes_op_set_inner_reduce(stokes_op *op, chebhip_reduce_fn reduce, void *ctx) {
  ARGCHK(op);
  op->in_reduce = reduce; op->in_reduce_ctx = ctx;
  return 0;
}

static int st_mult_schur(stokes_op *op, const double *pG, double *out, chebhip_apply_fn solve, void *solve_ctx, void *stream, bool cm) {
  chebhip::StageTimer tm(CHEBHIP_STAGE_STOKES_SCHUR, stream);
  hipStream_t st = (hipStream_t)stream;
  const size_t gv = (size_t)op->I * op->d;
  if (!op->sv0) {
    int rc = st_alloc(&op->sv0, gv ? gv : 1); if (rc) return rc; if ((rc = st_alloc(&op->sv1, gv ? gv : 1))) return rc;
    SHIPCHK(hipStreamSynchronize(nullptr));       // st_alloc clears on the null stream, which a non-blocking caller's stream does not wait for
  }
  int rc = st_mult_vp(op, pG, op->sv0, st, cm); if (rc) return rc;                            // :530
  if (solve) { if ((rc = solve(solve_ctx, op->sv0, op->sv1, st))) return rc; }               // KSPSolve(KSPSchurVelocity), :531
  else {
    if (cm) return chebhip_fail(CHEBHIP_ERR_ARG, "the built-in inner solver works on node-major vectors");
    if (!op->inner) { if ((rc = chebhip_fgmres_create((long)gv, op->in_restart, &op->inner))) return rc; }
    if ((rc = chebhip_fgmres_set_tolerances(op->inner, op->in_rtol, op->in_atol, op->in_maxit))) return rc;
    if ((rc = chebhip_fgmres_set_reduce(op->inner, op->in_reduce, op->in_reduce_ctx))) return rc;
    if ((rc = chebhip_fgmres_solve(op->inner, st_vv_apply, op, nullptr, nullptr, op->sv0, op->sv1, 0, st))) return rc;
    op->inner_its = chebhip_fgmres_iterations(op->inner);
  }
  if ((rc = st_mult_pv(op, op->sv1, out, st, cm))) return rc;                                // :532
  hipLaunchKernelGGL(k_st_neg, dim3(sgrid(op->I)), dim3(256), 0, st, op->I, out);
  SHIPCHK(hipGetLastError());
  return 0;
}
extern "C" int stokes_op_mult_schur(stokes_op *op, const double *pG, double *out, chebhip_apply_fn solve, void *solve_ctx, void *stream) {
  ARGCHK(op); ARGCHK(VEC_OK(pG) && VEC_OK(out));
  return st_mult_schur(op, pG, out, solve, solve_ctx, stream, false);
}
// ... with the inner velocity solve on component-major vectors: `solve` (required) receives and returns them in that layout
extern "C" int stokes_op_mult_schur_cm(stokes_op *op, const double *pG, double *out, chebhip_apply_fn solve_cm, void *solve_ctx, void *stream) {
  ARGCHK(op); ARGCHK(solve_cm); ARGCHK(VEC_OK(pG) && VEC_OK(out));
  return st_mult_schur(op, pG, out, solve_cm, solve_ctx, stream, true);
}

// ---- pencil side of the slab mode: arrays (nfields, gP0, ncol), lines along dimension 0 with stride ncol ---------
// out = DV[0] / DP[0] applied to nfields stacked pencil fields
extern "C" int stokes_op_pencil_sweep(stokes_op *op, int nfields, long ncol, const double *in, double *out, void *stream) {
  ARGCHK(op && in && out);
  if (nfields < 1 || ncol < 0) return chebhip_fail(CHEBHIP_ERR_ARG, "bad pencil geometry");
  if (ncol == 0) return 0;
  SweepParams sp = {};
  sp.ncols = (unsigned)(nfields * ncol); sp.inner = (unsigned)ncol;
  sp.in0 = in; sp.in_mode = IN_PLAIN; sp.out = out; sp.out_mode = OUT_STORE; sp.alpha = 1.0;
  SHIPCHK(sweep_launch(op->mats[op->gP0], sp, (hipStream_t)stream));
  return 0;
}

// x-line pressure extrapolation (stokes.C:1064-1074) in place on a pencil, then gp0 = DP[0] p
extern "C" int stokes_op_pencil_pressure(stokes_op *op, long ncol, double *p_pencil, double *gp0_pencil, void *stream);
// Both of a slab callback's pencil sweeps (kind 2 of the dimension-0 callback) as two jobs of ONE launch: at 8 ranks a 128^3 callback
// is a dozen latency-bound launches of 5-12 us, and these two were separate only because their matrices differ (D, and D with the
// end-point extrapolation folded in).
extern "C" int stokes_op_pencil_sweep_pressure(stokes_op *op, int nvel, long ncol, double *in, double *out, void *stream) {
  ARGCHK(op && in && out);
  if (nvel < 1 || ncol < 0) return chebhip_fail(CHEBHIP_ERR_ARG, "bad pencil geometry");
  if (ncol == 0) return 0;
  const size_t Np = (size_t)op->gP0 * (size_t)ncol;
  if (op->pext) {
    const DiffMat *m[2] = {&op->mats[op->gP0], &op->matsP[op->gP0]};
    SweepParams sp[2] = {};
    sp[0].ncols = (unsigned)(nvel * ncol); sp[0].inner = (unsigned)ncol;
    sp[0].in0 = in; sp[0].in_mode = IN_PLAIN; sp[0].out = out; sp[0].out_mode = OUT_STORE; sp[0].alpha = 1.0;
    sp[1].ncols = (unsigned)ncol; sp[1].inner = (unsigned)ncol;
    sp[1].in0 = in + (size_t)nvel * Np; sp[1].in_mode = IN_PLAIN; sp[1].out = out + (size_t)nvel * Np; sp[1].out_mode = OUT_STORE; sp[1].alpha = 1.0;
    SHIPCHK(sweep_launch_multi(2, m, sp, (hipStream_t)stream));      // (one launch where the 16-byte kernels run, else one each)
    return 0;
  }
  int rc = stokes_op_pencil_sweep(op, nvel, ncol, in, out, stream); if (rc) return rc;
  return stokes_op_pencil_pressure(op, ncol, in + (size_t)nvel * Np, out + (size_t)nvel * Np, stream);
}
// The same three entry points with the pencil's planes read IN PLACE from the ranks' slab fields (sweep.h GatherSrc; slabx.hip, direct
// transports): kind 0 = stokes_op_pencil_sweep on nf fields, 1 = stokes_op_pencil_pressure (its one field), 2 = stokes_op_pencil_sweep_pressure
// (nf - 1 velocity fields and the pressure field, two jobs of one launch).  Needs the extrapolation folded into the pressure matrix (pext:
// the default) -- the pass form rewrites the pencil, which does not exist here.  out: the pencil result (nf, P0, ncol), dense.
namespace chebhip {
bool stokes_pencil_gather_supported(const stokes_op *op) {
  if (!op || !op->pext || opt(OPT_GENERAL_KERNELS) || opt(OPT_SEPARATE_LAUNCHES)) return false;
  auto a = op->mats.find(op->gP0); auto b = op->matsP.find(op->gP0);
  return a != op->mats.end() && b != op->matsP.end() && a->second.KS >= 16 && b->second.KS == a->second.KS;
}
int stokes_pencil_gather_try(stokes_op *op, int kind, int nf, long ncol, const GatherSrc &g, double *out, hipStream_t st, bool *done) {
  *done = false;
  if (!op || !out || nf < 1 || ncol <= 0 || !op->pext || kind < 0 || kind > 2 || (kind == 1 && nf != 1) || (kind == 2 && nf < 2)) return 0;
  const size_t Np = (size_t)op->gP0 * (size_t)ncol;
  const DiffMat *m[9]; SweepParams sp[9];
  int ng = 0;
  if (kind != 2) {
    sp[0] = SweepParams{};
    sp[0].ncols = (unsigned)(nf * ncol); sp[0].inner = (unsigned)ncol;
    sp[0].in_mode = IN_PLAIN; sp[0].out = out; sp[0].out_mode = OUT_STORE; sp[0].alpha = 1.0;
    m[0] = kind == 1 ? &op->matsP[op->gP0] : &op->mats[op->gP0];
    ng = 1;
  } else {
    const int nvel = nf - 1;
    m[0] = &op->mats[op->gP0]; m[1] = &op->matsP[op->gP0];
    sp[0] = SweepParams{}; sp[1] = SweepParams{};
    sp[0].ncols = (unsigned)(nvel * ncol); sp[0].inner = (unsigned)ncol;
    sp[0].in_mode = IN_PLAIN; sp[0].out = out; sp[0].out_mode = OUT_STORE; sp[0].alpha = 1.0;
    sp[1].ncols = (unsigned)ncol; sp[1].inner = (unsigned)ncol; sp[1].gfield0 = (unsigned)nvel;
    sp[1].in_mode = IN_PLAIN; sp[1].out = out + (size_t)nvel * Np; sp[1].out_mode = OUT_STORE; sp[1].alpha = 1.0;
    ng = 2;
  }
  const unsigned gmask = (1u << ng) - 1u;
  // the callback's parked local sweeps (stokes_op::npend) ride in the same launch when the jobs can share one
  if (op->npend > 0 && ng + op->npend <= 9) {
    int n = ng;
    for (int j = 0; j < op->npend; j++, n++) { m[n] = op->pend_m[j]; sp[n] = op->pend_sp[j]; }
    bool all = false;
    SHIPCHK(sweep_launch_multi_gather_try(n, m, sp, gmask, g, st, &all));
    if (all) { op->npend = 0; *done = true; return 0; }
  }
  if (ng == 1) { SHIPCHK(sweep_launch_gather(*m[0], sp[0], g, st, done)); }
  else { SHIPCHK(sweep_launch_multi_gather_try(2, m, sp, gmask, g, st, done)); }
  return 0;
}
}  // namespace chebhip

extern "C" int stokes_op_pencil_pressure(stokes_op *op, long ncol, double *p_pencil, double *gp0_pencil, void *stream) {
  ARGCHK(op && p_pencil && gp0_pencil);
  if (ncol <= 0) return ncol == 0 ? 0 : chebhip_fail(CHEBHIP_ERR_ARG, "bad pencil geometry");
  hipStream_t st = (hipStream_t)stream;
  if (op->pext) {                                  // the extrapolation is part of the matrix: p_pencil is left as it is
    SweepParams sp = {};
    sp.ncols = (unsigned)ncol; sp.inner = (unsigned)ncol;
    sp.in0 = p_pencil; sp.in_mode = IN_PLAIN; sp.out = gp0_pencil; sp.out_mode = OUT_STORE; sp.alpha = 1.0;
    SHIPCHK(sweep_launch(op->matsP[op->gP0], sp, st));
    return 0;
  }
  hipLaunchKernelGGL(k_st_preduce, dim3(pgrid(ncol)), dim3(256), 0, st, p_pencil, ncol, 0L, 1L, 1L, 0L, 0L, ncol,
                     op->gP0, (const double *)op->w0[0], (const double *)op->w1[0]);
  return stokes_op_pencil_sweep(op, 1, ncol, p_pencil, gp0_pencil, stream);
}

// State at the ABI is in the reference's layout (strain[j]: N nodes x d components, component fastest)
static int st_state_ptr(stokes_op *op, int which, double **p, size_t *n, bool *soa) {
  *soa = false;
  if (which == 0) { *p = op->eta; *n = (size_t)op->N; }
  else if (which == 1) { *p = op->deta; *n = (size_t)op->N; }
  else if (which >= 2 && which < 2 + op->d) { *p = op->strain[which - 2]; *n = (size_t)op->N * op->d; *soa = true; }
  else return chebhip_fail(CHEBHIP_ERR_ARG, "which = %d out of range", which);
  return 0;
}

extern "C" int stokes_op_get_state(stokes_op *op, int which, double *dst) {
  ARGCHK(op && dst);
  double *p; size_t n; bool soa; int rc = st_state_ptr(op, which, &p, &n, &soa); if (rc) return rc;
  SHIPCHK(hipDeviceSynchronize());
  if (soa && op->strain_stale) { if ((rc = st_sync_strain(op, nullptr))) return rc; SHIPCHK(hipStreamSynchronize(nullptr)); }
  if (!soa) { SHIPCHK(hipMemcpy(dst, p, n * sizeof(double), hipMemcpyDeviceToHost)); return 0; }
  std::vector<double> tmp(n);
  const size_t N = (size_t)op->N; const int d = op->d;
  const int j = which - 2;
  for (int k = 0; k < d; k++) {
    // symmetric storage (stokes_op::sym): only the entries with first index <= second are the symmetrised strain
    const double *src = (op->sym && k < j) ? op->strain[k] + (size_t)j * N : p + (size_t)k * N;
    SHIPCHK(hipMemcpy(tmp.data() + (size_t)k * N, src, N * sizeof(double), hipMemcpyDeviceToHost));
  }
  for (size_t l = 0; l < N; l++) for (int k = 0; k < d; k++) dst[l * d + k] = tmp[k * N + l];
  return 0;
}

extern "C" int stokes_op_set_state(stokes_op *op, int which, const double *src) {
  ARGCHK(op && src);
  double *p; size_t n; bool soa; int rc = st_state_ptr(op, which, &p, &n, &soa); if (rc) return rc;
  SHIPCHK(hipDeviceSynchronize());
  // a state set by hand starts from the complete state of the last StokesFunction (an eta' given here will meet the strain)
  if (op->strain_stale) { if ((rc = st_sync_strain(op, nullptr))) return rc; SHIPCHK(hipStreamSynchronize(nullptr)); }
  if (!soa) {
    SHIPCHK(hipMemcpy(p, src, n * sizeof(double), hipMemcpyHostToDevice));
    if (which == 1) { bool nz = false; for (size_t i = 0; i < n && !nz; i++) nz = (src[i] != 0.0); op->deta_nonzero = nz; }
    if (which == 0) { bool same = n > 0; for (size_t i = 1; i < n && same; i++) same = (src[i] == src[0]); op->eta_uniform = same; op->eta_value = same ? src[0] : 1.0; }
    return 0;
  }
  std::vector<double> tmp(n);
  const size_t N = (size_t)op->N; const int d = op->d;
  for (size_t l = 0; l < N; l++) for (int k = 0; k < d; k++) tmp[k * N + l] = src[l * d + k];
  SHIPCHK(hipMemcpy(p, tmp.data(), n * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

// ---- min / max viscosity (the VecMin / VecMax of StokesFunction, stokes.C:731-734) ---------------------------
// Optional: StokesFunction itself does not reduce or print anything (a reduction plus a host read-back inside the
// residual would stall the stream); call this after it when the numbers are wanted.
__global__ __launch_bounds__(256) void k_minmax(long n, const double *__restrict__ a, double *__restrict__ part) {
  __shared__ double smin[256], smax[256];
  double lo = 1.0 / 0.0, hi = -1.0 / 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const double v = a[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + o]); smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + o]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = smin[0]; part[2 * blockIdx.x + 1] = smax[0]; }
}

extern "C" int stokes_op_viscosity_range(stokes_op *op, double *eta_min, double *eta_max, void *stream) {
  ARGCHK(op && eta_min && eta_max);
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = sgrid(op->N) > 256 ? 256 : sgrid(op->N);
  // p2 is free between callbacks: 2 * nb partial results
  if ((long)(2 * nb) > op->N) return chebhip_fail(CHEBHIP_ERR_SIZE, "grid too small");
  hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(256), 0, st, op->N, (const double *)op->eta, op->p2);
  std::vector<double> h(2 * nb);
  SHIPCHK(hipMemcpyAsync(h.data(), op->p2, 2 * nb * sizeof(double), hipMemcpyDeviceToHost, st));
  SHIPCHK(hipStreamSynchronize(st));
  double lo = h[0], hi = h[1];
  for (unsigned b = 1; b < nb; b++) { lo = h[2 * b] < lo ? h[2 * b] : lo; hi = h[2 * b + 1] > hi ? h[2 * b + 1] : hi; }
  *eta_min = lo; *eta_max = hi;
  return 0;
}

// ---- StokesStateView (stokes.C:1821-1894): the ASCII legacy-VTK dump behind -output_vtk ----------------------
// Same sections, order and number format as the reference ("%20e ", three values per point line, tensors as 3x3):
// POINTS (node coordinates), velocity (with Dirichlet values), pressure (boundary filled by
// StokesPressureReduceOrder), vel_force / div_force (the force vector treated the same way), eta, deta, strain.
extern "C" int stokes_op_write_vtk(stokes_op *op, const double *state_dev, const char *path) {
  ARGCHK(op && path); ARGCHK(VEC_OK(state_dev));
  if (op->slab) return chebhip_fail(CHEBHIP_ERR_ARG, "stokes_op_write_vtk: serial handles only");
  const int d = op->d; const long N = op->N;
  SHIPCHK(hipDeviceSynchronize());
  { int rc = st_sync_strain(op, nullptr); if (rc) return rc; }
  std::vector<double> v(N * d), p(N), fv(N * d, 0.0), fp(N, 0.0), eta(N), deta(N), strain((size_t)d * N * d);
  auto fetch = [&](const double *src, std::vector<double> &vel, std::vector<double> &pre) -> int {
    st_local(op, d + 1, d, src, op->dirloc, op->xL, op->pL, nullptr);                 // scatters + dirichlet (:1827-1838)
    st_pressure_extrapolate(op, op->pL, nullptr);                                       // :1837
    SHIPCHK(hipMemcpy(vel.data(), op->xL, (size_t)N * d * sizeof(double), hipMemcpyDeviceToHost));
    SHIPCHK(hipMemcpy(pre.data(), op->pL, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
  };
  int rc = fetch(state_dev, v, p); if (rc) return rc;
  if (op->force) { rc = fetch(op->force, fv, fp); if (rc) return rc; }                 // :1840-1851
  SHIPCHK(hipMemcpy(eta.data(), op->eta, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
  SHIPCHK(hipMemcpy(deta.data(), op->deta, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
  for (int j = 0; j < d; j++)
    for (int k = 0; k < d; k++) {       // symmetric storage: entries below the diagonal come from their mirror
      const double *src = (op->sym && k < j) ? op->strain[k] + (size_t)j * N : op->strain[j] + (size_t)k * N;
      SHIPCHK(hipMemcpy(strain.data() + (size_t)j * N * d + (size_t)k * N, src, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    }
  FILE *f = fopen(path, "w");
  if (!f) return chebhip_fail(CHEBHIP_ERR_ARG, "cannot open %s", path);
  const int m = op->dims[0], n = op->dims[1], pp = d > 2 ? op->dims[2] : 1;
  fprintf(f, "# vtk DataFile Version 2.0\nStokes Output\nASCII\nDATASET STRUCTURED_GRID\n");
  fprintf(f, "DIMENSIONS %d %d %d\nPOINTS %ld double\n", m, n, pp, N);
  {  // c->coord: x = cos(i pi/(dim-1)) per dimension (stokes.C:296), three values per line
    std::vector<int> ind(d, 0);
    for (long l = 0; l < N; l++) {
      for (int j = 0; j < d; j++) fprintf(f, "%20e ", cos(ind[j] * 3.14159265358979323846 / (op->dims[j] - 1)));
      for (int j = d; j < 3; j++) fprintf(f, "0 ");
      fprintf(f, "\n");
      for (int j = d - 1; j >= 0; j--) { if (++ind[j] < op->dims[j]) break; ind[j] = 0; }
    }
  }
  auto vec3 = [&](const std::vector<double> &a) {       // component-major work vector, printed node by node (StokesVecView)
    for (long l = 0; l < N; l++) { for (int j = 0; j < d; j++) fprintf(f, "%20e ", a[(size_t)j * N + l]); for (int j = d; j < 3; j++) fprintf(f, "0 "); fprintf(f, "\n"); }
  };
  auto scal = [&](const std::vector<double> &a) { for (long l = 0; l < N; l++) fprintf(f, "%20e \n", a[l]); };
  fprintf(f, "\nPOINT_DATA %ld\nVECTORS velocity double\n", N); vec3(v);
  fprintf(f, "\nSCALARS pressure double 1\nLOOKUP_TABLE default\n"); scal(p);
  fprintf(f, "\nVECTORS vel_force double\n"); vec3(fv);
  fprintf(f, "\nSCALARS div_force double 1\nLOOKUP_TABLE default\n"); scal(fp);
  fprintf(f, "\nSCALARS eta double 1\nLOOKUP_TABLE default\n"); scal(eta);
  fprintf(f, "\nSCALARS deta double 1\nLOOKUP_TABLE default\n"); scal(deta);
  fprintf(f, "\nTENSORS strain double\n");
  for (long l = 0; l < N; l++) {
    for (int j = 0; j < 3; j++) {
      for (int k = 0; k < 3; k++) fprintf(f, "%20e ", (j < d && k < d) ? strain[(size_t)j * N * d + (size_t)k * N + l] : 0.0);
      fprintf(f, "\n");
    }
    fprintf(f, "\n");
  }
  if (fclose(f) != 0) return chebhip_fail(CHEBHIP_ERR_ARG, "write to %s failed", path);
  return 0;
}

