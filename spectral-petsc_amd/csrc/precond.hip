// precond.hip -- the finite-difference preconditioner of the reference on the device.
//
// The reference preconditions its Krylov solves with a low-order matrix on the collocation nodes:
//   * FormJacobian (elliptic.C:537-590): the 2d+1-point finite-difference matrix P of
//     -div(eta grad u) - div(deta u grad u0) on the Gauss-Lobatto grid, Dirichlet rows eliminated, handed to
//     PCILU with 2 levels of fill (elliptic.C:184-185);
//   * StokesPCSetUp0 (stokes.C:1160-1241): the same stencil with eta only, per velocity component (MatVVPC),
//     inside KSPVelocity / KSPSchurVelocity (stokes.C:328-341).
// Here P is kept as 2d+1 coefficient arrays in the interior (global-vector) layout and applied by a stencil
// kernel (fd_mult: what MatMult on the AIJ matrix would give, used by the tests and by the defect correction
// below).  The approximate solve that ILU provides in the reference is replaced by something that suits the
// chip: on the tensor grid the constant-coefficient part of P is  sum_k I x .. x T_k x .. x I  with a 1-D
// three-point operator T_k, which is diagonalised ONCE (diffmat.cpp: fdm_line), so that
//     P_1^-1 = (S_0 x S_1 x S_2) diag(1 / (l_i + l_j + l_k)) (S_0^-1 x S_1^-1 x S_2^-1)
// is 2d batched dense line transforms -- the very sweep kernel the operator itself runs on (each transform
// is applied as its centro-symmetric plus its centro-antisymmetric part, two launches) -- and one pointwise
// scaling.  For eta == 1 that is the exact inverse of P; for variable coefficients the approximate solve is
// `sweeps` iterations of GMRES on P z = r with P_1^-1 (1/eta) as its right preconditioner (sweeps = 0: that
// preconditioner alone) -- a varying, hence flexible, preconditioner for the outer FGMRES, as ILU(2) is not.
#include "../../include/chebhip.h"
#include "ops.h"
#include "sweep.h"
#include "timers.h"
#include <cmath>
#include <map>
#include <new>
#include <vector>

using namespace chebhip;

#define PHIPCHK(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {

constexpr int MAXD = 10;
struct Geo { int d; int dims[MAXD]; long gs[MAXD]; };      // local extents, interior strides

static inline unsigned pgrid(long n) { long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
#define GS_LOOP(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// One row of P per interior node, exactly the arithmetic of elliptic.C:556-579 (gradu[j] null: the deta * du0
// terms are absent, stokes.C:1217-1222).  cf[0] diagonal, cf[1+2j] / cf[2+2j] the neighbours at -1 / +1 along
// dimension j; a neighbour on the boundary has no column (MatSetValues drops negative indices): coefficient 0.
struct GradPtrs { const double *p[MAXD]; };
struct CoordPtrs { const double *p[MAXD]; };
__global__ void k_fd_assemble(Geo geo, long N, long G, const int *__restrict__ ixL, const double *__restrict__ eta,
                              const double *__restrict__ deta, GradPtrs gu, CoordPtrs xs, double *__restrict__ cf,
                              double *__restrict__ eta_g) {
  GS_LOOP(l, N) {
    const int g = ixL[l];
    if (g < 0) continue;
    long rem = l, ls[MAXD]; int ind[MAXD];
    { long s = 1; for (int j = geo.d - 1; j >= 0; j--) { ls[j] = s; s *= geo.dims[j]; } }
    for (int j = 0; j < geo.d; j++) { ind[j] = (int)(rem / ls[j]); rem -= (long)ind[j] * ls[j]; }
    double v0 = 0.0;
    for (int j = 0; j < geo.d; j++) {
      const long iM = l - ls[j], iP = l + ls[j];
      const double x0 = xs.p[j][ind[j]], xMM = xs.p[j][ind[j] - 1], xPP = xs.p[j][ind[j] + 1];
      const double xM = 0.5 * (xMM + x0), idxM = 1.0 / (x0 - xMM), xP = 0.5 * (x0 + xPP), idxP = 1.0 / (xPP - x0), idx = 1.0 / (xP - xM);
      const double eM = 0.5 * (eta[iM] + eta[l]), eP = 0.5 * (eta[iP] + eta[l]);
      double tM = 0.0, tP = 0.0;
      if (gu.p[j]) {
        const double deM = 0.5 * (deta[iM] + deta[l]), du0M = 0.5 * (gu.p[j][iM] + gu.p[j][l]);
        const double deP = 0.5 * (deta[iP] + deta[l]), du0P = 0.5 * (gu.p[j][iP] + gu.p[j][l]);
        tM = 0.5 * deM * du0M; tP = 0.5 * deP * du0P;
      }
      const double vM = -idx * (idxM * eM - tM), vP = -idx * (idxP * eP + tP);
      v0 += idx * (idxP * eP + idxM * eM - (tP - tM));
      cf[(long)(1 + 2 * j) * G + g] = ixL[iM] >= 0 ? vM : 0.0;
      cf[(long)(2 + 2 * j) * G + g] = ixL[iP] >= 0 ? vP : 0.0;
    }
    cf[g] = v0;
    eta_g[g] = eta[l];
  }
}

// slab mode: only the viscosity at the interior nodes is needed (P_1^-1 (r / eta))
__global__ void k_eta_g(long N, const int *__restrict__ ixL, const double *__restrict__ eta, double *__restrict__ eta_g) {
  GS_LOOP(l, N) { const int g = ixL[l]; if (g >= 0) eta_g[g] = eta[l]; }
}

// y = P x on nf stacked fields of G interior values each
__global__ void k_fd_mult(Geo geo, long G, int nf, const double *__restrict__ cf, const double *__restrict__ x, double *__restrict__ y) {
  GS_LOOP(t, G * nf) {
    const long f = t / G, g = t - f * G;
    const double *xf = x + f * G;
    double s = cf[g] * xf[g];
    for (int j = 0; j < geo.d; j++) {
      const double cM = cf[(long)(1 + 2 * j) * G + g], cP = cf[(long)(2 + 2 * j) * G + g];
      if (cM != 0.0) s += cM * xf[g - geo.gs[j]];
      if (cP != 0.0) s += cP * xf[g + geo.gs[j]];
    }
    y[t] = s;
  }
}

// modal scaling: t /= (l_0[i_0] + ... + l_{d-1}[i_{d-1}])
struct LamPtrs { const double *p[MAXD]; };
__global__ void k_modal_scale(Geo geo, long G, int nf, LamPtrs lam, double *__restrict__ t) {
  GS_LOOP(q, G * nf) {
    long g = q % G; double s = 0.0;
    for (int j = 0; j < geo.d; j++) { const long i = g / geo.gs[j]; g -= i * geo.gs[j]; s += lam.p[j][i]; }
    t[q] = t[q] / s;
  }
}
// The same for d <= 3 without an integer-division chain per element (four 64-bit divisions cost more than the pass moves):
// block (x, line, field) scales a piece of one line of the last dimension; the line's indices are formed once per block.
__global__ __launch_bounds__(256) void k_modal_scale3(int d, int n1, int nl, long G, const double *__restrict__ l0, const double *__restrict__ l1,
                                                      const double *__restrict__ l2, double *__restrict__ t) {
  const unsigned line = blockIdx.y;                       // d = 3: i0 * n1 + i1; d = 2: i0; d = 1: 0
  double s = 0.0;
  if (d == 3) { const unsigned i0 = line / (unsigned)n1, i1 = line - i0 * (unsigned)n1; s = l0[i0] + l1[i1]; }
  else if (d == 2) s = l0[line];
  const double *ll = d == 3 ? l2 : (d == 2 ? l1 : l0);
  double *row = t + (long)blockIdx.z * G + (long)line * nl;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nl; i += gridDim.x * 256) row[i] = row[i] / (s + ll[i]);
}

// w = 1 / (l_0[i_0] + ... + l_{d-1}[i_{d-1}]) on nf stacked copies of the interior grid (d <= 3): the operand of the last forward
// line transform when it applies the modal scaling itself (OUT_MUL, sweep.h) -- one pass over the nf G values less per solve
__global__ __launch_bounds__(256) void k_modal_weights3(int d, int n1, int nl, long G, const double *__restrict__ l0, const double *__restrict__ l1,
                                                        const double *__restrict__ l2, double *__restrict__ w) {
  const unsigned line = blockIdx.y;
  double s = 0.0;
  if (d == 3) { const unsigned i0 = line / (unsigned)n1, i1 = line - i0 * (unsigned)n1; s = l0[i0] + l1[i1]; }
  else if (d == 2) s = l0[line];
  const double *ll = d == 3 ? l2 : (d == 2 ? l1 : l0);
  double *row = w + (long)blockIdx.z * G + (long)line * nl;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nl; i += gridDim.x * 256) row[i] = 1.0 / (s + ll[i]);
}

// out = (a - (y ? y : 0)) / eta_g  on nf stacked fields
// einv = 1 / eta_g on nf stacked copies: the operand of the first forward line transform when it divides by the viscosity itself (IN_MUL)
__global__ void k_eta_inverse(long G, const double *__restrict__ eta_g, double *__restrict__ einv) {
  const long f0 = (long)blockIdx.y * G;
  GS_LOOP(g, G) einv[f0 + g] = 1.0 / eta_g[g];
}
// (field = blockIdx.y: no 64-bit modulo per element)
__global__ void k_resid_over_eta(long G, int nf, const double *__restrict__ a, const double *__restrict__ y,
                                 const double *__restrict__ eta_g, double *__restrict__ out) {
  const long f0 = (long)blockIdx.y * G;
  GS_LOOP(g, G) { const long q = f0 + g; const double r = y ? a[q] - y[q] : a[q]; out[q] = r / eta_g[g]; }
}

// node-major interleaved (I nodes x d components, the reference's velocity vectors) <-> component-major
// one thread per node: its nf values are contiguous on the node-major side (a wave moves 64 nf contiguous doubles) and
// coalesced field by field on the other
__global__ void k_deinterleave(long G, int nf, const double *__restrict__ a, double *__restrict__ b) {
  GS_LOOP(g, G) { for (int f = 0; f < nf; f++) b[(long)f * G + g] = a[g * nf + f]; }
}
__global__ void k_interleave(long G, int nf, const double *__restrict__ b, double *__restrict__ a) {
  GS_LOOP(g, G) { for (int f = 0; f < nf; f++) a[g * nf + f] = b[(long)f * G + g]; }
}
// out (component-major) = a (node-major) / eta_g: the first step of P_1^-1 (1/eta) on the Stokes velocity vectors, without a
// component-major copy of the right-hand side in between
__global__ void k_deinterleave_over_eta(long G, int nf, const double *__restrict__ a, const double *__restrict__ eta_g, double *__restrict__ out) {
  GS_LOOP(g, G) { const double e = eta_g[g]; for (int f = 0; f < nf; f++) out[(long)f * G + g] = a[g * nf + f] / e; }
}

// ---------------------------------------------------------------------------------------------
// k_fdm_zsolve16: the last forward line transform, the modal scaling and the first backward line transform of the fast-diagonalisation
// solve in ONE launch.  The three act on the same contiguous lines (the last dimension) -- y = S ( W .* (S^-1 x) ) line by line -- so
// the modal coefficients never have to leave the chip: the parity halves the forward product leaves in the accumulators (the raw-mode
// layout of sweep.h: even modes at positions p < H, odd modes at M-1-q) are exactly the split input the backward product takes.
//   A  x lines -> LDS, parity-split          B  c = S^-1 x on the matrix cores; c .*= W; back to LDS as the already split halves
//   D  y = S c on the matrix cores -> HBM
// 24 B/value (x, W, y) instead of 40 for the two launches it replaces.  Lines of 66 .. 128 interior points (KS = 16), M even; tiles of
// 32 lines.  (W and y move as 8-byte pieces in accumulator layout; 16-byte pieces after a DPP lane exchange, as in the sweep kernels'
// epilogue, were built and gave nothing: 144 -> 145-147 us for the MatVVPC solve.)
typedef double fz_v4 __attribute__((ext_vector_type(4)));
struct FzParams { int M, H; unsigned ncols, ntiles; const double *x; double *y; const double *FE, *FO, *BE, *BO;
                  // the modal weights 1 / ((l_0[i] + l_1[j]) + l_z[k]) are formed on the fly (the association of k_modal_weights3: the same
                  // bits as its array W, which cost this launch a third of its bytes): line -> (i, j) within a field of `flines` lines
                  int d, n1; unsigned flines; const double *l0, *l1, *lz; };
// (image row pitch: odd, as in sweep_vec.hip / stokes.hip -- conflict-free operand reads, 8-byte-aligned lines)
#ifndef FZ_PAD
#define FZ_PAD 1
#endif
constexpr int FZ_KS = 16, FZ_LDJ = 4 * FZ_KS + FZ_PAD, FZ_NT = 32;
__device__ __forceinline__ void fz_put2(double *dst, double2 v) {
  if (FZ_LDJ % 2 == 0) *(double2 *)dst = v; else { dst[0] = v.x; dst[1] = v.y; }
}
__global__ __launch_bounds__(256, 3) void k_fdm_zsolve16(const FzParams p) {
  // Workgroups of 256 threads (wave = m-tile, two sub-tiles of 16 lines), three per CU, out of phase with each other; the matrix
  // fragments are fetched from L2 per stage (64 KB) rather than held: 168 registers have to do
  __shared__ __attribute__((aligned(16))) double sE[FZ_NT * FZ_LDJ], sO[FZ_NT * FZ_LDJ];
  const int tid = threadIdx.x, lane = tid & 63, mt = tid >> 6;
  const int kq = lane >> 4, l16 = lane & 15;
  const int M = p.M, H = p.H, mm = M - 1;
  const int oi = mt * 16 + l16;                                        // modal position / output point of this lane (mirror: mm - oi)
  auto chains = [&](const double *AE, const double *AO, fz_v4 (&ce)[2], fz_v4 (&co)[2]) {
    asm volatile("" : "+s"(AE), "+s"(AO));
    double ae[FZ_KS], ao[FZ_KS];
#pragma unroll
    for (int s = 0; s < FZ_KS; s++) { const long q = ((long)(mt * FZ_KS + s)) * 64 + lane; ae[s] = AE[q]; ao[s] = AO[q]; }
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      ce[sub] = fz_v4{0.0, 0.0, 0.0, 0.0}; co[sub] = fz_v4{0.0, 0.0, 0.0, 0.0};
      const int frag = (sub * 16 + l16) * FZ_LDJ + kq;
#pragma unroll
      for (int s = 0; s < FZ_KS; s++) {
        ce[sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(sE[frag + 4 * s], ae[s], ce[sub], 0, 0, 0);
        co[sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(sO[frag + 4 * s], ao[s], co[sub], 0, 0, 0);
      }
    }
  };
  const double lze = oi < H ? p.lz[oi] : 0.0, lzo = oi < H ? p.lz[mm - oi] : 0.0;      // this lane's two modes along z
  for (unsigned tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    // ---- A: four slots per thread: line, points (j, j + 1) and their mirrors (mm - j - 1, mm - j); H odd: the last pair is its own mirror
    {
      double2 rj[4], rm[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int id = tid + 256 * u, line = id >> 5, j = 2 * (id & 31);
        const unsigned gl = tile * FZ_NT + line;
        rj[u] = make_double2(0.0, 0.0); rm[u] = rj[u];
        if (j < H && gl < p.ncols) { rj[u] = *(const double2 *)(p.x + (long)gl * M + j); rm[u] = *(const double2 *)(p.x + (long)gl * M + (mm - j - 1)); }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int id = tid + 256 * u, line = id >> 5, j = 2 * (id & 31);
        const bool two = j + 1 < H;                                    // (j + 1 == H: that point belongs to the mirror half)
        fz_put2(sE + line * FZ_LDJ + j, make_double2(rj[u].x + rm[u].y, two ? rj[u].y + rm[u].x : 0.0));
        fz_put2(sO + line * FZ_LDJ + j, make_double2(rj[u].x - rm[u].y, two ? rj[u].y - rm[u].x : 0.0));
      }
    }
    __syncthreads();
    // ---- B: modal coefficients, scaled, back to LDS as the split halves of the backward product
    {
      double we[2][4], wo[2][4];
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned gl = tile * FZ_NT + sub * 16 + 4 * r + kq;
          const bool ok = oi < H && gl < p.ncols;
          double sxy = 0.0;
          if (ok) {
            const unsigned lf = gl % p.flines;
            if (p.d == 3) { const unsigned i0 = lf / (unsigned)p.n1, i1 = lf - i0 * (unsigned)p.n1; sxy = p.l0[i0] + p.l1[i1]; }
            else sxy = p.l0[lf];
          }
          we[sub][r] = ok ? 1.0 / (sxy + lze) : 0.0; wo[sub][r] = ok ? 1.0 / (sxy + lzo) : 0.0;
        }
      fz_v4 ce[2], co[2];
      chains(p.FE, p.FO, ce, co);
      __syncthreads();                                                 // every wave has finished reading the x images
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int line = sub * 16 + 4 * r + kq;
          sE[line * FZ_LDJ + oi] = we[sub][r] * ce[sub][r];             // (positions H .. 4 KS - 1 get 0: the zero padding of the images)
          sO[line * FZ_LDJ + oi] = wo[sub][r] * co[sub][r];
        }
    }
    __syncthreads();
    // ---- D: y = S c: row i <- E + O, row mm - i <- E - O (the backward matrix is stored centro-symmetric)
    {
      fz_v4 ce[2], co[2];
      chains(p.BE, p.BO, ce, co);
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned gl = tile * FZ_NT + sub * 16 + 4 * r + kq;
          if (oi < H && gl < p.ncols) {
            double *row = p.y + (long)gl * M;
            row[oi] = ce[sub][r] + co[sub][r]; row[mm - oi] = ce[sub][r] - co[sub][r];
          }
        }
    }
    __syncthreads();                                                   // the images are rewritten by the next tile
  }
}

struct LineMats { DiffMat Fcs, Fca, Bcs, Bca, Fraw, Braw; double *lam = nullptr; bool ok = false; };

}  // namespace

struct chebhip_fdpc {
  Geo geo;
  long N = 0, G = 0;
  int nf = 1;                                  // 1: scalar operator; d: Stokes velocity (component-major inside)
  bool interleaved = false;                    // vectors at the ABI are node-major (Stokes)
  std::map<int, LineMats> lines;               // per distinct extent
  std::vector<unsigned> inner_g, ncols_g;
  double *xs[MAXD] = {nullptr};
  double *cf = nullptr, *eta_g = nullptr;
  double *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr, *t4 = nullptr, *t5 = nullptr;   // nf * G each
  double *W = nullptr; bool W_tried = false;   // reciprocal modal weights, nf stacked copies (built on first use: fdm_solve)
  double *Einv = nullptr; bool Einv_ok = false;   // 1 / eta_g, nf stacked copies (rebuilt by fdpc_update; used by fdm_solve)
  int sweeps = 1;
  chebhip_fgmres *inner = nullptr; int inner_m = 0;   // the approximate solve with variable coefficients
  bool assembled = false;
  ell_op *eop = nullptr; stokes_op *sop = nullptr;
  // Slab mode: the handle holds the interior nodes of a slab of planes of dimension 0; the line transforms along
  // dimension 0 need whole lines and go through `dim0` (slabx.hip: slab -> pencil, chebhip_fdpc_pencil_transform, back).
  // geo.dims[0] is the GLOBAL extent (matrices, eigenvalues), i0_off the global index of the slab's first interior plane.
  // Only the fast-diagonalisation solve z = P_1^-1 (r / eta) (sweeps = 0) exists on slabs: the stencil of P would need halos.
  bool slab = false; chebhip_fdpc_dim0_fn dim0 = nullptr; void *dim0_ctx = nullptr; long i0_off = 0;
};

static void fdpc_free(chebhip_fdpc *pc) {
  if (!pc) return;
  if (pc->inner) chebhip_fgmres_destroy(pc->inner);
  for (auto &kv : pc->lines) {
    if (!kv.second.ok) continue;
    diffmat_destroy(&kv.second.Fcs); diffmat_destroy(&kv.second.Fca); diffmat_destroy(&kv.second.Bcs); diffmat_destroy(&kv.second.Bca);
    diffmat_destroy(&kv.second.Fraw); diffmat_destroy(&kv.second.Braw);
    if (kv.second.lam) (void)hipFree(kv.second.lam);
  }
  for (int k = 0; k < MAXD; k++) if (pc->xs[k]) (void)hipFree(pc->xs[k]);
  double *all[] = {pc->cf, pc->eta_g, pc->t0, pc->t1, pc->t2, pc->t3, pc->t4, pc->t5, pc->W, pc->Einv};
  for (double *p : all) if (p) (void)hipFree(p);
  delete pc;
}

static int fdpc_create(const FdView &v0, int nf, bool interleaved, chebhip_fdpc **out, int gP0 = 0) {
  *out = nullptr;
  FdView v = v0;
  std::vector<int> gdims(v0.dims, v0.dims + (v0.d >= 1 && v0.d <= MAXD ? v0.d : 0));
  if (gP0 > 0 && !gdims.empty()) { gdims[0] = gP0; v.dims = gdims.data(); }      // slab mode: matrices of the global extent
  if (v.d < 1 || v.d > MAXD) return chebhip_fail(CHEBHIP_ERR_DIMS, "d = %d out of range", v.d);
  for (int k = 0; k < v.d; k++) {
    if (v.dims[k] < 3) return chebhip_fail(CHEBHIP_ERR_SIZE, "dims[%d] = %d: the preconditioner needs interior nodes", k, v.dims[k]);
    if (v.dims[k] > 258) return chebhip_fail(CHEBHIP_ERR_ARG, "dims[%d] = %d: fast diagonalisation supports at most 258 points per line", k, v.dims[k]);
  }
  chebhip_fdpc *pc = new (std::nothrow) chebhip_fdpc;
  if (!pc) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  pc->geo.d = v.d; pc->N = v.N; pc->G = v.G; pc->nf = nf; pc->interleaved = interleaved;
  for (int k = 0; k < v.d; k++) pc->geo.dims[k] = v.dims[k];
  { long s = 1; for (int k = v.d - 1; k >= 0; k--) { pc->geo.gs[k] = s; s *= (v.dims[k] - 2); } }
#define PCCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fdpc_free(pc); \
    return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
  pc->inner_g.resize(v.d); pc->ncols_g.resize(v.d);
  for (int k = 0; k < v.d; k++) {
    const int P = v.dims[k], M = P - 2;
    pc->inner_g[k] = (unsigned)pc->geo.gs[k];
    pc->ncols_g[k] = (unsigned)(v.G / M);                      // (dimension 0 of a slab: unused, its transforms run on pencils)
    std::vector<double> x(P);
    for (int i = 0; i < P; i++) x[i] = cos(i * 3.14159265358979323846 / (P - 1));          // elliptic.C:279, stokes.C:296
    PCCHK(hipMalloc((void **)&pc->xs[k], P * sizeof(double)));
    PCCHK(hipMemcpy(pc->xs[k], x.data(), P * sizeof(double), hipMemcpyHostToDevice));
    if (pc->lines.count(P)) continue;
    LineMats lm;
    std::vector<long double> S, Si, lam, part;
    if (!fdm_line(P, S, Si, lam)) { fdpc_free(pc); return chebhip_fail(CHEBHIP_ERR_ARG, "eigen-decomposition of the %d-point line operator failed", P); }
    centro_part(M, Si, 1, part); PCCHK(diffmat_from_dense(M, part.data(), 1, &lm.Fcs));
    centro_part(M, Si, 0, part); PCCHK(diffmat_from_dense(M, part.data(), 0, &lm.Fca));
    centro_part(M, S, 1, part);  PCCHK(diffmat_from_dense(M, part.data(), 1, &lm.Bcs));
    centro_part(M, S, 0, part);  PCCHK(diffmat_from_dense(M, part.data(), 0, &lm.Bca));
    {
      // The same two transforms as ONE product each.  fdm_line orders the modes by parity (even modes at the positions
      // p < H, the q-th odd mode at M-1-q), so with e, o the parity split of a nodal line
      //   forward:  c_p = sum_j Sinv[p][j] e_j (p < H),  c_{M-1-q} = sum_j Sinv[M-1-q][j] o_j      -> halves stored raw
      //   backward: x_i = sum_p S[i][p] c_p + sum_q S[i][M-1-q] c_{M-1-q},  x_{M-1-i} = the difference -> input taken raw
      const int m = M - 1, Hh = (M + 1) / 2;
      std::vector<long double> E((size_t)Hh * Hh, 0.0L), O((size_t)Hh * Hh, 0.0L);
      for (int q = 0; q < Hh; q++)
        for (int j = 0; j < Hh; j++) {
          E[(size_t)q * Hh + j] = Si[(size_t)q * M + j];
          O[(size_t)q * Hh + j] = (2 * q == m || 2 * j == m) ? 0.0L : Si[(size_t)(m - q) * M + j];
        }
      PCCHK(diffmat_from_blocks(M, E.data(), O.data(), 1, &lm.Fraw));
      for (int i = 0; i < Hh; i++)
        for (int j = 0; j < Hh; j++) {
          E[(size_t)i * Hh + j] = S[(size_t)i * M + j];
          O[(size_t)i * Hh + j] = (2 * j == m) ? 0.0L : S[(size_t)i * M + (m - j)];
        }
      PCCHK(diffmat_from_blocks(M, E.data(), O.data(), 1, &lm.Braw));
    }
    std::vector<double> ld(M); for (int i = 0; i < M; i++) ld[i] = (double)lam[i];
    PCCHK(hipMalloc((void **)&lm.lam, M * sizeof(double)));
    PCCHK(hipMemcpy(lm.lam, ld.data(), M * sizeof(double), hipMemcpyHostToDevice));
    lm.ok = true;
    pc->lines[P] = lm;
  }
  const size_t gb = (size_t)(v.G > 0 ? v.G : 1) * sizeof(double);
  PCCHK(hipMalloc((void **)&pc->cf, (2 * v.d + 1) * gb));
  PCCHK(hipMalloc((void **)&pc->eta_g, gb));
  PCCHK(hipMalloc((void **)&pc->t0, nf * gb)); PCCHK(hipMalloc((void **)&pc->t1, nf * gb));
  PCCHK(hipMalloc((void **)&pc->t2, nf * gb)); PCCHK(hipMalloc((void **)&pc->t3, nf * gb));
  PCCHK(hipMalloc((void **)&pc->t4, nf * gb)); PCCHK(hipMalloc((void **)&pc->t5, nf * gb));
  // the operand arrays of the pointwise steps that ride inside the line transforms (fdm_solve: filled on first use / by update);
  // allocated here so that no solve meets a hipMalloc.  A failed allocation only means the separate passes run.
  if (!opt(OPT_FDM_PASSES) && v.G > 0) {
    bool any_long = false;
    for (int k = 0; k < v.d; k++) any_long = any_long || v.dims[k] - 2 > 64;
    if (v.d >= 2 && v.d <= 3 && hipMalloc((void **)&pc->W, nf * gb) != hipSuccess) { pc->W = nullptr; (void)hipGetLastError(); }
    if (any_long && hipMalloc((void **)&pc->Einv, nf * gb) != hipSuccess) { pc->Einv = nullptr; (void)hipGetLastError(); }
  }
#undef PCCHK
  *out = pc;
  return 0;
}

// 1 / eta for the first forward transform (lines of more than 64 points only: the kernels that take IN_MUL)
static int fdpc_eta_inverse(chebhip_fdpc *pc, hipStream_t st) {
  pc->Einv_ok = false;
  if (opt(OPT_FDM_PASSES) || pc->G == 0) return 0;
  bool any_long = false;
  for (int k = 0; k < pc->geo.d; k++) any_long = any_long || pc->geo.dims[k] - 2 > 64;
  if (!any_long) return 0;
  if (!pc->Einv) return 0;              // (allocated at create)
  hipLaunchKernelGGL(k_eta_inverse, dim3(pgrid(pc->G), (unsigned)pc->nf), dim3(256), 0, st, pc->G, (const double *)pc->eta_g, pc->Einv);
  PHIPCHK(hipGetLastError());
  pc->Einv_ok = true;
  return 0;
}

static int fdpc_update(chebhip_fdpc *pc, hipStream_t st) {
  FdView v;
  int rc = pc->eop ? ell_op_fd_view_any(pc->eop, &v, nullptr) : stokes_op_fd_view_any(pc->sop, &v, nullptr);
  if (!rc && pc->eop) rc = ell_op_sync_coeffs(pc->eop, (void *)st);
  if (rc) return rc;
  if (pc->G == 0) { pc->assembled = true; return 0; }
  if (pc->slab) {
    hipLaunchKernelGGL(k_eta_g, dim3(pgrid(pc->N)), dim3(256), 0, st, pc->N, v.ixL, v.eta, pc->eta_g);
    PHIPCHK(hipGetLastError());
    pc->assembled = true;
    return fdpc_eta_inverse(pc, st);
  }
  GradPtrs gu; CoordPtrs xs;
  for (int k = 0; k < MAXD; k++) { gu.p[k] = k < v.d ? v.gradu[k] : nullptr; xs.p[k] = pc->xs[k]; }
  hipLaunchKernelGGL(k_fd_assemble, dim3(pgrid(pc->N)), dim3(256), 0, st, pc->geo, pc->N, pc->G, v.ixL, v.eta, v.deta, gu, xs, pc->cf, pc->eta_g);
  PHIPCHK(hipGetLastError());
  pc->assembled = true;
  return fdpc_eta_inverse(pc, st);
}

// y = S^-1 x (forward) or S x (backward) along dimension k of nf stacked interior fields (x != y): one raw-mode launch
// where the 16-byte kernels can run it, otherwise the centro-symmetric plus the centro-antisymmetric part (two launches)
// mul (forward transforms only; may be null): the result is multiplied by this array as it is stored (OUT_MUL) where the one-launch
// form runs, and *fused says whether it was
// in_mul (forward only; may be null): every input element is multiplied by this array as it is loaded (IN_MUL) -- the caller has
// asked line_in_mul_ok first
static int line_transform_g(LineMats &lm, unsigned ncols, unsigned inner, bool backward, const double *x, double *y, hipStream_t st,
                            const double *mul = nullptr, bool *fused = nullptr, const double *in_mul = nullptr);
static int line_transform(chebhip_fdpc *pc, int k, bool backward, const double *x, double *y, hipStream_t st, const double *mul = nullptr, bool *fused = nullptr,
                          const double *in_mul = nullptr) {
  if (fused) *fused = false;
  if (pc->slab && k == 0) return pc->dim0(pc->dim0_ctx, backward ? 1 : 0, pc->nf, x, y, (void *)st);      // collective: every rank of the slab partition
  return line_transform_g(pc->lines[pc->geo.dims[k]], pc->ncols_g[k] * (unsigned)pc->nf, pc->inner_g[k], backward, x, y, st, mul, fused, in_mul);
}
static SweepParams line_in_mul_params(unsigned ncols, unsigned inner, const double *x, double *y, const double *in_mul) {
  SweepParams sm = {};
  sm.ncols = ncols; sm.inner = inner; sm.in0 = x; sm.in1 = in_mul; sm.in_mode = IN_MUL; sm.out = y; sm.alpha = 1.0; sm.out_mode = OUT_STORE; sm.raw = 1;
  return sm;
}
static bool line_in_mul_ok(chebhip_fdpc *pc, int k, const double *x, double *y, const double *in_mul) {
  if ((pc->slab && k == 0) || !in_mul) return false;
  const unsigned ncols = pc->ncols_g[k] * (unsigned)pc->nf;
  return ncols != 0 && sweep_vec_raw_eligible(pc->lines[pc->geo.dims[k]].Fraw, line_in_mul_params(ncols, pc->inner_g[k], x, y, in_mul));
}
static int line_transform_g(LineMats &lm, unsigned ncols, unsigned inner, bool backward, const double *x, double *y, hipStream_t st,
                            const double *mul, bool *fused, const double *in_mul) {
  if (fused) *fused = false;
  if (ncols == 0) return 0;
  if (in_mul && !backward) { PHIPCHK(sweep_launch(lm.Fraw, line_in_mul_params(ncols, inner, x, y, in_mul), st)); return 0; }
  SweepParams sp = {};
  sp.ncols = ncols; sp.inner = inner;
  sp.in0 = x; sp.in_mode = IN_PLAIN; sp.out = y; sp.alpha = 1.0;
  sp.out_mode = OUT_STORE;
  if (mul && fused && !backward) {
    SweepParams sm = sp; sm.out_mode = OUT_MUL; sm.acc = mul; sm.raw = 1;
    if (sweep_vec_raw_eligible(lm.Fraw, sm)) { PHIPCHK(sweep_launch(lm.Fraw, sm, st)); *fused = true; return 0; }
  }
  if (sweep_vec_raw_eligible(backward ? lm.Braw : lm.Fraw, sp)) {       // one launch: the parity split is the transform's own
    sp.raw = backward ? 2 : 1;
    PHIPCHK(sweep_launch(backward ? lm.Braw : lm.Fraw, sp, st));
    return 0;
  }
  PHIPCHK(sweep_launch(backward ? lm.Bcs : lm.Fcs, sp, st));
  sp.out_mode = OUT_ACC; sp.acc = y;
  PHIPCHK(sweep_launch(backward ? lm.Bca : lm.Fca, sp, st));
  return 0;
}

// z = P_1^-1 r (the constant-coefficient part, exactly), or z = P_1^-1 (r / eta) with over_eta; r is not modified; t0 / t1 (/ t3 with
// over_eta) are scratch (r, z must be none of them)
static int fdm_solve(chebhip_fdpc *pc, const double *r, double *z, hipStream_t st, bool over_eta = false) {
  const int d = pc->geo.d;
  const double *src = r;
  double *a = pc->t0, *b = pc->t1;
  // Forward transforms commute: on slabs in 3-D the first one is a local direction (1) rather than the collective one (0), so that
  // it can take the division by eta on its load side.  Dimension d-1 stays LAST in every order: it takes the modal scaling on its
  // store side, and the one-launch z solve below replaces exactly that transform (a 2-D slab therefore keeps the order 0, 1: its
  // only local direction is the last one, and the division by eta is the separate pass).
  int order[MAXD];
  for (int k = 0; k < d; k++) order[k] = k;
  if (pc->slab && d >= 3) { order[0] = 1; order[1] = 0; }
  const double *in_mul = nullptr;
  if (over_eta) {
    if (pc->Einv_ok && !opt(OPT_FDM_PASSES) && d >= 2 && line_in_mul_ok(pc, order[0], r, a, pc->Einv)) in_mul = pc->Einv;
    else {
      hipLaunchKernelGGL(k_resid_over_eta, dim3(pgrid(pc->G), (unsigned)pc->nf), dim3(256), 0, st, pc->G, pc->nf, r, (const double *)nullptr,
                         (const double *)pc->eta_g, pc->t3);
      src = pc->t3;
    }
  }
  LamPtrs lam; for (int k = 0; k < MAXD; k++) lam.p[k] = k < d ? pc->lines[pc->geo.dims[k]].lam : nullptr;
  const int nl = pc->geo.dims[d - 1] - 2;
  const long lines = pc->G / nl;
  if (pc->slab && d >= 2) lam.p[0] += pc->i0_off;         // the slab's first interior plane is plane i0_off of the global line
  // The modal scaling rides on the store of the last forward transform (dimension d-1 > 0: local on slabs too) where that
  // transform is one launch of the 16-byte kernels: it multiplies by W = 1 / (l_i + l_j + l_k), built here on first use (the
  // separate pass divides: the two differ in the last bit).  Option "fdm_passes" = 1 keeps both pointwise passes (A/B).
  const bool w_ok = pc->W && d >= 2 && d <= 3 && lines > 0 && lines <= 65535 && pc->nf <= 65535 && !opt(OPT_FDM_PASSES);
  if (w_ok && !pc->W_tried) {           // (filled here rather than at create: a slab handle learns its plane offset after create)
    pc->W_tried = true;
    const int n1 = d == 3 ? pc->geo.dims[1] - 2 : 1;
    hipLaunchKernelGGL(k_modal_weights3, dim3((unsigned)((nl + 255) / 256), (unsigned)lines, (unsigned)pc->nf), dim3(256), 0, st, d, n1, nl, pc->G,
                       lam.p[0], lam.p[1], lam.p[2], pc->W);
    PHIPCHK(hipGetLastError());
  }
  // The last forward transform, the scaling and the first backward transform act on the same contiguous lines: one launch
  // (k_fdm_zsolve16) where those lines have 66 .. 128 interior points, an even number of them (option "fdm_z_separate" = 1: A/B)
  bool zsolve = false;
  {
    const int M = pc->geo.dims[d - 1] - 2;
    LineMats &lm = pc->lines[pc->geo.dims[d - 1]];
    zsolve = w_ok && d >= 2 && !opt(OPT_FDM_Z_SEPARATE) && !opt(OPT_NO_RAW_TRANSFORMS) && !opt(OPT_GENERAL_KERNELS) && (M & 1) == 0 &&
             lm.Fraw.KS == FZ_KS && lm.Braw.KS == FZ_KS && lm.Braw.sym == 1 && pc->G * pc->nf / M < 0x7fffffffL - FZ_NT;
  }
  bool scaled = false;
  for (int q = 0; q < (zsolve ? d - 1 : d); q++) {
    const int k = order[q];
    const bool last = q == d - 1 && q > 0 && w_ok;
    int rc = line_transform(pc, k, false, src, a, st, last ? pc->W : nullptr, last ? &scaled : nullptr, q == 0 ? in_mul : nullptr); if (rc) return rc;
    src = a; std::swap(a, b);
  }
  if (!scaled && !zsolve) {
    if (lines == 0) { /* a slab without interior planes: nothing to scale (it still takes part in the transforms along dimension 0) */ }
    else if (d <= 3 && lines <= 65535 && pc->nf <= 65535) {
      const int n1 = d == 3 ? pc->geo.dims[1] - 2 : 1;
      hipLaunchKernelGGL(k_modal_scale3, dim3((unsigned)((nl + 255) / 256), (unsigned)lines, (unsigned)pc->nf), dim3(256), 0, st, d, n1, nl, pc->G,
                         lam.p[0], lam.p[1], lam.p[2], (double *)src);
    } else if (pc->slab) return chebhip_fail(CHEBHIP_ERR_ARG, "slab-mode preconditioner: d <= 3 and at most 65535 lines per slab");
    else
      hipLaunchKernelGGL(k_modal_scale, dim3(pgrid(pc->G * pc->nf)), dim3(256), 0, st, pc->geo, pc->G, pc->nf, lam, (double *)src);
  }
  if (zsolve) {
    const int M = pc->geo.dims[d - 1] - 2;
    LineMats &lm = pc->lines[pc->geo.dims[d - 1]];
    FzParams fp = {};
    fp.M = M; fp.H = (M + 1) / 2; fp.ncols = (unsigned)(pc->G * pc->nf / M); fp.ntiles = (fp.ncols + FZ_NT - 1) / FZ_NT;
    fp.x = src; fp.y = (d == 1) ? z : a;
    fp.d = d; fp.n1 = d == 3 ? pc->geo.dims[1] - 2 : 1; fp.flines = (unsigned)(pc->G / M);
    fp.l0 = lam.p[0]; fp.l1 = lam.p[1]; fp.lz = lam.p[d - 1];
    fp.FE = lm.Fraw.fragE; fp.FO = lm.Fraw.fragO; fp.BE = lm.Braw.fragE; fp.BO = lm.Braw.fragO;
    hipError_t cu_err; const int ncu = sweep_num_cus(&cu_err); PHIPCHK(cu_err);
    const unsigned grid = fp.ntiles < 3u * (unsigned)ncu ? fp.ntiles : 3u * (unsigned)ncu;       // three workgroups per CU
    if (grid) hipLaunchKernelGGL(k_fdm_zsolve16, dim3(grid), dim3(256), 0, st, fp);
    PHIPCHK(hipGetLastError());
    src = a; std::swap(a, b);
  }
  for (int k = (zsolve ? d - 2 : d - 1); k >= 0; k--) {
    double *dst = (k == 0) ? z : a;
    int rc = line_transform(pc, k, true, src, dst, st); if (rc) return rc;
    src = dst; std::swap(a, b);
  }
  PHIPCHK(hipGetLastError());
  return 0;
}

static int fdpc_mult(chebhip_fdpc *pc, const double *x, double *y, hipStream_t st) {
  if (pc->slab) return chebhip_fail(CHEBHIP_ERR_ARG, "chebhip_fdpc_mult: the stencil of P is not available on slabs (no halo exchange)");
  if (!pc->assembled) { int rc = fdpc_update(pc, st); if (rc) return rc; }
  if (pc->G == 0) return 0;
  const long n = pc->G * pc->nf;
  const double *xin = x; double *yout = y;
  if (pc->interleaved) { hipLaunchKernelGGL(k_deinterleave, dim3(pgrid(pc->G)), dim3(256), 0, st, pc->G, pc->nf, x, pc->t2); xin = pc->t2; yout = pc->t3; }
  hipLaunchKernelGGL(k_fd_mult, dim3(pgrid(n)), dim3(256), 0, st, pc->geo, pc->G, pc->nf, (const double *)pc->cf, xin, yout);
  if (pc->interleaved) hipLaunchKernelGGL(k_interleave, dim3(pgrid(pc->G)), dim3(256), 0, st, pc->G, pc->nf, (const double *)pc->t3, y);
  PHIPCHK(hipGetLastError());
  return 0;
}

// component-major callbacks of the inner solve: y = P x and z = P_1^-1 (r / eta)
static int cb_fd_mult(void *ctx, const double *x, double *y, void *stream) {
  chebhip_fdpc *pc = (chebhip_fdpc *)ctx;
  hipLaunchKernelGGL(k_fd_mult, dim3(pgrid(pc->G * pc->nf)), dim3(256), 0, (hipStream_t)stream, pc->geo, pc->G, pc->nf, (const double *)pc->cf, x, y);
  PHIPCHK(hipGetLastError());
  return 0;
}
static int cb_fdm(void *ctx, const double *r, double *z, void *stream) {
  chebhip_fdpc *pc = (chebhip_fdpc *)ctx;
  return fdm_solve(pc, r, z, (hipStream_t)stream, true);
}

static int fdpc_apply(chebhip_fdpc *pc, const double *r, double *z, hipStream_t st) {
  if (!pc->assembled) { int rc = fdpc_update(pc, st); if (rc) return rc; }
  if (pc->slab && pc->sweeps != 0) return chebhip_fail(CHEBHIP_ERR_ARG, "slab-mode preconditioner: sweeps must be 0 (P_1^-1 (r / eta))");
  if (pc->G == 0 && !pc->slab) return 0;              // (a slab without unknowns still takes part in the exchanges of dimension 0)
  const long n = pc->G * pc->nf;
  // component-major copies of r and of the iterate where the ABI vectors are node-major (Stokes velocity)
  const double *rin = r; double *zc = z;
  if (pc->interleaved && pc->sweeps == 0) {    // z = P_1^-1 (r / eta), the division applied while the components are pulled apart
    zc = pc->t4;
    hipLaunchKernelGGL(k_deinterleave_over_eta, dim3(pgrid(pc->G)), dim3(256), 0, st, pc->G, pc->nf, r, (const double *)pc->eta_g, pc->t3);
    int rc = fdm_solve(pc, pc->t3, zc, st); if (rc) return rc;
  } else if (pc->sweeps == 0) {                // z = P_1^-1 (r / eta)
    int rc = cb_fdm(pc, rin, zc, st); if (rc) return rc;
  } else {
    if (pc->interleaved) { hipLaunchKernelGGL(k_deinterleave, dim3(pgrid(pc->G)), dim3(256), 0, st, pc->G, pc->nf, r, pc->t2); rin = pc->t2; zc = pc->t4; }
    // `sweeps` iterations of GMRES on P z = r, right-preconditioned by P_1^-1 (1/eta): monotone in the residual for
    // any coefficient state (a stationary defect correction diverges once eta varies by more than a factor ~2)
    if (!pc->inner || pc->inner_m != pc->sweeps) {
      if (pc->inner) chebhip_fgmres_destroy(pc->inner);
      pc->inner = nullptr;
      int rc = chebhip_fgmres_create(n, pc->sweeps, &pc->inner); if (rc) return rc;
      pc->inner_m = pc->sweeps;
    }
    int rc = chebhip_fgmres_set_tolerances(pc->inner, 1e-12, 1e-300, pc->sweeps); if (rc) return rc;
    if ((rc = chebhip_fgmres_solve(pc->inner, cb_fd_mult, pc, cb_fdm, pc, rin, zc, 0, st))) return rc;
  }
  if (pc->interleaved) hipLaunchKernelGGL(k_interleave, dim3(pgrid(pc->G)), dim3(256), 0, st, pc->G, pc->nf, (const double *)zc, z);
  PHIPCHK(hipGetLastError());
  return 0;
}

// ---- C ABI: scalar elliptic operator -------------------------------------------------------------
extern "C" int ell_pc_create(ell_op *op, chebhip_fdpc **out) {
  if (!op || !out) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  FdView v; int rc = ell_op_fd_view(op, &v); if (rc) return rc;
  rc = fdpc_create(v, 1, false, out); if (rc) return rc;
  (*out)->eop = op;
  return 0;
}

// ---- C ABI: Stokes velocity block (MatVVPC) ------------------------------------------------------
extern "C" int stokes_pc_create(stokes_op *op, chebhip_fdpc **out) {
  if (!op || !out) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  FdView v; int rc = stokes_op_fd_view(op, &v); if (rc) return rc;
  rc = fdpc_create(v, v.d, true, out); if (rc) return rc;
  (*out)->sop = op;
  return 0;
}

// ---- slab mode (called by the slab drivers of slabx.hip) -------------------------------------------------------
static int fdpc_make_slab(chebhip_fdpc *pc, long i0_off, chebhip_fdpc_dim0_fn dim0, void *ctx) {
  if (!dim0 || i0_off < 0) return chebhip_fail(CHEBHIP_ERR_ARG, "slab-mode preconditioner: bad arguments");
  if (pc->geo.d < 2 || pc->geo.d > 3) return chebhip_fail(CHEBHIP_ERR_DIMS, "slab-mode preconditioner: d = 2 or 3");
  pc->slab = true; pc->dim0 = dim0; pc->dim0_ctx = ctx; pc->i0_off = i0_off; pc->sweeps = 0;
  return 0;
}
extern "C" int stokes_pc_create_slab(stokes_op *op, long i0_offset, chebhip_fdpc_dim0_fn dim0, void *ctx, chebhip_fdpc **out) {
  if (!op || !out) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  FdView v; int gP0 = 0; int rc = stokes_op_fd_view_any(op, &v, &gP0); if (rc) return rc;
  rc = fdpc_create(v, v.d, true, out, gP0); if (rc) return rc;
  (*out)->sop = op;
  if ((rc = fdpc_make_slab(*out, i0_offset, dim0, ctx))) { fdpc_free(*out); *out = nullptr; }
  return rc;
}
extern "C" int ell_pc_create_slab(ell_op *op, long i0_offset, chebhip_fdpc_dim0_fn dim0, void *ctx, chebhip_fdpc **out) {
  if (!op || !out) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  FdView v; int gP0 = 0; int rc = ell_op_fd_view_any(op, &v, &gP0); if (rc) return rc;
  rc = fdpc_create(v, 1, false, out, gP0); if (rc) return rc;
  (*out)->eop = op;
  if ((rc = fdpc_make_slab(*out, i0_offset, dim0, ctx))) { fdpc_free(*out); *out = nullptr; }
  return rc;
}
// The line transform along dimension 0 on a pencil: `nfields` stacked arrays (M0, ncol), lines of M0 = dims[0] - 2 interior
// points with stride ncol.  backward = 0: S^-1 (nodal -> modal), 1: S.
extern "C" int chebhip_fdpc_pencil_transform(chebhip_fdpc *pc, int backward, int nfields, long ncol, const double *in_dev, double *out_dev, void *stream) {
  if (!pc || nfields < 1 || ncol < 0 || ((!in_dev || !out_dev) && ncol > 0)) return chebhip_fail(CHEBHIP_ERR_ARG, "bad argument");
  if (ncol == 0) return 0;
  if ((unsigned long long)nfields * (unsigned long long)ncol > 0x7fffffffull) return chebhip_fail(CHEBHIP_ERR_DIMS, "pencil too large");
  return line_transform_g(pc->lines[pc->geo.dims[0]], (unsigned)(nfields * ncol), (unsigned)ncol, backward != 0, in_dev, out_dev, (hipStream_t)stream);
}

extern "C" int chebhip_fdpc_destroy(chebhip_fdpc *pc) { fdpc_free(pc); return 0; }
extern "C" int chebhip_fdpc_update(chebhip_fdpc *pc, void *stream) { if (!pc) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL handle"); return fdpc_update(pc, (hipStream_t)stream); }
extern "C" int chebhip_fdpc_set_sweeps(chebhip_fdpc *pc, int sweeps) {
  if (!pc || sweeps < 0 || sweeps > 64) return chebhip_fail(CHEBHIP_ERR_ARG, "sweeps must be in 0..64");
  if (pc->slab && sweeps != 0) return chebhip_fail(CHEBHIP_ERR_ARG, "slab-mode preconditioner: sweeps must be 0");
  pc->sweeps = sweeps; return 0;
}
extern "C" int chebhip_fdpc_mult(chebhip_fdpc *pc, const double *x, double *y, void *stream) {
  if (!pc || ((!x || !y) && pc->G)) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (x && x == y) return chebhip_fail(CHEBHIP_ERR_ARG, "x and y must be distinct");
  return fdpc_mult(pc, x, y, (hipStream_t)stream);
}
// Stokes velocity preconditioner on component-major vectors: r, z are nf = d stacked interior fields, which is the layout the
// line transforms work in -- z = P_1^-1 (r / eta) with no (de)interleaving pass on either side
extern "C" int chebhip_fdpc_apply_cm(void *ctx, const double *r, double *z, void *stream) {
  chebhip_fdpc *pc = (chebhip_fdpc *)ctx;
  if (!pc || ((!r || !z) && pc->G)) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (r && r == z) return chebhip_fail(CHEBHIP_ERR_ARG, "r and z must be distinct");
  if (!pc->interleaved || pc->sweeps != 0) return chebhip_fail(CHEBHIP_ERR_ARG, "chebhip_fdpc_apply_cm: a Stokes velocity preconditioner with sweeps = 0");
  StageTimer tm(CHEBHIP_STAGE_FDPC_APPLY, stream);
  hipStream_t st = (hipStream_t)stream;
  if (!pc->assembled) { int rc = fdpc_update(pc, st); if (rc) return rc; }
  if (pc->G == 0 && !pc->slab) return 0;
  return cb_fdm(pc, r, z, st);
}
extern "C" int chebhip_fdpc_apply(void *ctx, const double *r, double *z, void *stream) {
  chebhip_fdpc *pc = (chebhip_fdpc *)ctx;
  if (!pc || ((!r || !z) && pc->G)) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (r && r == z) return chebhip_fail(CHEBHIP_ERR_ARG, "r and z must be distinct");
  StageTimer tm(CHEBHIP_STAGE_FDPC_APPLY, stream);
  return fdpc_apply(pc, r, z, (hipStream_t)stream);
}
