// saddle.hip -- the block preconditioners of the Stokes saddle-point system on the device:
// StokesPCApply0..3 (stokes.C:1714-1817) composed from the operator's own blocks (stokes_op_mult_vv / _pv / _vp /
// _schur), the finite-difference velocity matrix MatVVPC (precond.hip) and three inner Krylov solves with the
// roles the reference gives its KSPs (StokesCreate, stokes.C:328-341):
//   KSPVelocity       operators (MatVV, MatVVPC), prefix vel_    -> flexible GMRES on MatVV, M = MatVVPC solve
//   KSPSchur          operator MatSchur, PCJACOBI with the diagonal StokesMatGetDiagonalSchur supplies -- 1 / eta at the
//                     interior nodes (stokes.C:330-331, :538-553): every residual is multiplied by the viscosity --,
//                     constant null space removed from every Krylov vector (stokes.C:1020-1021), prefix schur_
//   KSPSchurVelocity  operators (MatVV, MatVVPC), prefix svel_   -> inside StokesMatMultSchur (stokes.C:531)
// Defaults follow the options the reference's README recommends (README:43): -vel_ksp_max_it 4,
// -schur_ksp_max_it 3, -svel_ksp_type preonly (one application of the MatVVPC solve).
// Everything here goes through the public C ABI of the operator: vectors never leave HBM.
#include "../../include/chebhip.h"
#include "ops.h"
#include "sweep.h"
#include "timers.h"
#include <hip/hip_runtime.h>
#include <new>

int chebhip_fail(int code, const char *fmt, ...);   // chebhip.hip

#define SHIPCHK2(expr)                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {
static inline unsigned sgrid(long n) { long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
#define GS_LOOP(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// full global vector (I nodes x [v_0 .. v_{d-1}, p]) <-> velocity (I x d) and pressure (I) parts: scatterGV / GP / VG / PG
// cm: the velocity part is component-major (component c of node n at c I + n; the default of the inner solves, see stokes_saddle)
__global__ void k_split(long I, int d, const double *__restrict__ x, double *__restrict__ v, double *__restrict__ p, int cm) {
  GS_LOOP(q, I * (d + 1)) {
    const long n = q / (d + 1); const int c = (int)(q - n * (d + 1));
    if (c < d) { if (v) v[cm ? c * I + n : n * d + c] = x[q]; } else if (p) p[n] = x[q];
  }
}
// y_v = (addv ? y_v : 0) + v,  y_p = p   (either part may be null: left untouched)
__global__ void k_merge(long I, int d, const double *__restrict__ v, int addv, const double *__restrict__ p, double *__restrict__ y, int cm) {
  GS_LOOP(q, I * (d + 1)) {
    const long n = q / (d + 1); const int c = (int)(q - n * (d + 1));
    if (c < d) { if (v) { const double w = v[cm ? c * I + n : n * d + c]; y[q] = addv ? y[q] + w : w; } }
    else if (p) y[q] = p[n];
  }
}
// a = s * a + (b ? b : 0)
__global__ void k_scale_add(long n, double s, double *__restrict__ a, const double *__restrict__ b) { GS_LOOP(q, n) a[q] = s * a[q] + (b ? b[q] : 0.0); }
// PCApply of KSPSchur: PCJACOBI divides by the diagonal 1 / eta (StokesMatGetDiagonalSchur, stokes.C:538-553: scatterLP of
// eta, VecReciprocal), i.e. out = eta .* in at the interior nodes.  in == out allowed.
__global__ void k_eta_scale(long N, const int *__restrict__ ixL, const double *__restrict__ eta, const double *in, double *out) {
  GS_LOOP(l, N) { const int n = ixL[l]; if (n >= 0) out[n] = eta[l] * in[n]; }
}
// out = in - mean(in): MatNullSpaceRemove with the constant vector, in two launches with a fixed summation order
// (256 chunk sums, then every block adds the 256 partials in the same order).  The first version summed the vector in ONE
// workgroup: 1.08 ms per call at 128^3, 39 % of the device time of the config-5 solve (profiles/r02_stokes_power_kernel_summary.txt).
constexpr int MEAN_RB = 256, MEAN_RT = 256;
__global__ __launch_bounds__(MEAN_RT) void k_mean_partial(long n, const double *__restrict__ in, double *__restrict__ part) {
  __shared__ double sh[MEAN_RT];
  double s = 0.0;
  for (long q = blockIdx.x * (long)MEAN_RT + threadIdx.x; q < n; q += (long)MEAN_RB * MEAN_RT) s += in[q];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = MEAN_RT / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
// Slab mode: the mean is over the pressure unknowns of ALL ranks.  tot[0] = this rank's sum (the 256 partials added in a fixed
// order), tot[1] = its count; after the all-reduce of the two values every rank subtracts tot[0] / tot[1].
__global__ __launch_bounds__(MEAN_RT) void k_mean_total(long n, const double *__restrict__ part, double *__restrict__ tot) {
  __shared__ double sh[MEAN_RB];
  sh[threadIdx.x] = part[threadIdx.x];
  __syncthreads();
  for (int o = MEAN_RB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) { tot[0] = sh[0]; tot[1] = (double)n; }
}
__global__ void k_mean_subtract_g(long n, const double *__restrict__ tot, const double *in, double *out) {   // in == out allowed
  const double mean = tot[0] / tot[1];
  GS_LOOP(q, n) out[q] = in[q] - mean;
}
__global__ __launch_bounds__(MEAN_RT) void k_mean_subtract(long n, const double *__restrict__ part, const double *in, double *out) {   // in == out allowed
  __shared__ double sh[MEAN_RB];
  sh[threadIdx.x] = part[threadIdx.x];
  __syncthreads();
  for (int o = MEAN_RB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  const double mean = sh[0] / (double)n;
  GS_LOOP(q, n) out[q] = in[q] - mean;
}
}  // namespace

struct stokes_saddle {
  stokes_op *op = nullptr;
  chebhip_fdpc *vvpc = nullptr;
  int d = 0; long I = 0, gv = 0, gp = 0, g = 0;
  int type = 0;                                    // -pc_saddle_type (stokes.C:177-187)
  double *v0 = nullptr, *v1 = nullptr, *p0 = nullptr, *p1 = nullptr;      // vG0, vG1, pG0, pG1 (stokes.C:56-57)
  double *red = nullptr;                           // partial sums of remove_mean
  chebhip_fgmres *kvel = nullptr, *kschur = nullptr, *ksvel = nullptr;
  // restart / max_it / rtol per inner solve; max_it 0 = "preonly": one application of the preconditioner
  int m_vel = 4, m_schur = 3, m_svel = 0;
  double rtol_vel = 1e-5, rtol_schur = 1e-5, rtol_svel = 1e-5;
  int its_vel = 0, its_schur = 0;                  // operator applies of the last apply, for monitoring
  chebhip::FdView view;                            // ixL and eta of the operator (the Jacobi scaling of KSPSchur)
  bool schur_jacobi = true;
  // slab mode: the vectors are this rank's pieces; inner products, norms and the pressure mean are completed over
  // the ranks by `reduce`; MatVVPC is the slab driver's (borrowed: chebhip_dist_stokes_pc)
  bool slab = false, own_pc = true;
  chebhip_reduce_fn reduce = nullptr; void *reduce_ctx = nullptr;
  // The velocity vectors INSIDE an apply (v0, v1, the Krylov bases of the two velocity solves, the work vectors of MatSchur) are
  // component-major: d stacked scalar fields, the layout MatVVPC's line transforms work in, so that no inner iteration pays a
  // (de)interleaving pass around its preconditioner solve (stokes_op_mult_*_cm, chebhip_fdpc_apply_cm).  The full vectors at
  // the interface keep the reference's node-major layout: split / merge convert.  Inner products run over the same values in
  // another order: the iterates agree with the node-major route to rounding (option "saddle_node_major" = 1 at create: A/B).
  bool cm = true, cm_opt = true;       // cm_opt: the option at create; cm = cm_opt and MatVVPC without inner sweeps (the only component-major solve)
};

static int vv_apply(void *ctx, const double *x, double *y, void *stream) {
  stokes_saddle *s = (stokes_saddle *)ctx;
  return s->cm ? stokes_op_mult_vv_cm(s->op, x, y, stream) : stokes_op_mult_vv(s->op, x, y, stream);
}
static inline chebhip_apply_fn pc_fn(const stokes_saddle *s) { return s->cm ? chebhip_fdpc_apply_cm : chebhip_fdpc_apply; }
// KSPSolve(KSPSchurVelocity) (stokes.C:531): preonly -> one MatVVPC solve; otherwise GMRES on MatVV with it
static int svel_solve(void *ctx, const double *rhs, double *sol, void *stream) {
  stokes_saddle *s = (stokes_saddle *)ctx;
  if (s->m_svel == 0) return pc_fn(s)(s->vvpc, rhs, sol, stream);
  int rc = chebhip_fgmres_set_tolerances(s->ksvel, s->rtol_svel, 1e-50, s->m_svel); if (rc) return rc;
  return chebhip_fgmres_solve(s->ksvel, vv_apply, s, pc_fn(s), s->vvpc, rhs, sol, 0, stream);
}
// MatSchur followed by the removal of the constant: KSPSchur carries the constant null space (stokes.C:1020-1021) and
// PETSc's (left-preconditioned) GMRES removes it from every vector it builds, i.e. it solves  P S x = P b  on
// zero-mean vectors, P = I - 1 1^T / n.  (S P z = b would be inconsistent: S is singular and not symmetric.)
static int remove_mean(stokes_saddle *s, const double *in, double *out, hipStream_t st) {
  unsigned g = (unsigned)((s->gp + MEAN_RT - 1) / MEAN_RT); if (g < 1) g = 1; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_mean_partial, dim3(MEAN_RB), dim3(MEAN_RT), 0, st, s->gp, in, s->red);
  if (s->slab && s->reduce) {
    hipLaunchKernelGGL(k_mean_total, dim3(1), dim3(MEAN_RT), 0, st, s->gp, (const double *)s->red, s->red + MEAN_RB);
    int rc = s->reduce(s->reduce_ctx, s->red + MEAN_RB, 2, (void *)st); if (rc) return rc;
    hipLaunchKernelGGL(k_mean_subtract_g, dim3(g), dim3(MEAN_RT), 0, st, s->gp, (const double *)(s->red + MEAN_RB), in, out);
    return 0;
  }
  hipLaunchKernelGGL(k_mean_subtract, dim3(g), dim3(MEAN_RT), 0, st, s->gp, (const double *)s->red, in, out);
  return 0;
}
static void eta_scale(stokes_saddle *s, const double *in, double *out, hipStream_t st) {
  if (s->schur_jacobi) hipLaunchKernelGGL(k_eta_scale, dim3(sgrid(s->view.N)), dim3(256), 0, st, s->view.N, s->view.ixL, s->view.eta, in, out);
}
// One step of KSPSchur's left-preconditioned GMRES (PETSc's default side): y = P (eta .* (S x))
static int schur_apply(void *ctx, const double *x, double *y, void *stream) {
  stokes_saddle *s = (stokes_saddle *)ctx;
  int rc = s->cm ? stokes_op_mult_schur_cm(s->op, x, y, svel_solve, s, stream) : stokes_op_mult_schur(s->op, x, y, svel_solve, s, stream);
  if (rc) return rc;
  eta_scale(s, y, y, (hipStream_t)stream);
  if ((rc = remove_mean(s, y, y, (hipStream_t)stream))) return rc;
  SHIPCHK2(hipGetLastError());
  return 0;
}
// KSPSolve(KSPVelocity, b, x)
static int vel_solve(stokes_saddle *s, const double *b, double *x, void *stream) {
  if (s->m_vel == 0) return pc_fn(s)(s->vvpc, b, x, stream);
  int rc = chebhip_fgmres_set_tolerances(s->kvel, s->rtol_vel, 1e-50, s->m_vel); if (rc) return rc;
  rc = chebhip_fgmres_solve(s->kvel, vv_apply, s, pc_fn(s), s->vvpc, b, x, 0, stream);
  s->its_vel += chebhip_fgmres_iterations(s->kvel);
  return rc;
}
// KSPSolve(KSPSchur, b, x): GMRES on P diag(eta) S x = P diag(eta) b over zero-mean vectors, converged on the
// preconditioned residual as PETSc's left-preconditioned GMRES is; b is overwritten by its preconditioned zero-mean part
static int schur_solve(stokes_saddle *s, double *b, double *x, void *stream) {
  eta_scale(s, b, b, (hipStream_t)stream);
  int rc = remove_mean(s, b, b, (hipStream_t)stream); if (rc) return rc;
  rc = chebhip_fgmres_set_tolerances(s->kschur, s->rtol_schur, 1e-50, s->m_schur > 0 ? s->m_schur : 1); if (rc) return rc;
  rc = chebhip_fgmres_solve(s->kschur, schur_apply, s, nullptr, nullptr, b, x, 0, stream);
  s->its_schur += chebhip_fgmres_iterations(s->kschur);
  return rc;
}

extern "C" int stokes_saddle_destroy(stokes_saddle *s) {
  if (!s) return 0;
  if (s->kvel) chebhip_fgmres_destroy(s->kvel);
  if (s->kschur) chebhip_fgmres_destroy(s->kschur);
  if (s->ksvel) chebhip_fgmres_destroy(s->ksvel);
  if (s->vvpc && s->own_pc) chebhip_fdpc_destroy(s->vvpc);
  double *all[] = {s->v0, s->v1, s->p0, s->p1, s->red};
  for (double *p : all) if (p) (void)hipFree(p);
  delete s;
  return 0;
}

static int saddle_create(stokes_op *op, chebhip_fdpc *slab_pc, chebhip_reduce_fn reduce, void *reduce_ctx, stokes_saddle **out);
extern "C" int stokes_saddle_create(stokes_op *op, stokes_saddle **out) { return saddle_create(op, nullptr, nullptr, nullptr, out); }
// On a slab-mode operator (chebhip_dist_stokes_op) with the slab driver's MatVVPC (chebhip_dist_stokes_pc; borrowed): the three
// inner Krylov solves complete their inner products through `reduce` (chebhip_comm_reduce with the driver's communicator), and so
// does the removal of the constant pressure mode.  Every rank calls stokes_saddle_apply collectively.
extern "C" int stokes_saddle_create_slab(stokes_op *slab_op, chebhip_fdpc *slab_pc, chebhip_reduce_fn reduce, void *reduce_ctx, stokes_saddle **out) {
  if (!slab_pc) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  return saddle_create(slab_op, slab_pc, reduce, reduce_ctx, out);
}
static int saddle_create(stokes_op *op, chebhip_fdpc *slab_pc, chebhip_reduce_fn reduce, void *reduce_ctx, stokes_saddle **out) {
  if (!op || !out) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  *out = nullptr;
  stokes_saddle *s = new (std::nothrow) stokes_saddle;
  if (!s) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  s->op = op; s->cm_opt = s->cm = chebhip::opt(chebhip::OPT_SADDLE_NODE_MAJOR) == 0;
  if (slab_pc) { s->slab = true; s->own_pc = false; s->vvpc = slab_pc; s->reduce = reduce; s->reduce_ctx = reduce_ctx; }
  s->I = stokes_op_size(op, 1); s->gv = stokes_op_size(op, 2); s->gp = stokes_op_size(op, 3); s->g = stokes_op_size(op, 4);
  s->d = s->I > 0 ? (int)(s->gv / s->I) : 2;
  int rc = s->slab ? 0 : stokes_pc_create(op, &s->vvpc);
  if (!rc) rc = stokes_op_fd_view_any(op, &s->view, nullptr);
  if (!rc) rc = chebhip_fdpc_set_sweeps(s->vvpc, 0);
  if (!rc) rc = chebhip_fgmres_create(s->gv, 30, &s->kvel);
  if (!rc) rc = chebhip_fgmres_create(s->gp, 30, &s->kschur);
  if (!rc) rc = chebhip_fgmres_create(s->gv, 30, &s->ksvel);
  if (!rc && s->slab && reduce) {
    rc = chebhip_fgmres_set_reduce(s->kvel, reduce, reduce_ctx);
    if (!rc) rc = chebhip_fgmres_set_reduce(s->kschur, reduce, reduce_ctx);
    if (!rc) rc = chebhip_fgmres_set_reduce(s->ksvel, reduce, reduce_ctx);
  }
  if (rc) { stokes_saddle_destroy(s); return rc; }
  const size_t nv = (size_t)(s->gv > 0 ? s->gv : 1) * sizeof(double), np = (size_t)(s->gp > 0 ? s->gp : 1) * sizeof(double);
  if (hipMalloc((void **)&s->v0, nv) != hipSuccess || hipMalloc((void **)&s->v1, nv) != hipSuccess ||
      hipMalloc((void **)&s->p0, np) != hipSuccess || hipMalloc((void **)&s->p1, np) != hipSuccess ||
      hipMalloc((void **)&s->red, (MEAN_RB + 2) * sizeof(double)) != hipSuccess) {
    stokes_saddle_destroy(s); return chebhip_fail(CHEBHIP_ERR_MEMORY, "device allocation failed");
  }
  *out = s;
  return 0;
}

extern "C" int stokes_saddle_set_type(stokes_saddle *s, int type) {
  if (!s || type < 0 || type > 3) return chebhip_fail(CHEBHIP_ERR_ARG, "pc_saddle_type %d not implemented (stokes.C:186)", type);
  s->type = type; return 0;
}
// which: 0 KSPVelocity (vel_), 1 KSPSchur (schur_), 2 KSPSchurVelocity (svel_).  max_it = 0 for the two velocity
// solves means -ksp_type preonly: one application of the MatVVPC solve
extern "C" int stokes_saddle_set_inner(stokes_saddle *s, int which, int max_it, double rtol) {
  if (!s || which < 0 || which > 2 || max_it < 0 || max_it > 100000 || !(rtol >= 0.0)) return chebhip_fail(CHEBHIP_ERR_ARG, "bad inner-solver setting");
  if (which == 0) { s->m_vel = max_it; s->rtol_vel = rtol; }
  else if (which == 1) { s->m_schur = max_it; s->rtol_schur = rtol; }
  else { s->m_svel = max_it; s->rtol_svel = rtol; }
  return 0;
}
// StokesPCSetUp0 (stokes.C:1160-1241): MatVVPC from the current eta -- call after stokes_op_function / set_state
extern "C" int stokes_saddle_setup(stokes_saddle *s, void *stream) {
  if (!s) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL handle");
  return chebhip_fdpc_update(s->vvpc, stream);
}
// inner GMRES steps on MatVVPC per application of its approximate solve (chebhip_fdpc_set_sweeps; default 0: the fast
// diagonalisation alone, exact for constant viscosity -- raise it when the viscosity varies strongly)
extern "C" int stokes_saddle_set_pc_sweeps(stokes_saddle *s, int sweeps) {
  if (!s) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL handle");
  int rc = chebhip_fdpc_set_sweeps(s->vvpc, sweeps); if (rc) return rc;
  s->cm = s->cm_opt && sweeps == 0;
  return 0;
}
// -schur_pc_type: 1 = jacobi (the reference's hard-wired choice, stokes.C:330-331), 0 = none
extern "C" int stokes_saddle_set_schur_jacobi(stokes_saddle *s, int on) {
  if (!s) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL handle");
  s->schur_jacobi = on != 0;
  return 0;
}
extern "C" int stokes_saddle_iterations(const stokes_saddle *s, int which) { return !s ? -1 : (which == 0 ? s->its_vel : s->its_schur); }

// StokesPCApply0..3 (stokes.C:1714-1817): y = M^-1 x on full global vectors; shape of chebhip_apply_fn
extern "C" int stokes_saddle_apply(void *ctx, const double *x, double *y, void *stream) {
  stokes_saddle *s = (stokes_saddle *)ctx;
  if (!s || ((!x || !y) && s->g)) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (s->g == 0 && !s->slab) return 0;               // (a slab without unknowns still takes part in the collectives)
  chebhip::StageTimer tm(CHEBHIP_STAGE_SADDLE_APPLY, stream);
  hipStream_t st = (hipStream_t)stream;
  const long I = s->I; const int d = s->d, cm = s->cm ? 1 : 0;
  const unsigned gg = sgrid(s->g), gpn = sgrid(s->gp), gvn = sgrid(s->gv);
  s->its_vel = s->its_schur = 0;
  int rc;
#define SPLIT(V, P) hipLaunchKernelGGL(k_split, dim3(gg), dim3(256), 0, st, I, d, x, V, P, cm)
#define MERGE(V, ADD, P) hipLaunchKernelGGL(k_merge, dim3(gg), dim3(256), 0, st, I, d, (const double *)(V), ADD, (const double *)(P), y, cm)
  auto mult_pv = [&](const double *v, double *p) { return s->cm ? stokes_op_mult_pv_cm(s->op, v, p, stream) : stokes_op_mult_pv(s->op, v, p, stream); };
  auto mult_vp = [&](const double *p, double *v) { return s->cm ? stokes_op_mult_vp_cm(s->op, p, v, stream) : stokes_op_mult_vp(s->op, p, v, stream); };
  switch (s->type) {
    case 0:   // block LU (stokes.C:1714-1740)
      SPLIT(s->v0, (double *)nullptr);                                             // scatterGV: v0 <- x_v
      if ((rc = vel_solve(s, s->v0, s->v1, stream))) return rc;                    // v1 <- A^-1 v0          (:1723)
      MERGE(s->v1, 0, nullptr);                                                    // y_v <- v1              (:1725)
      if ((rc = mult_pv(s->v1, s->p0))) return rc;        // p0 <- B v1             (:1726)
      SPLIT((double *)nullptr, s->p1);                                             // x_p
      hipLaunchKernelGGL(k_scale_add, dim3(gpn), dim3(256), 0, st, s->gp, -1.0, s->p0, (const double *)s->p1);   // p0 <- -p0 + x_p (:1727-1729)
      if ((rc = schur_solve(s, s->p0, s->p1, stream))) return rc;                  // p1 <- S^-1 p0          (:1732)
      if ((rc = mult_vp(s->p1, s->v0))) return rc;        // v0 <- B^T p1           (:1734)
      hipLaunchKernelGGL(k_scale_add, dim3(gvn), dim3(256), 0, st, s->gv, -1.0, s->v0, (const double *)nullptr);   // v0 <- -v0 (:1735)
      if ((rc = vel_solve(s, s->v0, s->v1, stream))) return rc;                    // v1 <- A^-1 v0          (:1736)
      if ((rc = remove_mean(s, s->p1, s->p1, st))) return rc;      // KSPSetNullSpace (:1019)
      MERGE(s->v1, 1, s->p1);                                                      // y_v += v1, y_p <- p1   (:1733,1737)
      break;
    case 1:   // block upper triangular (stokes.C:1747-1765)
      SPLIT(s->v1, s->p0);
      if ((rc = schur_solve(s, s->p0, s->p1, stream))) return rc;                  // p1 <- S^-1 p0
      if ((rc = mult_vp(s->p1, s->v0))) return rc;        // v0 <- B^T p1
      hipLaunchKernelGGL(k_scale_add, dim3(gvn), dim3(256), 0, st, s->gv, -1.0, s->v0, (const double *)s->v1);   // v0 <- -v0 + x_v
      if ((rc = vel_solve(s, s->v0, s->v1, stream))) return rc;
      if ((rc = remove_mean(s, s->p1, s->p1, st))) return rc;
      MERGE(s->v1, 0, s->p1);
      break;
    case 2:   // block diagonal (stokes.C:1772-1790)
      SPLIT(s->v0, s->p0);
      if ((rc = vel_solve(s, s->v0, s->v1, stream))) return rc;
      if ((rc = schur_solve(s, s->p0, s->p1, stream))) return rc;
      if ((rc = remove_mean(s, s->p1, s->p1, st))) return rc;
      MERGE(s->v1, 0, s->p1);
      break;
    default:  // block lower triangular (stokes.C:1797-1816)
      SPLIT(s->v0, s->p1);
      if ((rc = vel_solve(s, s->v0, s->v1, stream))) return rc;                    // v1 <- A^-1 v0
      if ((rc = mult_pv(s->v1, s->p0))) return rc;        // p0 <- B v1
      hipLaunchKernelGGL(k_scale_add, dim3(gpn), dim3(256), 0, st, s->gp, -1.0, s->p0, (const double *)s->p1);   // p0 <- -p0 + x_p
      MERGE(s->v1, 0, nullptr);                                                    // y_v <- v1 (before v1 is reused by the Schur solve)
      if ((rc = schur_solve(s, s->p0, s->p1, stream))) return rc;                  // p1 <- S^-1 p0
      if ((rc = remove_mean(s, s->p1, s->p1, st))) return rc;
      MERGE((const double *)nullptr, 0, s->p1);
      break;
  }
#undef SPLIT
#undef MERGE
  SHIPCHK2(hipGetLastError());
  return 0;
}
