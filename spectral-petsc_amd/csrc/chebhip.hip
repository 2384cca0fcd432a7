// chebhip.hip -- C ABI of libchebhip.so (see include/chebhip.h) and the small pointwise kernels
// that sit between sweeps.  All heavy arithmetic is in sweep.hip.
#include "../../include/chebhip.h"
#include "sweep.h"
#include "ops.h"
#include "timers.h"
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace chebhip;

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
// shared with stokes.hip
int chebhip_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char *chebhip_last_error(void) { return g_err.c_str(); }
extern "C" int chebhip_version(void) { return 100; }
extern "C" const char *chebhip_arch(void) { return "gfx950"; }
extern "C" long chebhip_launch_count(void) { return sweep_launch_count(); }

// ---------------------------------------------------------------------------------------------
// run-time options (include/chebhip.h; registry in options.cpp): the only switches of the library; the environment is never read
// ---------------------------------------------------------------------------------------------
extern "C" int chebhip_set_option(const char *name, int value) {
  if (!name) return fail(CHEBHIP_ERR_ARG, "NULL option name");
  if (!strcmp(name, "no_rocblas")) { opt_set(OPT_VENDOR_GEMM, value ? 0 : 1); return 0; }   // deprecated name (rounds 1-3), inverted meaning
  const int id = opt_find(name);
  if (id < 0) return fail(CHEBHIP_ERR_ARG, "unknown option '%s'", name);
  opt_set(id, value);
  return 0;
}
extern "C" int chebhip_get_option(const char *name, int *value) {
  if (!name || !value) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (!strcmp(name, "no_rocblas")) { *value = opt(OPT_VENDOR_GEMM) ? 0 : 1; return 0; }
  const int id = opt_find(name);
  if (id < 0) return fail(CHEBHIP_ERR_ARG, "unknown option '%s'", name);
  *value = opt(id);
  return 0;
}
extern "C" const char *chebhip_option_name(int index) { return opt_name(index); }
// Diagnostic builds (-DCHEB_STAMPS) only: device buffer of 256*8*4 uint64 receiving per-wave phase cycle sums.
static const double *g_stamp_buf = nullptr;
static int g_stamp_cnt = 0;
extern "C" void chebhip_debug_stamp_buffer(const void *dev) { g_stamp_buf = (const double *)dev; g_stamp_cnt = 0; }
const double *chebhip_stamp_buf() { return g_stamp_buf; }
int chebhip_stamp_next() { return g_stamp_cnt++; }

// ---------------------------------------------------------------------------------------------
// per-stage device timers (include/chebhip.h "Instrumentation")
// ---------------------------------------------------------------------------------------------
namespace {
struct Pending { int id; hipEvent_t e0, e1; };
std::mutex g_tm_mu;
bool g_tm_on = false;
std::vector<Pending> g_tm_pending;
double g_tm_ms[CHEBHIP_NSTAGES] = {0};
long g_tm_calls[CHEBHIP_NSTAGES] = {0};
const char *const g_tm_names[CHEBHIP_NSTAGES] = {
  "cheb_apply", "ell_op_mult", "ell_op_function", "stokes_op_mult", "stokes_op_mult_vv", "stokes_op_mult_pv", "stokes_op_mult_vp",
  "stokes_op_function", "stokes_op_mult_schur", "chebhip_fdpc_apply", "stokes_saddle_apply", "chebhip_fgmres_solve"};
void tm_drain_locked() {
  for (auto &p : g_tm_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) { g_tm_ms[p.id] += ms; g_tm_calls[p.id]++; }
    (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1);
  }
  g_tm_pending.clear();
}
}  // namespace
chebhip::StageTimer::StageTimer(int stage, void *stream) : id(stage), st((hipStream_t)stream), on(g_tm_on) {
  if (!on) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventRecord(e0, st) != hipSuccess) { on = false; if (e0) (void)hipEventDestroy(e0); }
}
chebhip::StageTimer::~StageTimer() {
  if (!on) return;
  hipEvent_t e1 = nullptr;
  if (hipEventCreate(&e1) != hipSuccess || hipEventRecord(e1, st) != hipSuccess) { (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); return; }
  std::lock_guard<std::mutex> lk(g_tm_mu);
  g_tm_pending.push_back({id, e0, e1});
  if (g_tm_pending.size() > 4096) tm_drain_locked();     // bound the number of live events
}
extern "C" int chebhip_timers_enable(int on) { std::lock_guard<std::mutex> lk(g_tm_mu); g_tm_on = on != 0; return 0; }
extern "C" int chebhip_timers_reset(void) {
  std::lock_guard<std::mutex> lk(g_tm_mu);
  tm_drain_locked();
  for (int i = 0; i < CHEBHIP_NSTAGES; i++) { g_tm_ms[i] = 0.0; g_tm_calls[i] = 0; }
  return 0;
}
extern "C" int chebhip_timers_read(int stage, double *total_ms, long *calls) {
  if (stage < 0 || stage >= CHEBHIP_NSTAGES || !total_ms || !calls) return fail(CHEBHIP_ERR_ARG, "stage %d out of range", stage);
  std::lock_guard<std::mutex> lk(g_tm_mu);
  tm_drain_locked();
  *total_ms = g_tm_ms[stage]; *calls = g_tm_calls[stage];
  return 0;
}
extern "C" const char *chebhip_stage_name(int stage) { return (stage >= 0 && stage < CHEBHIP_NSTAGES) ? g_tm_names[stage] : ""; }

static int require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(CHEBHIP_ERR_DEVICE, "no usable HIP device (%s); libchebhip has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// kernel-level plan (MatCreateCheb / ChebMult / ChebDestroy, chebyshev.c:89-235)
struct cheb_plan {
  int rank = 0, tr = 0;
  std::vector<int> dims;
  long N = 0;
  unsigned inner = 1, ncols = 0;
  DiffMat mat;
  DiffMat lap;                          // trimmed plans: interior D D
  bool trimmed = false;                 // created by cheb_plan_create_trimmed
  double *hx = nullptr, *hy = nullptr;  // staging for the host-pointer path
};

static int check_geom(int rank, int tr, const int *dims, long *N, unsigned *inner, bool allow_long = false) {
  if (!dims || rank < 1 || rank > 16) return fail(CHEBHIP_ERR_DIMS, "rank = %d must be in 1..16", rank);
  if (!(0 <= tr && tr < rank)) return fail(CHEBHIP_ERR_TDIM, "tdim out of range");              // chebyshev.c:106
  long n = 1, in = 1;
  for (int r = 0; r < rank; r++) {
    if (dims[r] < 1) return fail(CHEBHIP_ERR_DIMS, "dims[%d] = %d must be >= 1", r, dims[r]);
    n *= dims[r];
    if (r > tr) in *= dims[r];
    if (n > 0x7fffffffL) return fail(CHEBHIP_ERR_DIMS, "tensor of more than 2^31-1 points");
  }
  if (n < 2 || dims[tr] < 2) return fail(CHEBHIP_ERR_SIZE, "n = %ld but must be >= 2", n);       // chebyshev.c:18,98
  if (dims[tr] > 256 && !allow_long)
    return fail(CHEBHIP_ERR_ARG, "dims[tr] = %d: the interior-layout (slab) plans support <= 256 points per line", dims[tr]);
  if (dims[tr] > 4096) return fail(CHEBHIP_ERR_ARG, "dims[tr] = %d: at most 4096 points per line", dims[tr]);
  *N = n; *inner = (unsigned)in;
  return 0;
}

extern "C" int cheb_plan_create(int rank, int tr, const int *dims, cheb_plan **out) {
  if (!out) return fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  long N; unsigned inner;
  int rc = check_geom(rank, tr, dims, &N, &inner, true);
  if (rc) return rc;
  if ((rc = require_device())) return rc;
  cheb_plan *p = new (std::nothrow) cheb_plan;
  if (!p) return fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  p->rank = rank; p->tr = tr; p->dims.assign(dims, dims + rank);
  p->N = N; p->inner = inner; p->ncols = (unsigned)(N / dims[tr]);
  hipError_t e = hipSuccess;
  e = diffmat_create(dims[tr], &p->mat);          // > 256 points: dense matrix for cheb_sweep_long_kernel
  if (e != hipSuccess) { delete p; return fail(CHEBHIP_ERR_DEVICE, "plan matrices: %s", hipGetErrorString(e)); }
  *out = p;
  return 0;
}

extern "C" long cheb_plan_size(const cheb_plan *p) { return p ? p->N : -1; }

// Plan on a tensor that stores only the interior points 1..P-2 of every line along `tr`
// (dims[tr] = P-2 stored points, the two end points are implicit zeros): the layout of the
// reference's global vectors (SetupBC, elliptic.C:372-434) and of any slab or pencil cut from them.
extern "C" int cheb_plan_create_trimmed(int rank, int tr, const int *dims, cheb_plan **out) {
  if (!out) return fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!dims || rank < 1 || rank > 16) return fail(CHEBHIP_ERR_DIMS, "rank = %d must be in 1..16", rank);
  if (!(0 <= tr && tr < rank)) return fail(CHEBHIP_ERR_TDIM, "tdim out of range");
  std::vector<int> full(dims, dims + rank);
  if (dims[tr] < 1) return fail(CHEBHIP_ERR_SIZE, "dims[tr] = %d stored points but must be >= 1", dims[tr]);
  full[tr] = dims[tr] + 2;
  long N; unsigned inner;
  int rc = check_geom(rank, tr, full.data(), &N, &inner);
  if (rc) return rc;
  if ((rc = require_device())) return rc;
  cheb_plan *p = new (std::nothrow) cheb_plan;
  if (!p) return fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  p->rank = rank; p->tr = tr; p->dims.assign(dims, dims + rank); p->trimmed = true;
  p->N = N / full[tr] * dims[tr]; p->inner = inner; p->ncols = (unsigned)(N / full[tr]);
  hipError_t e = diffmat_create(full[tr], &p->mat);
  if (e == hipSuccess) e = diffmat_create_lap(full[tr], &p->lap);
  if (e != hipSuccess) { diffmat_destroy(&p->mat); delete p; return fail(CHEBHIP_ERR_DEVICE, "diffmat_create: %s", hipGetErrorString(e)); }
  *out = p;
  return 0;
}

// y = acc + alpha * (D_tr D_tr x) at the stored (interior) points, x extended by zero end points:
// one direction of the linear MatMult_Elliptic (elliptic.C:309-334 with eta = 1, deta = 0).
// acc may be NULL (treated as 0) and may alias y.
extern "C" int cheb_apply_lap1d(cheb_plan *p, const double *x, const double *acc, double alpha,
                                double *y, void *stream) {
  if (!p || !x || !y) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (!p->trimmed) return fail(CHEBHIP_ERR_ARG, "cheb_apply_lap1d needs a plan from cheb_plan_create_trimmed");
  if (x == y) return fail(CHEBHIP_ERR_ARG, "x and y must be distinct");
  SweepParams sp = {};
  sp.ncols = p->ncols; sp.inner = p->inner;
  sp.in0 = x; sp.in_mode = IN_PLAIN;
  sp.alpha = alpha; sp.out = y;
  if (acc) { sp.out_mode = OUT_ACC; sp.acc = acc; } else sp.out_mode = OUT_STORE;
  HIPCHK(sweep_launch(p->lap, sp, (hipStream_t)stream));
  return 0;
}

// cheb_apply_lap1d along the OUTERMOST stored dimension (tr = 0, or tr = 1 of a batch) with the planes of x scattered over the arrays of
// `g` (sweep.h GatherSrc): y = alpha * (D D x), y dense.  *done = false: this plan / these arrays cannot take the gather launch.
namespace chebhip {
int lap1d_gather_try(cheb_plan *p, const GatherSrc &g, double alpha, double *y, hipStream_t st, bool *done) {
  *done = false;
  if (!p || !p->trimmed || !y) return 0;
  SweepParams sp = {};
  sp.ncols = p->ncols; sp.inner = p->inner;
  sp.in_mode = IN_PLAIN; sp.alpha = alpha; sp.out = y; sp.out_mode = OUT_STORE;
  HIPCHK(sweep_launch_gather(p->lap, sp, g, st, done));
  return 0;
}
}  // namespace chebhip

namespace chebhip {
int lap1d_multi_gather_try(int n, cheb_plan *const *plans, const double *x, double *const *outs, cheb_plan *gp, const GatherSrc &g, double *gout,
                           double alpha, hipStream_t st, bool *done) {
  *done = false;
  if (n < 1 || n > 8 || !gp || !gp->trimmed || !gout) return 0;
  const DiffMat *m[9]; SweepParams sp[9];
  for (int k = 0; k < n; k++) {
    if (!plans[k] || !plans[k]->trimmed) return 0;
    sp[k] = SweepParams{};
    sp[k].ncols = plans[k]->ncols; sp[k].inner = plans[k]->inner;
    sp[k].in0 = x; sp[k].in_mode = IN_PLAIN; sp[k].alpha = alpha; sp[k].out = outs[k]; sp[k].out_mode = OUT_STORE;
    m[k] = &plans[k]->lap;
  }
  sp[n] = SweepParams{};
  sp[n].ncols = gp->ncols; sp[n].inner = gp->inner; sp[n].in_mode = IN_PLAIN; sp[n].alpha = alpha; sp[n].out = gout; sp[n].out_mode = OUT_STORE;
  m[n] = &gp->lap;
  HIPCHK(sweep_launch_multi_gather_try(n + 1, m, sp, 1u << n, g, st, done));
  return 0;
}
}  // namespace chebhip

// n trimmed plans on the same tensor (different directions): outs[k] = alpha * (D_k D_k x) as ONE launch of n jobs where the 16-byte
// kernels allow it (*done), otherwise nothing is launched.  The local directions of a small slab (dist.hip).
namespace chebhip {
int lap1d_multi_try(int n, cheb_plan *const *plans, const double *x, double *const *outs, double alpha, hipStream_t st, bool *done) {
  *done = false;
  if (n < 2 || n > 9) return 0;
  const DiffMat *m[9]; SweepParams sp[9];
  for (int k = 0; k < n; k++) {
    if (!plans[k] || !plans[k]->trimmed) return 0;
    sp[k] = SweepParams{};
    sp[k].ncols = plans[k]->ncols; sp[k].inner = plans[k]->inner;
    sp[k].in0 = x; sp[k].in_mode = IN_PLAIN; sp[k].alpha = alpha; sp[k].out = outs[k]; sp[k].out_mode = OUT_STORE;
    m[k] = &plans[k]->lap;
  }
  HIPCHK(sweep_launch_multi_try(n, m, sp, st, done));
  return 0;
}
}  // namespace chebhip

extern "C" int cheb_apply(cheb_plan *p, const double *x, double *y, void *stream) {
  if (!p || !x || !y) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (x == y) return fail(CHEBHIP_ERR_ARG, "x and y must be distinct (as every ChebMult call site)");
  if (p->trimmed) return fail(CHEBHIP_ERR_ARG, "plan is trimmed: use cheb_apply_lap1d");
  StageTimer tm(CHEBHIP_STAGE_CHEB_APPLY, stream);
  SweepParams sp = {};
  sp.ncols = p->ncols; sp.inner = p->inner;
  sp.in0 = x; sp.out = y; sp.alpha = 1.0;
  sp.in_mode = IN_PLAIN; sp.out_mode = OUT_STORE;
  HIPCHK(sweep_launch(p->mat, sp, (hipStream_t)stream));
  return 0;
}

extern "C" int cheb_apply_host(cheb_plan *p, const double *x, double *y) {
  if (!p || !x || !y) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  const size_t bytes = (size_t)p->N * sizeof(double);
  if (!p->hx) { HIPCHK(hipMalloc((void **)&p->hx, bytes)); HIPCHK(hipMalloc((void **)&p->hy, bytes)); }
  HIPCHK(hipMemcpy(p->hx, x, bytes, hipMemcpyHostToDevice));
  int rc = cheb_apply(p, p->hx, p->hy, nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(y, p->hy, bytes, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int cheb_plan_destroy(cheb_plan *p) {
  if (!p) return 0;
  diffmat_destroy(&p->mat);
  diffmat_destroy(&p->lap);
  if (p->hx) (void)hipFree(p->hx);
  if (p->hy) (void)hipFree(p->hy);
  delete p;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// slab <-> exchange-buffer copies of the multi-GPU path (no counterpart in the serial reference)
// ---------------------------------------------------------------------------------------------
// A slab is (m0, M1, R) row-major.  The exchange buffer holds, for every peer s, the block
// slab[:, c1[s]:c1[s+1], :] contiguously (blocks in rank order): what all_to_all_single sends in the forward
// transpose and what it delivers in the backward one.
struct SlabSplit { int G; long c1[65]; };

__device__ __forceinline__ long slab_buf_index(const SlabSplit &sp, long m0, long M1, long R, long e) {
  const long i0 = e / (M1 * R), rem = e - i0 * (M1 * R);
  const long j = rem / R, r = rem - j * R;
  int s = 0;
  while (s + 1 < sp.G && j >= sp.c1[s + 1]) s++;
  const long w = sp.c1[s + 1] - sp.c1[s];
  return m0 * sp.c1[s] * R + (i0 * w + (j - sp.c1[s])) * R + r;
}

__global__ void k_slab_pack(SlabSplit sp, long m0, long M1, long R, const double *__restrict__ slab, double *__restrict__ buf) {
  const long n = m0 * M1 * R;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    buf[slab_buf_index(sp, m0, M1, R, e)] = slab[e];
}

// out = acc + alpha * buf (buf in exchange order)
__global__ void k_slab_unpack_add(SlabSplit sp, long m0, long M1, long R, const double *__restrict__ buf,
                                  const double *acc, double alpha, double *out) {
  const long n = m0 * M1 * R;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const double t = alpha * buf[slab_buf_index(sp, m0, M1, R, e)];
    out[e] = acc ? acc[e] + t : t;
  }
}

static int slab_split(int G, const long *c1, long M1, SlabSplit *sp) {
  if (G < 1 || G > 64 || !c1) return fail(CHEBHIP_ERR_ARG, "G = %d must be in 1..64", G);
  sp->G = G;
  for (int s = 0; s <= G; s++) {
    sp->c1[s] = c1[s];
    if (c1[s] < 0 || c1[s] > M1 || (s > 0 && c1[s] < c1[s - 1])) return fail(CHEBHIP_ERR_ARG, "column splits must be non-decreasing in 0..M1");
  }
  if (c1[0] != 0 || c1[G] != M1) return fail(CHEBHIP_ERR_ARG, "column splits must cover 0..M1");
  return 0;
}

extern "C" int cheb_slab_pack(long m0, long M1, long R, int G, const long *c1, const double *slab, double *buf, void *stream) {
  if (!slab || !buf || m0 < 0 || M1 < 0 || R < 1) return fail(CHEBHIP_ERR_ARG, "bad argument");
  SlabSplit sp; int rc = slab_split(G, c1, M1, &sp); if (rc) return rc;
  const long n = m0 * M1 * R;
  if (n == 0) return 0;
  long g = (n + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_slab_pack, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, sp, m0, M1, R, slab, buf);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int cheb_slab_unpack_add(long m0, long M1, long R, int G, const long *c1, const double *buf, const double *acc,
                                    double alpha, double *out, void *stream) {
  if (!buf || !out || m0 < 0 || M1 < 0 || R < 1) return fail(CHEBHIP_ERR_ARG, "bad argument");
  SlabSplit sp; int rc = slab_split(G, c1, M1, &sp); if (rc) return rc;
  const long n = m0 * M1 * R;
  if (n == 0) return 0;
  long g = (n + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_slab_unpack_add, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, sp, m0, M1, R, buf, acc, alpha, out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// pointwise kernels of the elliptic callbacks
// ---------------------------------------------------------------------------------------------
// w0 = VecScatter(GL)(U) then VecScatter(DL)(dirichlet): elliptic.C:486-493.
__global__ void k_gather_bc(long N, const int *__restrict__ ixL, const double *__restrict__ U,
                            const double *__restrict__ dirloc, double *__restrict__ w0) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const int g = ixL[i];
    w0[i] = g >= 0 ? U[g] : (dirloc ? dirloc[i] : 0.0);
  }
}

// w0 as above and eta = 1 + gamma u^e, deta = e gamma u^(e-1) (elliptic.C:508-509) in one pass over the local
// vector (FormFunction, elliptic.C:486-509).  For a small integer
// exponent (the reference's default is 2, elliptic.C:141) u^(e-1) is a product and u^e = u^(e-1) * u: no pow().
__global__ void k_gather_coeff(long N, const int *__restrict__ ixL, const double *__restrict__ U,
                               const double *__restrict__ dirloc, double gamma, double expo, int iexp,
                               double *__restrict__ w0, double *__restrict__ eta, double *__restrict__ deta) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const int g = ixL[i];
    const double v = g >= 0 ? U[g] : (dirloc ? dirloc[i] : 0.0);
    w0[i] = v;
    if (iexp > 0) {
      double pw = 1.0;
      for (int q = 1; q < iexp; q++) pw *= v;
      eta[i] = 1.0 + gamma * (pw * v);
      deta[i] = expo * gamma * pw;
    } else {
      eta[i] = 1.0 + gamma * pow(v, expo);
      deta[i] = expo * gamma * pow(v, expo - 1.0);
    }
  }
}

// The same two points per thread (N even): 16-byte stores of w0, eta, deta.  The pass writes twice what it reads; with
// 8-byte stores it ran at 4.7 TB/s.
__global__ void k_gather_coeff2(long N, const int *__restrict__ ixL, const double *__restrict__ U,
                                const double *__restrict__ dirloc, double gamma, double expo, int iexp,
                                double *__restrict__ w0, double *__restrict__ eta, double *__restrict__ deta) {
  const long half = N >> 1;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < half; t += (long)gridDim.x * blockDim.x) {
    const int2 g = ((const int2 *)ixL)[t];
    const long i = 2 * t;
    double v[2], e[2], de[2];
    v[0] = g.x >= 0 ? U[g.x] : (dirloc ? dirloc[i] : 0.0);
    v[1] = g.y >= 0 ? U[g.y] : (dirloc ? dirloc[i + 1] : 0.0);
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (iexp > 0) {
        double pw = 1.0;
        for (int r = 1; r < iexp; r++) pw *= v[q];
        e[q] = 1.0 + gamma * (pw * v[q]); de[q] = expo * gamma * pw;
      } else { e[q] = 1.0 + gamma * pow(v[q], expo); de[q] = expo * gamma * pow(v[q], expo - 1.0); }
    }
    ((double2 *)w0)[t] = make_double2(v[0], v[1]);
    ((double2 *)eta)[t] = make_double2(e[0], e[1]);
    ((double2 *)deta)[t] = make_double2(de[0], de[1]);
  }
}

// ec_k = {eta, c_k = deta * du0_k}: the two coefficients of the linearised flux eta g + c_k u (elliptic.C:321)
// side by side, so that the Jacobian apply fetches both with one 16-byte load; formed once per state
__global__ void k_cprod(long N, const double *__restrict__ eta, const double *__restrict__ deta, const double *__restrict__ du,
                        double2 *__restrict__ ec) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x)
    ec[i] = make_double2(eta[i], 0.5 * (deta[i] * du[i]));   // c / 2: the fused kernels hold 2 u (parity sums)
}

// f = eta * g (+ deta * u * du0): the pointwise flux of elliptic.C:319-323 / :511 as a pass of its own (slab mode,
// where the divergence sweep along dimension 0 takes a plain array); in place on g
__global__ void k_flux(long N, const double *__restrict__ eta, const double *__restrict__ deta, const double *__restrict__ u,
                       const double *__restrict__ du0, double *__restrict__ g) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    double f = eta[i] * g[i];
    if (deta) f = f + deta[i] * u[i] * du0[i];
    g[i] = f;
  }
}

// V = (a + b) (+ c): the terms of the one-launch constant-coefficient apply, added in the order of the accumulating chain
__global__ void k_sum_terms(long n, const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c, double *__restrict__ V) {
  const long half = n >> 1;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < half; t += (long)gridDim.x * blockDim.x) {
    const double2 x = ((const double2 *)a)[t], y = ((const double2 *)b)[t];
    double2 v = make_double2(x.x + y.x, x.y + y.y);
    if (c) { const double2 z = ((const double2 *)c)[t]; v.x = v.x + z.x; v.y = v.y + z.y; }
    ((double2 *)V)[t] = v;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { double v = a[n - 1] + b[n - 1]; if (c) v = v + c[n - 1]; V[n - 1] = v; }
}

// V = VecScatter(LG)(W): interior nodes of the local vector to the global one (elliptic.C:336)
__global__ void k_scatter_lg(long N, const int *__restrict__ ixL, const double *__restrict__ W, double *__restrict__ V) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const int g = ixL[i];
    if (g >= 0) V[g] = W[i];
  }
}

__global__ void k_fill(long N, double v, double *__restrict__ a) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) a[i] = v;
}

// rhs -= b : VecAXPY(rhs,-1,b) elliptic.C:530
__global__ void k_axpy(long N, double alpha, const double *__restrict__ x, double *__restrict__ y) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

// eta = 1 + gamma w0^2, deta = 2 gamma w0 (elliptic.C:508-509 with the default exponent 2) from the stored state w0
__global__ void k_coeff_sq(long N, const double *__restrict__ w0, double gamma, double *__restrict__ eta, double *__restrict__ deta) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const double v = w0[i];
    eta[i] = 1.0 + gamma * (v * v); deta[i] = 2.0 * gamma * v;
  }
}
// the coefficient pairs of the Jacobian apply straight from w0: {1 + gamma w0^2, (2 gamma w0) du0_k / 2}
__global__ void k_cprod_sq(long N, const double *__restrict__ w0, double gamma, const double *__restrict__ du, double2 *__restrict__ ec) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const double v = w0[i];
    ec[i] = make_double2(1.0 + gamma * (v * v), 0.5 * ((2.0 * gamma * v) * du[i]));
  }
}

static inline unsigned pw_grid(long N) { long g = (N + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

// ---------------------------------------------------------------------------------------------
// operator level: MatElliptic (elliptic.C:78-86)
// ---------------------------------------------------------------------------------------------
enum CoeffMode { COEFF_UNIT = 0, COEFF_FULL = 1 };

struct ell_op {
  int d = 0;
  std::vector<int> dims;
  long N = 0, G = 0;
  std::map<int, DiffMat> mats;          // one matrix per distinct extent
  std::map<int, DiffMat> laps;          // interior (D D) per distinct extent: the constant-coefficient path
  std::vector<unsigned> inner, ncols;   // per direction
  std::vector<unsigned> inner_g, ncols_g; // per direction, in the interior (global-vector) layout
  std::vector<int *> gcol;              // per direction: device [ncols_k]
  std::vector<long> gstride;            // per direction, in the global (interior) layout
  int *ixL = nullptr;                   // device [N]: global index or -1 (c->isL, elliptic.C:426)
  std::vector<double *> g;              // d work vectors: gradients / fluxes (c->w[1..d])
  double *W = nullptr;                  // accumulator (c->w[0] after VecZeroEntries)
  unsigned wpad = 0;                    // constant-coefficient path: row pitch of W in its padded interior layout (0: dense)
  double *Wj[2] = {nullptr, nullptr};   // constant-coefficient path, small grids: the other directions' terms of the one-launch apply (G doubles each)
  double *W2 = nullptr; size_t wsize = 0;   // ... large 3-D grids: the second direction's term in W's padded layout (two-launch apply, made on first use)
  double *w0 = nullptr;                 // local copy of the input (c->w[0] before), lazily allocated
  std::vector<double *> gradu;          // c->gradu[d], lazily allocated
  // slab mode (multi-GPU): the handle owns the planes [lo, lo + dims[0]) of a grid whose dimension 0 has
  // gP0 points; sweeps along dimension 0 go through `dim0` (transposes + pencil sweep in the driver)
  bool slab = false;
  int gP0 = 0, lo = 0;
  ell_dim0_fn dim0 = nullptr;
  void *dim0_ctx = nullptr;
  bool has_long = false;                // some extent > 256: every sweep goes through the unfused path (cheb_sweep_long_kernel)
  std::vector<double *> cprod;          // pairs {eta, deta * gradu[k] / 2} (2N doubles): what the Jacobian apply reads, refreshed when the state changes
  bool cdirty = true;
  double *eta = nullptr, *deta = nullptr, *dirloc = nullptr;
  bool dir_nonzero = false;             // some Dirichlet value != 0 (set_dirichlet): the interior-line FormFunction path does not apply
  // FormFunction on the interior line space (ell_fused4_function_trim) leaves w0 and gradu current but NOT eta / deta:
  // they are 1 + gamma w0^2 and 2 gamma w0 (exponent 2) and are formed when something asks for them (ell_sync_coeffs)
  bool coef_stale = false; double coef_gamma = 0.0;
  bool bdy_lines_dirty = false;         // w0 / gradu hold non-zero values on lines inside the boundary (left by the general path)
  // w0 and gradu[k] live SH elements into allocations of N + 16 doubles.  SH = 0 normally.  The interior-line FormFunction
  // stores them (8-byte pieces, lanes across the lines of the last dimension) starting at local column 1: with SH = 15
  // that column sits on a 128-byte boundary and a wave's row pieces are whole cache lines (256^3: 560 -> 505 us; at
  // SH = 0 the counters showed 12-16 % more bytes written than stored).  Every FormFunction rewrites these arrays, so the
  // shift is simply chosen per call (ell_state_layout); readers go through the pointers and are scalar kernels.
  double *w0_alloc = nullptr; std::vector<double *> gradu_alloc, cprod_alloc;
  int w0_shift = 0; std::vector<int> g_shift;
  CoeffMode mode = COEFF_UNIT;
  double *hU = nullptr, *hV = nullptr, *hB = nullptr;  // staging for host-pointer calls
};

static int ell_alloc_state(ell_op *op) {
  const size_t bytes = (size_t)op->N * sizeof(double), sbytes = bytes + 16 * sizeof(double);
  // The fills below run on the null stream, which does not order itself against a caller's non-blocking stream: the call
  // that allocates waits for them (once per handle).  Without this the clear of w0 / gradu could land AFTER the first
  // kernels of the caller's stream had written them (seen with thread ranks, each on a stream of its own).
  const bool fresh = !op->w0 || !op->eta || op->gradu.empty();
  if (!op->w0) { HIPCHK(hipMalloc((void **)&op->w0_alloc, sbytes)); HIPCHK(hipMemset(op->w0_alloc, 0, sbytes)); op->w0 = op->w0_alloc; op->w0_shift = 0; }   // boundary nodes read as zero until a pass writes them
  if (!op->eta) {
    HIPCHK(hipMalloc((void **)&op->eta, bytes));
    HIPCHK(hipMalloc((void **)&op->deta, bytes));
    hipLaunchKernelGGL(k_fill, dim3(pw_grid(op->N)), dim3(256), 0, nullptr, op->N, 1.0, op->eta);   // VecSet(eta,1) elliptic.C:265
    HIPCHK(hipMemset(op->deta, 0, bytes));                                                           // VecSet(deta,0) :266
  }
  if (op->gradu.empty()) {
    op->gradu.assign(op->d, nullptr); op->cprod.assign(op->d, nullptr); op->gradu_alloc.assign(op->d, nullptr); op->g_shift.assign(op->d, 0); op->cprod_alloc.assign(op->d, nullptr);
    for (int k = 0; k < op->d; k++) {
      HIPCHK(hipMalloc((void **)&op->gradu_alloc[k], sbytes));
      HIPCHK(hipMemset(op->gradu_alloc[k], 0, sbytes));
      op->gradu[k] = op->gradu_alloc[k];
      HIPCHK(hipMalloc((void **)&op->cprod_alloc[k], 2 * sbytes));
      op->cprod[k] = op->cprod_alloc[k] + 2 * (k < op->d - 1 ? 15 : 0);   // the pairs of local column 1 on a 128-byte boundary for the strided directions (see w0_alloc): Jacobian apply 502 -> 495 us
    }
    op->cdirty = true;
  }
  if (fresh) HIPCHK(hipStreamSynchronize(nullptr));
  return 0;
}

// eta / deta as the reference would hold them after the last FormFunction (elliptic.C:508-509), for whoever reads the arrays
static int ell_sync_coeffs(ell_op *op, hipStream_t st) {
  if (!op->coef_stale) return 0;
  hipLaunchKernelGGL(k_coeff_sq, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const double *)op->w0, op->coef_gamma, op->eta, op->deta);
  HIPCHK(hipGetLastError());
  op->coef_stale = false;
  return 0;
}
int ell_op_sync_coeffs(ell_op *op, void *stream) { return op ? ell_sync_coeffs(op, (hipStream_t)stream) : 0; }   // precond.hip, before it reads the view

int ell_op_fd_view(ell_op *op, chebhip::FdView *v) {
  if (!op || !v) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (op->slab) return fail(CHEBHIP_ERR_ARG, "slab-mode handle: the preconditioner comes from chebhip_dist_ell_pc");
  return ell_op_fd_view_any(op, v, nullptr);
}
int ell_op_fd_view_any(ell_op *op, chebhip::FdView *v, int *gP0) {
  if (!op || !v) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (gP0) *gP0 = op->gP0;
  if (op->d > 10) return fail(CHEBHIP_ERR_DIMS, "d > 10");
  int rc = ell_alloc_state(op); if (rc) return rc;
  v->d = op->d; v->dims = op->dims.data(); v->N = op->N; v->G = op->G; v->ixL = op->ixL;
  v->eta = op->eta; v->deta = op->deta;
  for (int k = 0; k < op->d; k++) v->gradu[k] = op->gradu[k];
  return 0;
}

static inline bool ell_is_bdy(const ell_op *op, const int *ind) {
  const int g0 = ind[0] + op->lo;
  if (g0 == 0 || g0 == op->gP0 - 1) return true;
  for (int j = 1; j < op->d; j++) if (ind[j] == 0 || ind[j] == op->dims[j] - 1) return true;
  return false;
}

static int ell_create(int d, const int *gdims, int lo, int hi, ell_dim0_fn dim0, void *dim0_ctx, ell_op **out) {
  if (!out) return fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  const bool slab = dim0 != nullptr;
  if (slab && (!gdims || d < 2)) return fail(CHEBHIP_ERR_DIMS, "slab mode needs d >= 2");
  std::vector<int> dimv;
  if (gdims && d >= 1 && d <= 10) {
    dimv.assign(gdims, gdims + d);
    if (slab) {
      if (!(0 <= lo && lo < hi && hi <= gdims[0])) return fail(CHEBHIP_ERR_ARG, "slab planes [%d, %d) outside 0..%d", lo, hi, gdims[0]);
      if (gdims[0] < 3) return fail(CHEBHIP_ERR_SIZE, "slab mode needs dims[0] >= 3");
      dimv[0] = hi - lo;
    } else lo = 0;
  }
  const int *dims = dimv.empty() ? gdims : dimv.data();
  if (!dims || d < 1 || d > 10) return fail(CHEBHIP_ERR_DIMS, "d = %d must be in 1..10 (elliptic.C:137)", d);
  long N = 1, G = 1;
  for (int k = 0; k < d; k++) {
    long nk; unsigned ik;
    int rc = check_geom(d, k, gdims, &nk, &ik, true);
    if (rc) return rc;
    N *= dims[k];
    if (!(slab && k == 0)) G *= (dims[k] > 2 ? dims[k] - 2 : 0);
  }
  long nint0 = 0;                                       // interior planes of dimension 0 owned by this handle
  if (slab) {
    for (int i = lo; i < hi; i++) if (i > 0 && i < gdims[0] - 1) nint0++;
    G *= nint0;
    if (N > 0x7fffffffL) return fail(CHEBHIP_ERR_DIMS, "tensor of more than 2^31-1 points");
  }
  int rc = require_device();
  if (rc) return rc;
  ell_op *op = new (std::nothrow) ell_op;
  if (!op) return fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  op->d = d; op->dims.assign(dims, dims + d); op->N = N; op->G = G;
  op->slab = slab; op->gP0 = gdims[0]; op->lo = lo; op->dim0 = dim0; op->dim0_ctx = dim0_ctx;
  // from here on failures go through ell_op_destroy
#define OPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ell_op_destroy(op); \
    return fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
  for (int k = 0; k < d; k++) {      // dimension 0: the global extent (in slab mode it is applied on pencils)
    const int Pk = (k == 0) ? gdims[0] : dims[k];
    if (!op->mats.count(Pk)) {
      DiffMat m; OPCHK(diffmat_create(Pk, &m)); op->mats[Pk] = m;
      if (m.KS == 0) op->has_long = true;
      else if (Pk > 2 && !slab) { DiffMat l; OPCHK(diffmat_create_lap(Pk, &l)); op->laps[Pk] = l; }
    }
  }
  // SetupBC (elliptic.C:372-434): ixL in BlockIt order; interior strides of the global vector
  std::vector<long> gs(d, 1);
  for (int k = d - 2; k >= 0; k--) gs[k] = gs[k + 1] * (dims[k + 1] - 2);    // dims[k+1], k+1 >= 1: never the split dimension
  {
    std::vector<int> ixL((size_t)N);
    std::vector<int> ind(d, 0);
    long g = 0;
    for (long l = 0; l < N; l++) {
      ixL[l] = ell_is_bdy(op, ind.data()) ? -1 : (int)g++;
      for (int j = d - 1; j >= 0; j--) { if (++ind[j] < dims[j]) break; ind[j] = 0; }
    }
    OPCHK(hipMalloc((void **)&op->ixL, (size_t)N * sizeof(int)));
    OPCHK(hipMemcpy(op->ixL, ixL.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice));
  }
  op->inner.resize(d); op->ncols.resize(d); op->gcol.assign(d, nullptr); op->gstride = gs;
  op->inner_g.resize(d); op->ncols_g.resize(d);
  for (int k = 0; k < d; k++) {
    op->inner_g[k] = (unsigned)gs[k];
    op->ncols_g[k] = (dims[k] > 2 && !slab) ? (unsigned)(G / (dims[k] - 2)) : 0u;
  }
  for (int k = 0; k < d; k++) {
    unsigned in = 1; for (int r = k + 1; r < d; r++) in *= dims[r];
    op->inner[k] = in; op->ncols[k] = (unsigned)(N / dims[k]);
    // line c of direction k <-> multi-index over the other dims (row-major, dim k removed)
    std::vector<int> tab(op->ncols[k]);
    std::vector<int> ind(d, 0);
    for (unsigned c = 0; c < op->ncols[k]; c++) {
      bool interior = true; long gi = 0;
      const int off0 = (lo == 0) ? 1 : 0;               // local plane of the first interior plane of this handle
      for (int r = 0; r < d; r++) {
        if (r == k) continue;
        const int gr = ind[r] + (r == 0 ? lo : 0), Pr = (r == 0) ? gdims[0] : dims[r];
        if (gr == 0 || gr == Pr - 1) interior = false;
        gi += (long)(ind[r] - (r == 0 ? off0 : 1)) * gs[r];
      }
      tab[c] = (interior && dims[k] > 2 && !(slab && k == 0)) ? (int)gi : -1;
      for (int r = d - 1; r >= 0; r--) { if (r == k) continue; if (++ind[r] < dims[r]) break; ind[r] = 0; }
    }
    OPCHK(hipMalloc((void **)&op->gcol[k], tab.size() * sizeof(int)));
    OPCHK(hipMemcpy(op->gcol[k], tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  op->g.assign(d, nullptr);
  for (int k = 0; k < d; k++) OPCHK(hipMalloc((void **)&op->g[k], (size_t)N * sizeof(double)));
  size_t wsize = (size_t)N;
  if (!slab && !op->has_long && (d == 2 || d == 3)) {
    // padded accumulator of the constant-coefficient path (ell_op_mult): needs the 16-byte kernel's v3
    // (interior lines of more than 64 points) and even interior extents
    bool ok = true;
    for (int k = 0; k < d; k++) ok = ok && dims[k] - 2 > 64 && ((dims[k] - 2) & 1) == 0 && dims[k] <= 256;
    if (ok) {
      op->wpad = (unsigned)((dims[d - 1] - 2 + 15) / 16 * 16);
      size_t need = op->wpad;
      for (int k = 0; k < d - 1; k++) need *= (size_t)(dims[k] - 2);
      if (need > wsize) wsize = need;
    }
  }
  OPCHK(hipMalloc((void **)&op->W, wsize * sizeof(double)));
  op->wsize = wsize;
#undef OPCHK
  *out = op;
  return 0;
}

extern "C" int ell_op_create(int d, const int *dims, ell_op **out) { return ell_create(d, dims, 0, 0, nullptr, nullptr, out); }

// Slab of the planes [lo, hi) of grid dimension 0 (multi-GPU): vectors are the serial ones restricted to
// the slab (contiguous pieces, dimension 0 being outermost); sweeps along dimension 0 are delegated to `dim0`.
extern "C" int ell_op_create_slab(int d, const int *dims, int lo, int hi, ell_dim0_fn dim0, void *dim0_ctx, ell_op **out) {
  if (!dim0) return fail(CHEBHIP_ERR_ARG, "slab mode needs the dimension-0 callback");
  return ell_create(d, dims, lo, hi, dim0, dim0_ctx, out);
}

// Pencil side of the slab mode: out = D_0 in on an array (dims[0], ncol), lines along dimension 0 with stride ncol
extern "C" int ell_op_pencil_sweep(ell_op *op, long ncol, const double *in, double *out, void *stream) {
  if (!op || !in || !out || ncol < 0) return fail(CHEBHIP_ERR_ARG, "bad argument");
  if (ncol == 0) return 0;
  SweepParams sp = {};
  sp.ncols = (unsigned)ncol; sp.inner = (unsigned)ncol;
  sp.in0 = in; sp.in_mode = IN_PLAIN; sp.out = out; sp.out_mode = OUT_STORE; sp.alpha = 1.0;
  HIPCHK(sweep_launch(op->mats[op->gP0], sp, (hipStream_t)stream));
  return 0;
}

// The same with the pencil's planes read in place from the ranks' slab fields (sweep.h GatherSrc; slabx.hip, direct transports)
namespace chebhip {
bool ell_pencil_gather_supported(const ell_op *op) {
  if (!op || opt(OPT_GENERAL_KERNELS) || opt(OPT_SEPARATE_LAUNCHES)) return false;
  auto a = op->mats.find(op->gP0);
  return a != op->mats.end() && a->second.KS >= 16;
}
int ell_pencil_gather_try(ell_op *op, long ncol, const GatherSrc &g, double *out, hipStream_t st, bool *done) {
  *done = false;
  if (!op || !out || ncol <= 0) return 0;
  SweepParams sp = {};
  sp.ncols = (unsigned)ncol; sp.inner = (unsigned)ncol;
  sp.in_mode = IN_PLAIN; sp.out = out; sp.out_mode = OUT_STORE; sp.alpha = 1.0;
  HIPCHK(sweep_launch_gather(op->mats[op->gP0], sp, g, st, done));
  return 0;
}
}  // namespace chebhip

extern "C" int ell_op_destroy(ell_op *op) {
  if (!op) return 0;
  for (auto &kv : op->mats) diffmat_destroy(&kv.second);
  for (auto &kv : op->laps) diffmat_destroy(&kv.second);
  for (auto p : op->gcol) if (p) (void)hipFree(p);
  for (auto p : op->g) if (p) (void)hipFree(p);
  for (auto p : op->gradu_alloc) if (p) (void)hipFree(p);
  for (auto p : op->cprod_alloc) if (p) (void)hipFree(p);
  double *singles[] = {op->W, op->W2, op->Wj[0], op->Wj[1], op->w0_alloc, op->eta, op->deta, op->dirloc, op->hU, op->hV, op->hB};
  for (double *p : singles) if (p) (void)hipFree(p);
  if (op->ixL) (void)hipFree(op->ixL);
  delete op;
  return 0;
}

extern "C" long ell_op_local_size(const ell_op *op) { return op ? op->N : -1; }
extern "C" long ell_op_global_size(const ell_op *op) { return op ? op->G : -1; }
extern "C" long ell_op_dirichlet_size(const ell_op *op) { return op ? op->N - op->G : -1; }

// the 16-byte kernels with per-array geometry need 16-B aligned MatShell vectors (hipMalloc and PETSc give them); a vector that
// is not goes through the general kernels
static inline bool aligned16(const void *q) { return ((size_t)q & 15) == 0; }

// ---- the straight-line fused kernel (fused4.hip): d = 2, 3, every extent even and 66..256, no slab ----
static bool ell_fused4_ok(ell_op *op) {
  if (op->slab || op->has_long || (op->d != 2 && op->d != 3) || op->G == 0) return false;
  for (int k = 0; k < op->d; k++) if (!fused4_eligible(op->mats[op->dims[k]])) return false;
  return (size_t)op->N * 16 < 0x38000000ull;
}

// Jacobian apply, direction k: V (+)= -D_k( eta D_k U + c_k U ) on the interior lines.  U and V in the interior layout,
// the coefficient pairs in the local one (entered at the first interior line), W in its padded interior layout.
static int ell_fused4_jacobian(ell_op *op, int k, const double *U, double *V, hipStream_t st) {
  const int d = op->d;
  const unsigned n1 = d == 3 ? op->dims[1] : 0, nlast = op->dims[d - 1];
  const unsigned nl = nlast - 2, nm = d == 3 ? n1 - 2 : 1u, n0i = op->dims[0] - 2, wp = op->wpad;
  const unsigned s0 = d == 3 ? n1 * nlast : nlast;                       // local stride of dimension 0
  Fused4Params q = {};
  q.alpha = -1.0;                                                        // VecAXPY(w0,-1,.) elliptic.C:333
  q.in = U; q.in_bytes = (unsigned)((size_t)op->G * 8);
  size_t shift;                                                          // first interior line in the local layout
  const size_t wbytes = (size_t)(d == 3 ? n0i * nm : n0i) * wp * 8;
  const bool last = k == d - 1;
  if (!last) {
    const bool first = d == 3 && k == 0;
    q.qmax = nl;
    if (d == 2) { q.nouter = 1; q.gi = {0, 1, nl}; q.gc = {0, 1, nlast}; q.go = {0, 1, wp}; shift = 1; }
    else if (first) { q.nouter = nm; q.gi = {nl, 1, nm * nl}; q.gc = {nlast, 1, s0}; q.go = {wp, 1, nm * wp}; shift = nlast + 1; }
    else { q.nouter = n0i; q.gi = {nm * nl, 1, nl}; q.gc = {s0, 1, nlast}; q.go = {nm * wp, 1, wp}; shift = s0 + 1; }
    q.ga = q.go; q.out = op->W; q.out_bytes = (unsigned)wbytes;
    if (k > 0) { q.acc = op->W; q.acc_bytes = (unsigned)wbytes; }
  } else {
    q.nouter = d == 3 ? n0i : 1u; q.qmax = d == 3 ? nm : n0i;
    q.gi = {nm * nl, nl, 1}; q.gc = {s0, nlast, 1}; q.ga = {nm * wp, wp, 1}; q.go = q.gi;
    shift = d == 3 ? s0 + nlast : nlast;
    q.acc = op->W; q.acc_bytes = (unsigned)wbytes; q.out = V; q.out_bytes = (unsigned)((size_t)op->G * 8);
  }
  q.coef = (const void *)(op->cprod[k] + 2 * shift); q.coef_bytes = (unsigned)(((size_t)op->N - shift) * 16);
  HIPCHK(fused4_launch(op->mats[op->dims[k]], q, last, true, k > 0, false, st));
  return 0;
}

// FormFunction, direction k (elliptic.C:497-528): gradu[k] = D_k w0 is stored on the way, W (+)= -D_k( eta gradu[k] ) in
// the local layout; the last direction writes rhs = scatter(W) - b in the interior layout.
static int ell_fused4_function(ell_op *op, int k, double gamma, double exponent, const double *b, double *rhs, hipStream_t st) {
  const int d = op->d;
  const unsigned n0 = op->dims[0], n1 = d == 3 ? op->dims[1] : 0, nlast = op->dims[d - 1];
  const unsigned nl = nlast - 2, nm = d == 3 ? n1 - 2 : 1u, s0 = d == 3 ? n1 * nlast : nlast;
  const unsigned nbytes = (unsigned)((size_t)op->N * 8), gbytes = (unsigned)((size_t)op->G * 8);
  Fused4Params q = {};
  q.alpha = -1.0;
  q.in = op->w0; q.in_bytes = nbytes; q.coef = op->eta; q.coef_bytes = nbytes; q.gout = op->gradu[k]; q.gout_bytes = nbytes;
  // the reference's default exponent (elliptic.C:141): eta = 1 + gamma w0^2 is formed from the line in LDS instead of being read
  if (exponent == 2.0 && !opt(OPT_ETA_FROM_MEMORY)) { q.eta_square = 1; q.gamma4 = 0.25 * gamma; }
  const bool last = k == d - 1;
  if (!last) {
    if (k == 0) { q.nouter = 1; q.qmax = s0; q.gi = {0, 1, s0}; }         // lines along dimension 0: every (i1, i2)
    else { q.nouter = n0; q.qmax = nlast; q.gi = {s0, 1, nlast}; }
    q.gc = q.ga = q.go = q.gi;
    q.out = op->W; q.out_bytes = nbytes;
    if (k > 0) { q.acc = op->W; q.acc_bytes = nbytes; }
  } else {
    q.nouter = d == 3 ? n0 : 1u; q.qmax = d == 3 ? n1 : n0;
    q.gi = {s0, nlast, 1}; q.gc = q.ga = q.gi;
    q.go = {nm * nl, nl, 1};
    q.acc = op->W; q.acc_bytes = nbytes; q.out = rhs; q.out_bytes = gbytes;
    q.sub = b; q.sub_bytes = b ? gbytes : 0u;
  }
  HIPCHK(fused4_launch(op->mats[op->dims[k]], q, last, false, k > 0, last, st));
  return 0;
}

// Where w0 and gradu[k] sit inside their allocations for the FormFunction about to run (see ell_op::w0_alloc).  An array whose
// shift changes is cleared: the interior-line path relies on zeros on the lines inside the boundary, the general path
// rewrites every entry anyway.
static int ell_state_layout(ell_op *op, bool trim, hipStream_t st) {
  const size_t sbytes = ((size_t)op->N + 16) * sizeof(double);
  const int d = op->d;
  bool cleared_all = true;
  auto place = [&](double *alloc, double **ptr, int *cur, int want) -> hipError_t {
    if (*cur == want) { cleared_all = false; return hipSuccess; }
    hipError_t e = hipMemsetAsync(alloc, 0, sbytes, st);
    *cur = want; *ptr = alloc + want;
    return e;
  };
  HIPCHK(place(op->w0_alloc, &op->w0, &op->w0_shift, trim ? 15 : 0));
  for (int k = 0; k < d; k++) HIPCHK(place(op->gradu_alloc[k], &op->gradu[k], &op->g_shift[k], (trim && k < d - 1) ? 15 : 0));
  if (trim && op->bdy_lines_dirty && !cleared_all) {          // same layout as last time, but the general path left boundary-line values
    HIPCHK(hipMemsetAsync(op->w0_alloc, 0, sbytes, st));
    for (int k = 0; k < d; k++) HIPCHK(hipMemsetAsync(op->gradu_alloc[k], 0, sbytes, st));
  }
  if (trim) op->bdy_lines_dirty = false;
  return 0;
}

// FormFunction, direction k, on the INTERIOR line space (homogeneous Dirichlet rows, exponent 2): the launch reads U in
// the interior layout of the MatShell vector as the Jacobian mode does (end points of a line are implicit zeros, lines
// inside the boundary are identically zero and are not visited), forms eta = 1 + gamma u^2 on chip, stores gradu[k] (and,
// in the first direction, w0) through a base shifted to the first interior line of the local layout, keeps W in its padded
// interior layout and, in the last direction, writes rhs = W - D_k f - b.  No gather pass: 104 B/point per residual
// against 132 with it (SURVEY 8d model: 160).
static int ell_fused4_function_trim(ell_op *op, int k, double gamma, const double *U, const double *b, double *rhs, hipStream_t st) {
  const int d = op->d;
  const unsigned n1 = d == 3 ? op->dims[1] : 0, nlast = op->dims[d - 1];
  const unsigned nl = nlast - 2, nm = d == 3 ? n1 - 2 : 1u, n0i = op->dims[0] - 2, wp = op->wpad;
  const unsigned s0 = d == 3 ? n1 * nlast : nlast;                       // local stride of dimension 0
  Fused4Params q = {};
  q.alpha = -1.0; q.trimf = 1; q.eta_square = 1; q.gamma4 = 0.25 * gamma;
  q.in = U; q.in_bytes = (unsigned)((size_t)op->G * 8);
  size_t shift;                                                          // first interior line in the local layout
  const size_t wbytes = (size_t)(d == 3 ? n0i * nm : n0i) * wp * 8;
  const bool last = k == d - 1;
  if (!last) {
    const bool first = d == 3 && k == 0;
    q.qmax = nl;
    if (d == 2) { q.nouter = 1; q.gi = {0, 1, nl}; q.gc = {0, 1, nlast}; q.go = {0, 1, wp}; shift = 1; }
    else if (first) { q.nouter = nm; q.gi = {nl, 1, nm * nl}; q.gc = {nlast, 1, s0}; q.go = {wp, 1, nm * wp}; shift = nlast + 1; }
    else { q.nouter = n0i; q.gi = {nm * nl, 1, nl}; q.gc = {s0, 1, nlast}; q.go = {nm * wp, 1, wp}; shift = s0 + 1; }
    q.ga = q.go; q.out = op->W; q.out_bytes = (unsigned)wbytes;
    if (k > 0) { q.acc = op->W; q.acc_bytes = (unsigned)wbytes; }
  } else {
    q.nouter = d == 3 ? n0i : 1u; q.qmax = d == 3 ? nm : n0i;
    q.gi = {nm * nl, nl, 1}; q.gc = {s0, nlast, 1}; q.ga = {nm * wp, wp, 1}; q.go = q.gi;
    shift = d == 3 ? s0 + nlast : nlast;
    q.acc = op->W; q.acc_bytes = (unsigned)wbytes; q.out = rhs; q.out_bytes = (unsigned)((size_t)op->G * 8);
    q.sub = b; q.sub_bytes = b ? (unsigned)((size_t)op->G * 8) : 0u;
  }
  q.gout = op->gradu[k] + shift; q.gout_bytes = (unsigned)(((size_t)op->N - shift) * 8);
  if (k == 0) { q.w0out = op->w0 + shift; q.w0_bytes = q.gout_bytes; }
  HIPCHK(fused4_launch(op->mats[op->dims[k]], q, last, false, k > 0, false, st));
  return 0;
}

// Where the k-th term -D_k f_k of the divergence goes: W = -t0; W -= t_k; out_global = scatter(W - t_{d-1}).
static void ell_out_chain(ell_op *op, int k, double *out_global, SweepParams *sp) {
  const int d = op->d;
  sp->alpha = -1.0;                                              // VecAXPY(w0,-1,.) elliptic.C:333
  sp->gcol = op->gcol[k]; sp->gstride = op->gstride[k];
  if (k == d - 1) { sp->out_mode = OUT_ACC_SCATTER; sp->out = out_global; sp->acc = (d > 1) ? op->W : nullptr; }  // + VecScatter LG (:336)
  else if (k == 0) { sp->out_mode = OUT_STORE; sp->out = op->W; }
  else { sp->out_mode = OUT_ACC; sp->out = op->W; sp->acc = op->W; }
}

// One sweep D_k of the slab mode: along dimension 0 through the driver, otherwise a local launch
static int ell_slab_sweep(ell_op *op, int k, const double *x, double *y, hipStream_t st) {
  if (k == 0) return op->dim0(op->dim0_ctx, 0, 1, x, nullptr, 1.0, y, st);
  SweepParams sp = {};
  sp.ncols = op->ncols[k]; sp.inner = op->inner[k];
  sp.in0 = x; sp.in_mode = IN_PLAIN; sp.out = y; sp.out_mode = OUT_STORE; sp.alpha = 1.0;
  HIPCHK(sweep_launch(op->mats[op->dims[k]], sp, st));
  return 0;
}

// out_global = scatter( -sum_k D_k f_k ) in slab mode; f_k = flux(src[k]) is formed by a pointwise pass for k = 0
// (the sweep along dimension 0 takes a plain array) and on load for the others.  mode: IN_PLAIN / IN_FLUX_ETA / IN_FLUX_FULL
static int ell_slab_divergence(ell_op *op, int in_mode, double *const *src, double *out_global, hipStream_t st) {
  const int d = op->d;
  if (in_mode != IN_PLAIN)
    hipLaunchKernelGGL(k_flux, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const double *)op->eta,
                       (const double *)(in_mode == IN_FLUX_FULL ? op->deta : nullptr), (const double *)op->w0,
                       (const double *)(in_mode == IN_FLUX_FULL ? op->gradu[0] : nullptr), src[0]);
  int rc = op->dim0(op->dim0_ctx, 0, 1, src[0], nullptr, -1.0, op->W, st);          // W = -D_0 f_0
  if (rc) return rc;
  for (int k = 1; k < d; k++) {
    SweepParams sp = {};
    sp.ncols = op->ncols[k]; sp.inner = op->inner[k];
    sp.in0 = src[k]; sp.in1 = op->eta; sp.in2 = op->deta; sp.in3 = op->w0;
    sp.in4 = op->gradu.empty() ? nullptr : op->gradu[k];
    sp.in_mode = in_mode; sp.alpha = -1.0;
    if (k == d - 1) { sp.out_mode = OUT_ACC_SCATTER; sp.out = out_global; sp.acc = op->W; sp.gcol = op->gcol[k]; sp.gstride = op->gstride[k]; }
    else { sp.out_mode = OUT_ACC; sp.out = op->W; sp.acc = op->W; }
    HIPCHK(sweep_launch(op->mats[op->dims[k]], sp, st));
  }
  return 0;
}

// Long lines (some extent > 256): every sweep is a PLAIN one, so that it can take the library-GEMM route of
// sweep_launch; gather, flux and scatter are pointwise passes -- literally the reference's structure.
static int ell_plain_sweep(ell_op *op, int k, const double *x, double *y, int out_mode, const double *acc, double alpha, hipStream_t st) {
  SweepParams sp = {};
  sp.ncols = op->ncols[k]; sp.inner = op->inner[k];
  sp.in0 = x; sp.in_mode = IN_PLAIN; sp.out = y; sp.out_mode = out_mode; sp.acc = acc; sp.alpha = alpha;
  HIPCHK(sweep_launch(op->mats[op->dims[k]], sp, st));
  return 0;
}

// V = scatter(-sum_k D_k flux_k(src[k])); flux: 0 none, 1 eta * g, 2 eta * g + deta * w0 * gradu[k]; src is overwritten
static int ell_plain_divergence(ell_op *op, int flux, double *const *src, double *V, hipStream_t st) {
  for (int k = 0; k < op->d; k++) {
    if (flux)
      hipLaunchKernelGGL(k_flux, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const double *)op->eta,
                         (const double *)(flux == 2 ? op->deta : nullptr), (const double *)op->w0,
                         (const double *)(flux == 2 ? op->gradu[k] : nullptr), src[k]);
    int rc = ell_plain_sweep(op, k, src[k], op->W, k == 0 ? OUT_STORE : OUT_ACC, op->W, -1.0, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_scatter_lg, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const int *)op->ixL, (const double *)op->W, V);
  HIPCHK(hipGetLastError());
  return 0;
}

static int ell_mult_plain(ell_op *op, const double *U, double *V, hipStream_t st) {
  int rc = ell_alloc_state(op); if (rc) return rc;
  hipLaunchKernelGGL(k_gather_bc, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, op->ixL, U, (const double *)nullptr, op->w0);
  for (int k = 0; k < op->d; k++) if ((rc = ell_plain_sweep(op, k, op->w0, op->g[k], OUT_STORE, nullptr, 1.0, st))) return rc;
  return ell_plain_divergence(op, op->mode == COEFF_UNIT ? 0 : 2, op->g.data(), V, st);
}

static int ell_mult_slab(ell_op *op, const double *U, double *V, hipStream_t st) {
  int rc = ell_alloc_state(op); if (rc) return rc;                    // w0
  hipLaunchKernelGGL(k_gather_bc, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, op->ixL, U, (const double *)nullptr, op->w0);
  for (int k = 0; k < op->d; k++) if ((rc = ell_slab_sweep(op, k, op->w0, op->g[k], st))) return rc;
  // a rank without interior nodes still takes part in the exchanges of the others
  // (its boundary plane carries the flux f_0 = eta g_0 of the lines that cross it: w0 is zero there, elliptic.C:305-308)
  if (op->G == 0) {
    if (op->mode != COEFF_UNIT)
      hipLaunchKernelGGL(k_flux, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const double *)op->eta, (const double *)op->deta,
                         (const double *)op->w0, (const double *)op->gradu[0], op->g[0]);
    return op->dim0(op->dim0_ctx, 0, 1, op->g[0], nullptr, -1.0, op->W, st);
  }
  return ell_slab_divergence(op, op->mode == COEFF_UNIT ? IN_PLAIN : IN_FLUX_FULL, op->g.data(), V, st);
}

// Largest padded field (doubles) for which the large-grid constant-coefficient apply runs as two launches (ell_op_mult): measured on
// MI355X (profiles/r06_two_launch_ab.txt) -- a win while the four fields of the apply stay in the 256-MB Infinity Cache, a loss at 256^3.
constexpr size_t TWO_LAUNCH_MAX = 9000000;

extern "C" int ell_op_mult(ell_op *op, const double *U, double *V, void *stream) {
  // empty vectors (a slab that owns only boundary planes) may be NULL
  if (!op || ((!U || !V) && !(op->slab && op->G == 0))) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (U && U == V) return fail(CHEBHIP_ERR_ARG, "U and V must be distinct (MatMult never aliases its vectors)");
  StageTimer tm(CHEBHIP_STAGE_ELL_MULT, stream);
  if (op->slab) return ell_mult_slab(op, U, V, (hipStream_t)stream);
  if (op->G == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (op->has_long) return ell_mult_plain(op, U, V, st);
  if (op->mode == COEFF_UNIT) {
    // Linear state (eta == 1, deta == 0; homogeneous Dirichlet rows, elliptic.C:305-308): every
    // array of the apply lives in the interior layout of the global vectors, so there is no
    // gather/scatter at all -- lines through boundary nodes carry zeros in and are not read out.
    // V = -sum_k D_k D_k w0 restricted to the interior, one launch per direction.
    // The two sweeps of a direction collapse into one product with the interior block of D D (constant coefficient):
    // half the MFMA work of the fused gradient -> flux -> divergence launch the variable-coefficient path needs.
    // Small grids (2 <= d <= 3, fewer than 6 M unknowns; option "poisson_launches"): the d directions as ONE launch of d
    // jobs, each storing its term, and a pointwise sum in the order of the chain below (the same bits).  80 B/point instead
    // of 64, but at these sizes the vectors sit in the Infinity Cache and the apply is bound by its launches, not its bytes:
    // A/B on one handle (tools/poisson_ab.py): 32^3 14.2 -> 11.6 us, 64^3 15.8 -> 11.4, 96^3 33.1 -> 24.9, 128^3 37.8 -> 36.6,
    // 160^3 124 -> 106; 192^3 151 -> 160 and 256^3 246 -> 327 are why there is a size limit.  Falls through when the jobs
    // cannot share a launch (mixed line lengths, odd strides, unaligned vectors).
    {
      const int mode = opt(OPT_POISSON_LAUNCHES);
      const int d = op->d;
      if (d >= 2 && d <= 3 && mode != 2 && (mode == 1 || op->G < 6000000L) && aligned16(U) && aligned16(V)) {
        if (!op->Wj[0]) {
          HIPCHK(hipMalloc((void **)&op->Wj[0], (size_t)(op->G + 2) * sizeof(double)));
          HIPCHK(hipMalloc((void **)&op->Wj[1], (size_t)(op->G + 2) * sizeof(double)));
        }
        const DiffMat *m[3]; SweepParams sp[3];
        double *term[3] = {op->W, op->Wj[0], op->Wj[1]};
        // d = 3, lines of at most 128 points (round 4): the first two directions as one launch of two jobs, and the LAST direction takes
        // both terms as it stores, V = (t_0 + t_1) - L_2 U (OUT_ACC2: the chain's order, the same bits) -- two launches and 64 B/point
        // instead of a three-job launch, 80 B/point and a sum pass.  poisson_launches = 1 keeps the three-job route (A/B, one handle,
        // tools/poisson_ab.py): 128^3 36.5 -> 32.8 us; 96^3 25.4 -> 29.3 and 64^3 10.7 -> 13.3 the other way (there the second launch
        // costs more than the bytes it saves), hence the lower size limit.
        if (d == 3 && mode == 0 && op->G >= 1500000L) {
          SweepParams last = {};
          last.ncols = op->ncols_g[2]; last.inner = op->inner_g[2];
          last.in0 = U; last.in_mode = IN_PLAIN; last.alpha = -1.0; last.out_mode = OUT_ACC2; last.acc = term[0]; last.acc2 = term[1]; last.out = V;
          if (!opt(OPT_GENERAL_KERNELS) && sweep_vec_eligible(op->laps[op->dims[2]], last)) {
            for (int k = 0; k < 2; k++) {
              sp[k] = SweepParams{};
              sp[k].ncols = op->ncols_g[k]; sp[k].inner = op->inner_g[k];
              sp[k].in0 = U; sp[k].in_mode = IN_PLAIN; sp[k].alpha = -1.0; sp[k].out_mode = OUT_STORE; sp[k].out = term[k];
              m[k] = &op->laps[op->dims[k]];
            }
            bool done2 = false;
            HIPCHK(sweep_launch_multi_try(2, m, sp, st, &done2));
            if (done2) { HIPCHK(sweep_launch(op->laps[op->dims[2]], last, st)); return 0; }
          }
        }
        for (int k = 0; k < d; k++) {
          sp[k] = SweepParams{};
          sp[k].ncols = op->ncols_g[k]; sp[k].inner = op->inner_g[k];
          sp[k].in0 = U; sp[k].in_mode = IN_PLAIN; sp[k].alpha = -1.0; sp[k].out_mode = OUT_STORE; sp[k].out = term[k];
          m[k] = &op->laps[op->dims[k]];
        }
        bool done = false;
        HIPCHK(sweep_launch_multi_try(d, m, sp, st, &done));
        if (done) {
          hipLaunchKernelGGL(k_sum_terms, dim3(pw_grid((op->G + 1) >> 1)), dim3(256), 0, st, op->G, (const double *)term[0], (const double *)term[1],
                             (const double *)(d == 3 ? term[2] : nullptr), V);
          HIPCHK(hipGetLastError());
          return 0;
        }
      }
    }
    if (op->wpad && aligned16(U) && aligned16(V)) {
      // d = 2, 3 with lines of more than 64 points: the accumulator W keeps its rows padded to a multiple of
      // 128 B (pitch wpad), so that the strided launches read and write whole cache lines of it; U and V stay
      // dense.  Tiles of the strided directions are (outer index, 32 neighbouring points of the last dimension).
      const int d = op->d;
      const unsigned nl = (unsigned)op->dims[d - 1] - 2, nm = d == 3 ? (unsigned)op->dims[1] - 2 : 1u, wp = op->wpad;
      // direction k: input U (dense), result / operand in W's padded layout; the caller sets the output side
      auto dir_params = [&](int k) {
        SweepParams sp = {};
        sp.in0 = U; sp.in_mode = IN_PLAIN; sp.alpha = -1.0;
        if (k == d - 1) {                                          // contiguous lines
          sp.ncols = op->ncols_g[k]; sp.inner = 1;
          sp.in_os = nl; sp.acc_os = wp; sp.out_os = nl;
        } else {
          const bool first = (k == 0 && d == 3);                   // lines along dimension 0 of a 3-D grid: outer index = dimension 1
          sp.ncols = op->ncols_g[k]; sp.qmax = nl;
          sp.nouter = d == 2 ? 1u : (first ? nm : (unsigned)op->dims[0] - 2);
          sp.in_rs = d == 2 ? nl : (first ? nm * nl : nl);  sp.in_os = d == 2 ? nl : (first ? nl : nm * nl);
          const unsigned w_rs = d == 2 ? wp : (first ? nm * wp : wp), w_os = d == 2 ? wp : (first ? wp : nm * wp);
          sp.acc_rs = sp.out_rs = w_rs; sp.acc_os = sp.out_os = w_os;
          sp.inner = sp.in_rs;
        }
        return sp;
      };
      // d = 3 (round 6): the first two directions as ONE launch of two jobs, each storing its term in the padded layout (W, W2), and
      // the last direction adds both as it stores, V = (t_0 + t_1) - L_2 U (OUT_ACC2) -- the chain's order, the same bits, the same
      // 64 B/point, but two launches instead of three: one matrix fetch + pipeline fill and one launch boundary less (~13 us of a
      // 245-us matvec at 256^3), and every workgroup of the first launch walks 16 tiles per fill instead of 8.  At KS = 32 the
      // kernel keeps ONE set of the two operands (sweep_vec.hip, ONEBUF).  Option poisson_launches = 2: a launch per direction (A/B).
      const int pl = opt(OPT_POISSON_LAUNCHES);
      if (d == 3 && pl != 2 && (pl == 3 || op->wsize <= (size_t)TWO_LAUNCH_MAX) && !opt(OPT_SEPARATE_LAUNCHES) && !opt(OPT_GENERAL_KERNELS)) {
        if (!op->W2) HIPCHK(hipMalloc((void **)&op->W2, op->wsize * sizeof(double)));
        SweepParams last = dir_params(2);
        last.out_mode = OUT_ACC2; last.acc = op->W; last.acc2 = op->W2; last.out = V;
        if (sweep_vec_eligible(op->laps[op->dims[2]], last)) {
          const DiffMat *m[2]; SweepParams sp[2];
          for (int k = 0; k < 2; k++) {
            sp[k] = dir_params(k);
            sp[k].out_mode = OUT_STORE; sp[k].out = k == 0 ? op->W : op->W2;
            m[k] = &op->laps[op->dims[k]];
          }
          bool done2 = false;
          HIPCHK(sweep_launch_multi_try(2, m, sp, st, &done2));
          if (done2) { HIPCHK(sweep_launch(op->laps[op->dims[2]], last, st)); return 0; }
        }
      }
      for (int k = 0; k < d; k++) {
        SweepParams sp = dir_params(k);
        if (k == 0) { sp.out_mode = OUT_STORE; sp.out = op->W; }
        else if (k == d - 1) { sp.out_mode = OUT_ACC; sp.out = V; sp.acc = op->W; }
        else { sp.out_mode = OUT_ACC; sp.out = op->W; sp.acc = op->W; }
        HIPCHK(sweep_launch(op->laps[op->dims[k]], sp, st));
      }
      return 0;
    }
    for (int k = 0; k < op->d; k++) {
      SweepParams sp = {};
      sp.ncols = op->ncols_g[k]; sp.inner = op->inner_g[k];
      sp.in0 = U; sp.in_mode = IN_PLAIN;
      sp.alpha = -1.0;                                             // VecAXPY(w0,-1,.) elliptic.C:333
      if (k == 0 && op->d > 1) { sp.out_mode = OUT_STORE; sp.out = op->W; }
      else if (k == 0) { sp.out_mode = OUT_STORE; sp.out = V; }
      else if (k == op->d - 1) { sp.out_mode = OUT_ACC; sp.out = V; sp.acc = op->W; }
      else { sp.out_mode = OUT_ACC; sp.out = op->W; sp.acc = op->W; }
      HIPCHK(sweep_launch(op->laps[op->dims[k]], sp, st));
    }
    return 0;
  }
  // General coefficients: w0 = gather(U) on the fly (VecScatter GL + dirichlet0, elliptic.C:305-308),
  // V = scatter( -sum_k D_k( eta D_k w0 + deta w0 du0_k ) ); the gradient and the flux
  // (elliptic.C:309-323) never leave the chip.
  const bool f4 = op->wpad && ell_fused4_ok(op) && aligned16(U) && aligned16(V);
  if (!f4) { int rc = ell_sync_coeffs(op, st); if (rc) return rc; }      // the other kernels read eta / deta themselves
  if (op->cdirty) {
    for (int k = 0; k < op->d; k++) {
      if (op->coef_stale)     // state left by the interior-line FormFunction: the pairs come straight from w0
        hipLaunchKernelGGL(k_cprod_sq, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const double *)op->w0, op->coef_gamma,
                           (const double *)op->gradu[k], (double2 *)op->cprod[k]);
      else
        hipLaunchKernelGGL(k_cprod, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const double *)op->eta, (const double *)op->deta,
                           (const double *)op->gradu[k], (double2 *)op->cprod[k]);
    }
    op->cdirty = false;
  }
  if (f4) {
    for (int k = 0; k < op->d; k++) { int rc = ell_fused4_jacobian(op, k, U, V, st); if (rc) return rc; }
    return 0;
  }
  for (int k = 0; k < op->d; k++) {
    SweepParams sp = {};
    sp.ncols = op->ncols[k]; sp.inner = op->inner[k];
    sp.in0 = U; sp.in_mode = IN_GATHER;
    sp.coef_mode = COEF_FULL; sp.in1 = op->eta; sp.in2 = op->cprod[k];
    ell_out_chain(op, k, V, &sp);
    HIPCHK(fused_launch(op->mats[op->dims[k]], sp, st));
  }
  return 0;
}

extern "C" int ell_op_function(ell_op *op, double gamma, double exponent, const double *U,
                               const double *b, double *rhs, void *stream) {
  if (!op || ((!U || !rhs) && !(op->slab && op->G == 0))) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (U && (U == rhs || b == rhs)) return fail(CHEBHIP_ERR_ARG, "rhs must not alias U or b");
  StageTimer tm(CHEBHIP_STAGE_ELL_FUNCTION, stream);
  hipStream_t st = (hipStream_t)stream;
  int rc = ell_alloc_state(op);
  if (rc) return rc;
  const int d = op->d;
  const int iexp = (exponent == std::floor(exponent) && exponent >= 1.0 && exponent <= 8.0) ? (int)exponent : 0;
  {
    // Homogeneous Dirichlet rows and the reference's default exponent 2 (elliptic.C:141, :468-476) on the straight-line
    // kernel: three launches on the interior lines, no gather pass; eta / deta are formed from w0 when somebody reads them
    const bool notrim = opt(OPT_ETA_FROM_MEMORY) || opt(OPT_GATHER_PASS);
    if (!notrim && exponent == 2.0 && !op->dir_nonzero && op->wpad && op->G > 0 && ell_fused4_ok(op) && aligned16(U) && aligned16(rhs)) {
      if ((rc = ell_state_layout(op, true, st))) return rc;  // lines inside the boundary: zero in this state, not visited by the launches
      for (int k = 0; k < d; k++) if ((rc = ell_fused4_function_trim(op, k, gamma, U, b, rhs, st))) return rc;
      op->cdirty = true; op->coef_stale = true; op->coef_gamma = gamma;
      op->mode = (gamma == 0.0) ? COEFF_UNIT : COEFF_FULL;
      return 0;
    }
  }
  op->coef_stale = false;                                    // the pass below writes w0, eta and deta
  if ((rc = ell_state_layout(op, false, st))) return rc;
  if (op->dir_nonzero) op->bdy_lines_dirty = true;
  if ((op->N & 1) == 0)
    hipLaunchKernelGGL(k_gather_coeff2, dim3(pw_grid(op->N >> 1) * 2), dim3(256), 0, st, op->N, (const int *)op->ixL, U,
                       (const double *)op->dirloc, gamma, exponent, iexp, op->w0, op->eta, op->deta);   // elliptic.C:486-493, 508-509
  else
    hipLaunchKernelGGL(k_gather_coeff, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const int *)op->ixL, U,
                       (const double *)op->dirloc, gamma, exponent, iexp, op->w0, op->eta, op->deta);
  op->cdirty = true;
  // eta stays exactly 1 and deta exactly 0 only when gamma == 0 and no pow() can produce inf/nan
  const bool unit = (gamma == 0.0) && (exponent == std::floor(exponent)) && exponent >= 1.0;
  op->mode = unit ? COEFF_UNIT : COEFF_FULL;
  if (op->slab) {
    for (int k = 0; k < d; k++) if ((rc = ell_slab_sweep(op, k, op->w0, op->gradu[k], st))) return rc;     // :497-499
    // w_k = eta * gradu[k] (:511): dimension 0 needs it as an array (g[0]); the others form it on load
    HIPCHK(hipMemcpyAsync(op->g[0], op->gradu[0], (size_t)op->N * sizeof(double), hipMemcpyDeviceToDevice, st));
    std::vector<double *> src(op->gradu); src[0] = op->g[0];
    if (op->G == 0) { hipLaunchKernelGGL(k_flux, dim3(pw_grid(op->N)), dim3(256), 0, st, op->N, (const double *)op->eta, (const double *)nullptr,
                                         (const double *)nullptr, (const double *)nullptr, src[0]);
                      return op->dim0(op->dim0_ctx, 0, 1, src[0], nullptr, -1.0, op->W, st); }
    if ((rc = ell_slab_divergence(op, IN_FLUX_ETA, src.data(), rhs, st))) return rc;
    if (b) hipLaunchKernelGGL(k_axpy, dim3(pw_grid(op->G)), dim3(256), 0, st, op->G, -1.0, b, rhs);  // :530
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (op->has_long) {
    for (int k = 0; k < d; k++) if ((rc = ell_plain_sweep(op, k, op->w0, op->gradu[k], OUT_STORE, nullptr, 1.0, st))) return rc;   // :497-499
    if (op->G == 0) return 0;
    for (int k = 0; k < d; k++)                                                 // w_k = eta * gradu[k] (:511), formed in g[k]
      HIPCHK(hipMemcpyAsync(op->g[k], op->gradu[k], (size_t)op->N * sizeof(double), hipMemcpyDeviceToDevice, st));
    if ((rc = ell_plain_divergence(op, 1, op->g.data(), rhs, st))) return rc;
    if (b) hipLaunchKernelGGL(k_axpy, dim3(pw_grid(op->G)), dim3(256), 0, st, op->G, -1.0, b, rhs);  // :530
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (ell_fused4_ok(op) && aligned16(rhs)) {
    for (int k = 0; k < d; k++) if ((rc = ell_fused4_function(op, k, gamma, exponent, b, rhs, st))) return rc;   // includes rhs -= b (:530)
    return 0;
  } else {
    // fused: gradu[k] = D_k w0 is stored on the way (:497-499), w_k = eta gradu[k] (:511) feeds the
    // divergence without leaving the chip (:521-528)
    for (int k = 0; k < d; k++) {
      SweepParams sp = {};
      sp.ncols = op->ncols[k]; sp.inner = op->inner[k];
      sp.in0 = op->w0; sp.in_mode = IN_PLAIN;
      sp.coef_mode = COEF_ETA; sp.in1 = op->eta; sp.gout = op->gradu[k];
      ell_out_chain(op, k, rhs, &sp);
      if (op->G == 0) { sp.out_mode = OUT_STORE; sp.out = op->W; sp.acc = nullptr; }
      HIPCHK(fused_launch(op->mats[op->dims[k]], sp, st));
    }
    if (op->G == 0) return 0;
  }
  if (b) hipLaunchKernelGGL(k_axpy, dim3(pw_grid(op->G)), dim3(256), 0, st, op->G, -1.0, b, rhs);  // :530
  HIPCHK(hipGetLastError());
  return 0;
}

static int ell_stage(ell_op *op) {
  const size_t gb = (size_t)(op->G > 0 ? op->G : 1) * sizeof(double);
  if (!op->hU) { HIPCHK(hipMalloc((void **)&op->hU, gb)); HIPCHK(hipMalloc((void **)&op->hV, gb)); HIPCHK(hipMalloc((void **)&op->hB, gb)); }
  return 0;
}

extern "C" int ell_op_mult_host(ell_op *op, const double *U, double *V) {
  if (!op || !U || !V) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  int rc = ell_stage(op); if (rc) return rc;
  const size_t gb = (size_t)op->G * sizeof(double);
  HIPCHK(hipMemcpy(op->hU, U, gb, hipMemcpyHostToDevice));
  if ((rc = ell_op_mult(op, op->hU, op->hV, nullptr))) return rc;
  HIPCHK(hipMemcpy(V, op->hV, gb, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int ell_op_function_host(ell_op *op, double gamma, double exponent, const double *U,
                                    const double *b, double *rhs) {
  if (!op || !U || !rhs) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  int rc = ell_stage(op); if (rc) return rc;
  const size_t gb = (size_t)op->G * sizeof(double);
  HIPCHK(hipMemcpy(op->hU, U, gb, hipMemcpyHostToDevice));
  if (b) HIPCHK(hipMemcpy(op->hB, b, gb, hipMemcpyHostToDevice));
  if ((rc = ell_op_function(op, gamma, exponent, op->hU, b ? op->hB : nullptr, op->hV, nullptr))) return rc;
  HIPCHK(hipMemcpy(rhs, op->hV, gb, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int ell_op_set_dirichlet(ell_op *op, const double *values) {
  if (!op || !values) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  // expand the compact BlockIt-ordered boundary vector (elliptic.C:399-403) to the local layout
  std::vector<double> loc((size_t)op->N, 0.0);
  std::vector<int> ind(op->d, 0);
  long dd = 0;
  for (long l = 0; l < op->N; l++) {
    if (ell_is_bdy(op, ind.data())) loc[l] = values[dd++];
    for (int j = op->d - 1; j >= 0; j--) { if (++ind[j] < op->dims[j]) break; ind[j] = 0; }
  }
  if (!op->dirloc) HIPCHK(hipMalloc((void **)&op->dirloc, (size_t)op->N * sizeof(double)));
  HIPCHK(hipMemcpy(op->dirloc, loc.data(), (size_t)op->N * sizeof(double), hipMemcpyHostToDevice));
  op->dir_nonzero = false;
  for (long i = 0; i < dd && !op->dir_nonzero; i++) op->dir_nonzero = values[i] != 0.0;
  return 0;
}

static int ell_state_ptr(ell_op *op, int which, double **p) {
  int rc = ell_alloc_state(op);
  if (rc) return rc;
  if (which == 0) *p = op->eta;
  else if (which == 1) *p = op->deta;
  else if (which >= 2 && which < 2 + op->d) *p = op->gradu[which - 2];
  else return fail(CHEBHIP_ERR_ARG, "which = %d out of range", which);
  return 0;
}

extern "C" int ell_op_get_state(ell_op *op, int which, double *dst) {
  if (!op || !dst) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  double *p; int rc = ell_state_ptr(op, which, &p); if (rc) return rc;
  // a FormFunction still queued on a non-blocking stream must have written w0 before the null-stream rebuild of eta / deta reads it
  HIPCHK(hipDeviceSynchronize());
  if (which < 2 && (rc = ell_sync_coeffs(op, nullptr))) return rc;
  HIPCHK(hipStreamSynchronize(nullptr));
  HIPCHK(hipMemcpy(dst, p, (size_t)op->N * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int ell_op_set_state(ell_op *op, int which, const double *src) {
  if (!op || !src) return fail(CHEBHIP_ERR_ARG, "NULL argument");
  double *p; int rc = ell_state_ptr(op, which, &p); if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());                               // as in ell_op_get_state: callbacks on non-blocking streams first
  if ((rc = ell_sync_coeffs(op, nullptr))) return rc;           // the untouched one of eta / deta must be current
  HIPCHK(hipStreamSynchronize(nullptr));
  HIPCHK(hipMemcpy(p, src, (size_t)op->N * sizeof(double), hipMemcpyHostToDevice));
  if (which >= 2) op->bdy_lines_dirty = true;
  op->mode = COEFF_FULL; op->cdirty = true;
  return 0;
}
