// sweep.h -- internal interface between the C-ABI layer and the gfx950 sweep kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace chebhip {

// How one element of the line being differentiated is produced on load.
enum InMode : int {
  IN_PLAIN = 0,     // v = in0[a]
  IN_GATHER = 1,    // v = interior ? in0_global[g] : 0            (VecScatter GL + dirichlet0, elliptic.C:305-308)
  IN_FLUX_ETA = 2,  // v = in1[a] * in0[a]                         (eta * g, elliptic.C:511)
  IN_FLUX_FULL = 3, // v = in1[a]*in0[a] + in2[a]*in3[a]*in4[a]    (eta*g + deta*u*du0, elliptic.C:321)
  IN_SUM3 = 4,      // v = (in0[a] + in1[a]) + in2[a]              (lines of at most 64 points, 16-byte kernels only: the
                    //   divergence of the uniform-viscosity Stokes path as the sum of its three terms, stokes.hip)
  IN_MUL = 5        // v = in0[a] * in1[a]                         (lines of more than 64 points, 16-byte kernels with raw = 1 only:
                    //   the 1 / eta of the fast-diagonalisation solve folded into its first forward line transform, precond.hip)
};

// What happens to one element r of the derivative on store.
enum OutMode : int {
  OUT_STORE = 0,        // out[a] = alpha*r
  OUT_ACC = 1,          // out[a] = acc[a] + alpha*r               (VecAXPY, elliptic.C:333)
  OUT_ACC_SCATTER = 2,  // interior: out_global[g] = (acc ? acc[a] : 0) + alpha*r   (+ VecScatter LG, elliptic.C:336)
  OUT_ACC2 = 4,         // out[a] = (acc[a] + acc2[a]) + alpha*r      (16-byte kernels, lines of at most 128 points: the last direction of the
                        //   constant-coefficient MatMult_Elliptic on small grids takes the two earlier terms as they are, chebhip.hip)
  OUT_MUL = 3           // out[a] = acc[a] * (alpha*r)               (16-byte kernels with raw = 1 only: the modal scaling of the
                        //   fast-diagonalisation solve folded into its last forward line transform, precond.hip)
};

// Pointwise coefficient applied between the two halves of a fused D_k( flux( D_k u ) ) launch.
enum CoefMode : int {
  COEF_UNIT = 0,   // f = g                                   (eta == 1, deta == 0)
  COEF_ETA = 1,    // f = eta * g                             (elliptic.C:511)
  COEF_FULL = 2    // f = eta * g + c * u,  c = deta * du0    (elliptic.C:321; c is formed once per linearisation)
};

struct SweepParams {
  int P;              // points along the transform dim
  int H;              // ceil(P/2): even/odd half length
  unsigned ncols;     // number of lines = N / P
  unsigned inner;     // stride (elements) of the transform dim in the local layout
  const double *in0, *in1, *in2, *in3, *in4;
  double *out;
  const double *acc;
  const double *acc2; // OUT_ACC2: the second operand, geometry of acc
  const int *gcol;    // [ncols] global index of the (j=1) node of a line, -1 for boundary lines
  long gstride;       // stride of the transform dim in the global (interior) layout
  double alpha;
  int in_mode, out_mode;
  const double *fragE, *fragO;  // differentiation matrix halves in MFMA fragment order
  const double *fragE2, *fragO2; // the same with the fragments of k-steps 2g, 2g+1 side by side ([m-tile][KS/2][64 lanes][2]): 16-byte loads
  const double *longDT;         // lines of more than 256 points: dense D^T ([j][i], P x P) instead of fragments
  const double *longD;          // ... and dense D (row-major) for the library-GEMM route
  const double *zero;           // a few zero doubles in HBM: the source of every masked-off load
  double *sink;                 // 512 x 16 B in HBM: where masked-off stores of the straight-line kernel go
  unsigned ntiles;
  int sym;            // mirror rows are a - b (centro-symmetric matrix) instead of b - a
  int raw;            // 16-byte kernels, STORE only (the line transforms of precond.hip): 1 = the two halves of the product are
                      //   stored as they are (row i <- (ME e)_i, row m-i <- (MO o)_i) instead of being recombined;
                      //   2 = the input is taken as already split (e_j = x_j, o_j = x_{m-j})
  int trim;           // fused launches only: arrays in0/out/acc hold interior points only (see fused.hip)
  int coef_mode;      // fused launches only: CoefMode; COEF_ETA: eta = in1; COEF_FULL: in2 = pairs {eta, c / 2}, c = deta * du0 (local layout)
  double *gout;       // fused launches only: if non-null the gradient g = D u is also stored here (local layout)
  // Per-array geometry of the 16-byte kernels (sweep_vec.hip, v3 / v4); 0 = derive from P / inner / ncols.
  //   COLFAST: a tile is (outer block o, NT neighbouring columns q < qmax); element (o, q, point j) of
  //            array X sits at o * X_os + q + j * X_rs.  The default is the dense tensor: nouter =
  //            ncols / inner, qmax = inner, X_os = P * inner, X_rs = inner.
  //   JFAST:   line c of array X starts at c * X_os (default X_os = P), points are contiguous.
  // Lets the accumulator of the constant-coefficient operator keep rows padded to 128 B while the
  // MatShell vectors stay in the reference's dense interior layout.
  unsigned nouter, qmax;
  unsigned in_os, in_rs, acc_os, acc_rs, out_os, out_rs;
  unsigned in_bytes, acc_bytes, out_bytes;   // set by the launcher: exact sizes for the buffer descriptors of v4
  // Input made of stacked FIELDS that are not contiguous (cheb_sweep_vec4_kernel only): field f of the input starts
  // in_fskip elements further than the dense stacking would put it, per field (i.e. at f * (field size + in_fskip)).
  // in_fblocks = outer blocks (COLFAST) / lines (JFAST, a multiple of the tile's lines) per field; 0 = dense.
  // The Stokes stress tensor is stored as its 6 distinct components this way (stokes.hip).
  unsigned in_fblocks, in_fskip;
  unsigned in_fblocks_inv;                   // set by the launcher: ceil(2^32 / in_fblocks)
  unsigned gfield0;                          // gather launches (GatherSrc): the job's outer blocks are the vectors / fields gfield0, gfield0 + 1, .. of the arrays
};

// Host description of the even/odd split differentiation matrices for P points.
struct DiffMat {
  int P = 0, H = 0;
  int KS = 0;          // k-steps of 4 (power of two >= 4 for the register-resident kernel); 0 = long lines (longDT)
  int MTP = 0;         // padded m-tiles of 16 rows = KS/4
  double *fragE = nullptr, *fragO = nullptr;  // device, [MTP][KS][64]
  double *fragE2 = nullptr, *fragO2 = nullptr; // device, [MTP][KS/2][64][2] (same allocation as fragO)
  double *longDT = nullptr;                   // device, dense D^T for P > 256 (then fragE holds only zero/sink)
  double *longD = nullptr;                    // device, dense D (row-major) for the library-GEMM route of long lines
  double *zero = nullptr;                     // device, 8 zero doubles (tail of the fragE allocation)
  double *sink = nullptr;                     // device, 1024 doubles after `zero`: target of masked-off stores
  int sym = 0;                                // 0: centro-antisymmetric (D), 1: centro-symmetric (interior D D)
  int xl_ks = 0;                              // long lines of up to 1024 points (KS == 0): k-steps of the fragment arrays for cheb_sweep_xl_kernel
};

// Builds (in long double) and uploads the fragment-ordered matrices.  Returns hipSuccess or error.
hipError_t diffmat_create(int P, DiffMat *out);
// Interior second-derivative operator (D D)[1..n-1,1..n-1] of a zero-Dirichlet line of P points; the
// returned DiffMat describes lines of P-2 stored points.
hipError_t diffmat_create_lap(int P, DiffMat *out);
// D acting on the interior values of a line whose two end values are extrapolated from them (3 <= P <= 256; see diffmat.cpp)
hipError_t diffmat_create_pext(int P, DiffMat *out);
// D D on all P points of a line (3 <= P <= 256; centro-symmetric): the second derivative of a line that carries its end values
hipError_t diffmat_create_dd(int P, DiffMat *out);
void diffmat_destroy(DiffMat *m);
// Fragments of an arbitrary centro-symmetric (sym = 1) / centro-antisymmetric (sym = 0) dense M x M matrix, M <= 256.
hipError_t diffmat_from_dense(int M, const long double *A, int sym, DiffMat *out);
// The same from the two H x H blocks themselves (H = ceil(M/2), row-major): y_i = (ME e)_i + (MO o)_i etc.
hipError_t diffmat_from_blocks(int M, const long double *ME, const long double *MO, int sym, DiffMat *out);
// Host-side dense differentiation matrix (row-major P x P), for tests and the adapter.
void diffmat_dense_host(int P, double *D);

// Launches one sweep.  jfast selects the line-contiguous tiling.
hipError_t sweep_launch(const DiffMat &m, SweepParams p, hipStream_t stream);

// Launches one fused pair: out (+)= alpha * D( coef( D in ) ) along the plan's dimension.
// in_mode must be IN_PLAIN or IN_GATHER.
hipError_t fused_launch(const DiffMat &m, SweepParams p, hipStream_t stream);
void sweep_note_launch();

// Straight-line fused kernel (fused4.hip).  The launch walks a line space of nouter blocks x qmax lines; element
// (block o, line q, point j) of array X sits at o * X.os + q * X.ls + j * X.rs, counted in elements of X (COLFAST: ls = 1,
// JFAST: rs = 1).  Arrays marked "trimmed" hold the points 1..n-1 of a line only (point j at (j-1) * rs): the interior
// layout of the MatShell vectors.
struct F4Geom { unsigned os, ls, rs; };
struct Fused4Params {
  int P, H;
  unsigned nouter, qmax, ntiles, tpo, tpo_inv;    // tpo = tiles per block, tpo_inv = ceil(2^32 / tpo)
  const double *in; F4Geom gi; unsigned in_bytes;              // trimmed in the Jacobian mode (full), whole lines otherwise
  const void *coef; F4Geom gc; unsigned coef_bytes;            // full: pairs {eta, c / 2} (16 B); otherwise eta (8 B)
  double *gout; unsigned gout_bytes;                           // eta mode: g = D u is stored here, geometry gc
  const double *acc; F4Geom ga; unsigned acc_bytes;            // trimmed in the Jacobian mode
  double *out; F4Geom go; unsigned out_bytes;                  // trimmed in the Jacobian mode and in the window mode
  const double *sub; unsigned sub_bytes;                       // window mode / last trimf launch: out -= sub (geometry go), may be null
  int trimf;                                                   // FormFunction on the interior line space: `in`, `acc`, `out` trimmed as in the Jacobian mode, eta on chip
                                                               //   (eta_square), gout through a base shifted to the first interior line (geometry gc); homogeneous Dirichlet rows only
  double *w0out; unsigned w0_bytes;                            // trimf: if non-null the line itself is stored here (geometry gc): the local copy w0 of the state
  double alpha;
  int eta_square; double gamma4;                               // eta mode: eta = 1 + 4 gamma4 u^2 is formed from the line itself (coef is not read)
  const double *fragE2, *fragO2;
};
// full: Jacobian mode (coefficient pairs, trimmed vectors); acc: out = acc + alpha t; win (JFAST only): `out` and `sub`
// are indexed by (o - 1, q - 1) and exist only for 1 <= o <= nouter - 2 (nouter > 1), 1 <= q <= qmax - 2
bool fused4_eligible(const DiffMat &m);
hipError_t fused4_launch(const DiffMat &m, Fused4Params p, bool jfast, bool full, bool acc, bool win, hipStream_t stream);

// Lines of 257 .. 1024 points on the matrix cores (sweep_xl.hip): plain input, STORE / ACC output
bool sweep_xl_eligible(const DiffMat &m, const SweepParams &p);
hipError_t sweep_xl_launch(const DiffMat &m, SweepParams p, hipStream_t stream);

// 16-byte-access specialisation (sweep_vec.hip); used by sweep_launch when eligible
bool sweep_vec_eligible(const DiffMat &m, const SweepParams &p);
// ... and may carry p.raw != 0 (the kernel generations that implement the raw modes)
bool sweep_vec_raw_eligible(const DiffMat &m, const SweepParams &p);
hipError_t sweep_vec_launch(const DiffMat &m, SweepParams p, hipStream_t stream);
// Lines along the OUTERMOST dimension of a tensor whose planes live in up to GATHER_MAX different arrays (the pencil of a slab
// partition read straight from the ranks' slabs, dist.hip): row i of every line is plane i - s0[s] of array p[s] (s: s0[s] <= i <
// s0[s+1]), the columns [col0, col0 + qmax) of that plane, rowlen doubles per plane; vector o of a batch starts lq[s] doubles further.
// pmax[s] clamps the plane index (the NULL transport reads one array with every rank's geometry).
// push != 0: the RESULT goes the same way back -- row i of every output line is stored into array dp[s] of the plane's owner (same
// geometry as p[s]: plane i - s0[s], columns [col0, col0 + qmax), vectors lq[s] apart) instead of the dense output; `out` of the sweep
// must still be a valid array of at least 16 KiB (lanes with nothing to store write there).
constexpr int GATHER_MAX = 16;
struct GatherSrc { const double *p[GATHER_MAX]; int s0[GATHER_MAX + 1]; unsigned lq[GATHER_MAX]; int pmax[GATHER_MAX]; int G; unsigned rowlen, col0;
                   double *dp[GATHER_MAX]; int push; };
// the launch (plain input, STORE, strided lines of 66 .. 256 points, dense output); *done = false: not eligible, nothing launched
hipError_t sweep_launch_gather(const DiffMat &m, SweepParams p, const GatherSrc &g, hipStream_t stream, bool *done);
hipError_t sweep_vec_launch_gather(const DiffMat &m, SweepParams p, const GatherSrc &g, hipStream_t stream, bool *done);
hipError_t sweep_vec_launch_multi(int n, const DiffMat *const *m, SweepParams *jobs, hipStream_t stream, bool *done);
hipError_t sweep_vec_launch_multi_gather(int n, const DiffMat *const *m, SweepParams *jobs, unsigned gmask, const GatherSrc &g, hipStream_t stream, bool *done);
// n sweeps as ONE launch, the jobs of gmask (bit j = job j) reading their lines through g; *done = false: they cannot share a launch, nothing is launched
hipError_t sweep_launch_multi_gather_try(int n, const DiffMat *const *m, const SweepParams *p, unsigned gmask, const GatherSrc &g, hipStream_t stream, bool *done);
// n independent sweeps (plain in, STORE out): one launch when they qualify (sweep_vec.hip), else n launches
hipError_t sweep_launch_multi(int n, const DiffMat *const *m, const SweepParams *p, hipStream_t stream);
// ... only if they can share ONE launch (*done = true); otherwise nothing is launched (*done = false)
hipError_t sweep_launch_multi_try(int n, const DiffMat *const *m, const SweepParams *p, hipStream_t stream, bool *done);

long sweep_launch_count();
}  // namespace chebhip
struct cheb_plan;
struct stokes_op;
struct ell_op;
namespace chebhip {
// chebhip.hip: ell_op_pencil_sweep with the pencil's planes read in place from the ranks' slab field (slabx.hip)
bool ell_pencil_gather_supported(const ell_op *op);
int ell_pencil_gather_try(ell_op *op, long ncol, const GatherSrc &g, double *out, hipStream_t st, bool *done);
// stokes.hip: the dimension-0 sweeps of a slab-mode Stokes callback (stokes_op_pencil_sweep / _pressure / _sweep_pressure) with the
// pencil's planes read from the arrays of g (the ranks' slab fields) instead of a materialised pencil; *done = false: not eligible
int stokes_pencil_gather_try(stokes_op *op, int kind, int nf, long ncol, const GatherSrc &g, double *out, hipStream_t st, bool *done);
bool stokes_pencil_gather_supported(const stokes_op *op);     // the matrices of dimension 0 are the long-line kernel's and carry the extrapolation
// chebhip.hip: the interior second derivative along n directions of one tensor as ONE launch (see there)
int lap1d_multi_try(int n, cheb_plan *const *plans, const double *x, double *const *outs, double alpha, hipStream_t st, bool *done);
int lap1d_gather_try(cheb_plan *p, const GatherSrc &g, double alpha, double *y, hipStream_t st, bool *done);
// the n local directions (x) and the gather direction (plan gp over the arrays of g) as ONE launch of n + 1 jobs
int lap1d_multi_gather_try(int n, cheb_plan *const *plans, const double *x, double *const *outs, cheb_plan *gp, const GatherSrc &g, double *gout,
                           double alpha, hipStream_t st, bool *done);
// compute units of the CURRENT device (cached per device id); 0 on error with *err set
int sweep_num_cus(hipError_t *err);

// Run-time options of the library (chebhip_set_option, include/chebhip.h): named integer switches read where they apply.
// Nothing in the library reads the environment.
enum OptId { OPT_GENERAL_KERNELS = 0, OPT_SEPARATE_LAUNCHES, OPT_VENDOR_GEMM, OPT_NO_RAW_TRANSFORMS, OPT_EQUAL_SHARES, OPT_FORCE_GEMM,
             OPT_STOKES_SINGLE_STREAM, OPT_ETA_FROM_MEMORY, OPT_GATHER_PASS, OPT_RCCL_SELF_MESSAGES, OPT_LOCAL_TIMEOUT_S, OPT_FULL_STRESS, OPT_DIST_SINGLE_STREAM, OPT_LONG_LINES_GEMM, OPT_PRESSURE_PASSES, OPT_GENERAL_VISCOUS, OPT_POISSON_LAUNCHES, OPT_DIST_EXACT_ORDER, OPT_FDM_PASSES, OPT_SADDLE_NODE_MAJOR, OPT_STOKES_Z_SEPARATE, OPT_FDM_Z_SEPARATE, OPT_STOKES_PRESSURE_STREAM, OPT_KRYLOV_EXACT_NORM, OPT_STOKES_PRESSURE_SWEEPS, OPT_DIST_PACKED_EXCHANGE, OPT_COUNT };
int opt(int id);
void opt_set(int id, int value);
const char *opt_name(int id);
int opt_find(const char *name);      // -1: no such option

// Fast diagonalisation of the 1-D three-point operator of the finite-difference preconditioners (diffmat.cpp)
// Modes are ordered by parity: position p < ceil(M/2) holds the p-th even mode, position M-1-q the q-th odd one
bool fdm_line(int P, std::vector<long double> &S, std::vector<long double> &Sinv, std::vector<long double> &lam);
void centro_part(int M, const std::vector<long double> &A, int part, std::vector<long double> &out);

}  // namespace chebhip
