// diffmat.cpp -- Chebyshev collocation differentiation matrix on the Gauss-Lobatto nodes
// x_i = cos(i pi/n), split by parity and laid out as f64 MFMA operand fragments.
//
// The reference never forms this matrix: chebyshev.c:142-199 applies it as
// DCT-I -> (times k) -> DST-I -> /(2n sin) plus two endpoint sums.  In exact arithmetic
// that chain IS multiplication by D below (the derivative of the degree-n interpolant at
// the nodes), so y = D x reproduces ChebMult to rounding.  P = 256 and 128 make the
// logical FFT length 2(P-1) = 510 = 2*3*5*17 and 254 = 2*127, hostile to butterflies; a
// dense product on the matrix cores, halved by the centro-antisymmetry of D, is both
// faster on gfx950 and more accurate (see DESIGN.md "Why dense").
//
// Parity split (n = P-1, H = ceil(P/2)); for j < H with 2j != n:
//     e_j = x_j + x_{n-j},  o_j = x_j - x_{n-j};   self-paired middle (2j == n): e_j = x_j, o_j = 0
//     ME[i][j] = (D[i][j] + D[i][n-j]) / 2   (middle column: D[i][j])
//     MO[i][j] = (D[i][j] - D[i][n-j]) / 2   (middle column: 0)
// and because D[n-i][n-j] = -D[i][j]:
//     y_i = (ME e)_i + (MO o)_i,      y_{n-i} = (MO o)_i - (ME e)_i,     i < H.
#include "sweep.h"
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <vector>

namespace chebhip {

static const long double PI_L = 3.14159265358979323846264338327950288L;

// D[i][j] in long double.  Off-diagonal: (c_i/c_j) (-1)^(i+j) / (x_i - x_j) with
// x_i - x_j = -2 sin((i+j) pi/2n) sin((i-j) pi/2n) (no cancellation); diagonal from the
// closed forms  D00 = (2n^2+1)/6 = -Dnn,  Dii = -x_i / (2 sin^2(i pi/n)).
static long double dentry(int i, int j, int n) {
  if (i == j) {
    if (i == 0) return (2.0L * n * n + 1.0L) / 6.0L;
    if (i == n) return -(2.0L * n * n + 1.0L) / 6.0L;
    long double s = sinl(PI_L * i / n);
    return -cosl(PI_L * i / n) / (2.0L * s * s);
  }
  long double ci = (i == 0 || i == n) ? 2.0L : 1.0L;
  long double cj = (j == 0 || j == n) ? 2.0L : 1.0L;
  long double sgn = ((i + j) & 1) ? -1.0L : 1.0L;
  long double dx = -2.0L * sinl(PI_L * (i + j) / (2.0L * n)) * sinl(PI_L * (i - j) / (2.0L * n));
  return (ci / cj) * sgn / dx;
}

void diffmat_dense_host(int P, double *D) {
  const int n = P - 1;
  for (int i = 0; i < P; i++)
    for (int j = 0; j < P; j++) D[(size_t)i * P + j] = (double)dentry(i, j, n);
}

// [m-tile][k-step][lane] -> [m-tile][k-step pair][lane][2]: a lane fetches two fragments with one 16-byte load
static hipError_t upload_paired(const std::vector<double> &f, int MTP, int KS, double *dev) {
  std::vector<double> g(f.size());
  for (int mt = 0; mt < MTP; mt++)
    for (int s = 0; s < KS; s++)
      for (int l = 0; l < 64; l++)
        g[(((size_t)mt * (KS / 2) + s / 2) * 64 + l) * 2 + (s & 1)] = f[((size_t)mt * KS + s) * 64 + l];
  return hipMemcpy(dev, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice);
}

// P > 256: the matrix does not fit the register file of a workgroup; cheb_sweep_long_kernel streams the dense
// transpose from L2 instead (a correctness path for any extent the reference accepts, not a tuned one).
static hipError_t diffmat_create_long(int P, DiffMat *out) {
  std::vector<double> DT((size_t)P * P), D((size_t)P * P);
  const int n = P - 1;
  for (int i = 0; i < P; i++) for (int j = 0; j < P; j++) { const double v = (double)dentry(i, j, n); DT[(size_t)j * P + i] = v; D[(size_t)i * P + j] = v; }
  DiffMat m;
  m.P = P; m.H = (P + 1) / 2; m.KS = 0; m.MTP = 0;
  // Lines of up to 1024 points: the even / odd halves in MFMA-operand order for cheb_sweep_xl_kernel (sweep_xl.hip):
  // [m-tile][k-step][64 lanes], m-tiles padded by a workgroup's worth (16), k-steps to a multiple of 8; zero padded.
  std::vector<double> fe, fo;
  size_t cnt = 0;
  if (P <= 1024) {
    const int H = m.H;
    m.MTP = (H + 15) / 16 + 16;                  // (a workgroup covers up to 16 m-tiles: its last one starts below ceil(H/16))
    m.xl_ks = ((H + 3) / 4 + 7) / 8 * 8;
    cnt = (size_t)m.MTP * m.xl_ks * 64;
    fe.assign(cnt, 0.0); fo.assign(cnt, 0.0);
    for (int mt = 0; mt < m.MTP; mt++)
      for (int s = 0; s < m.xl_ks; s++)
        for (int l = 0; l < 64; l++) {
          const int i = mt * 16 + (l & 15), j = 4 * s + (l >> 4);
          if (i >= H || j >= H) continue;
          long double me, mo;
          if (2 * j == n) { me = dentry(i, j, n); mo = 0.0L; }
          else { const long double a = dentry(i, j, n), b = dentry(i, n - j, n); me = 0.5L * (a + b); mo = 0.5L * (a - b); }
          fe[((size_t)mt * m.xl_ks + s) * 64 + l] = (double)me;
          fo[((size_t)mt * m.xl_ks + s) * 64 + l] = (double)mo;
        }
  }
  hipError_t e = hipMalloc((void **)&m.fragE, (cnt + 8 + 1024) * sizeof(double));
  if (e != hipSuccess) return e;
  m.zero = m.fragE + cnt; m.sink = m.zero + 8;
  e = hipMemset(m.zero, 0, 8 * sizeof(double));
  if (e == hipSuccess && cnt) e = hipMemcpy(m.fragE, fe.data(), cnt * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess && cnt) e = hipMalloc((void **)&m.fragO, cnt * sizeof(double));
  if (e == hipSuccess && cnt) e = hipMemcpy(m.fragO, fo.data(), cnt * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void **)&m.longDT, DT.size() * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(m.longDT, DT.data(), DT.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void **)&m.longD, D.size() * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(m.longD, D.data(), D.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(m.fragE); if (m.fragO) (void)hipFree(m.fragO); if (m.longDT) (void)hipFree(m.longDT); if (m.longD) (void)hipFree(m.longD); return e; }
  *out = m;
  return hipSuccess;
}

hipError_t diffmat_create(int P, DiffMat *out) {
  // option "force_gemm" (A/B measurements only): every extent takes the long-line route, i.e. a library DGEMM
  const int force = opt(OPT_FORCE_GEMM);
  if (P > 256 || (force && P >= 4)) return diffmat_create_long(P, out);
  const int n = P - 1;
  const int H = (P + 1) / 2;
  int KS = 4;
  while (4 * KS < H) KS *= 2;
  const int MTP = KS / 4;
  const size_t cnt = (size_t)MTP * KS * 64;
  std::vector<double> fe(cnt, 0.0), fo(cnt, 0.0);
  for (int mt = 0; mt < MTP; mt++)
    for (int s = 0; s < KS; s++)
      for (int l = 0; l < 64; l++) {
        const int i = mt * 16 + (l & 15);  // output row held by this lane
        const int j = 4 * s + (l >> 4);    // reduction index
        if (i >= H || j >= H) continue;
        long double me, mo;
        if (2 * j == n) { me = dentry(i, j, n); mo = 0.0L; }
        else {
          const long double a = dentry(i, j, n), b = dentry(i, n - j, n);
          me = 0.5L * (a + b); mo = 0.5L * (a - b);
        }
        fe[((size_t)mt * KS + s) * 64 + l] = (double)me;
        fo[((size_t)mt * KS + s) * 64 + l] = (double)mo;
      }
  DiffMat m;
  m.P = P; m.H = H; m.KS = KS; m.MTP = MTP;
  hipError_t e = hipMalloc((void **)&m.fragE, (cnt + 8 + 1024) * sizeof(double));
  if (e != hipSuccess) return e;
  e = hipMalloc((void **)&m.fragO, 3 * cnt * sizeof(double));
  if (e != hipSuccess) { (void)hipFree(m.fragE); return e; }
  m.fragE2 = m.fragO + cnt; m.fragO2 = m.fragO + 2 * cnt;
  m.zero = m.fragE + cnt;
  m.sink = m.zero + 8;
  e = hipMemset(m.zero, 0, 8 * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(m.fragE, fe.data(), cnt * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m.fragO, fo.data(), cnt * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = upload_paired(fe, MTP, KS, m.fragE2);
  if (e == hipSuccess) e = upload_paired(fo, MTP, KS, m.fragO2);
  if (e != hipSuccess) { (void)hipFree(m.fragE); (void)hipFree(m.fragO); return e; }
  *out = m;
  return hipSuccess;
}

// Fragments of a dense M x M matrix A (row-major, long double) that is centro-symmetric (sym = 1:
// A[m-i][m-j] = A[i][j], m = M-1) or centro-antisymmetric (sym = 0).  With e, o as above
//     y_i = (ME e)_i + (MO o)_i,   y_{m-i} = (ME e)_i - (MO o)_i   (sym = 1; sym = 0: (MO o)_i - (ME e)_i).
hipError_t diffmat_from_dense(int M, const long double *A, int sym, DiffMat *out) {
  if (M < 1 || M > 256) return hipErrorInvalidValue;
  const int m = M - 1, H = (M + 1) / 2;
  std::vector<long double> ME((size_t)H * H), MO((size_t)H * H);
  for (int i = 0; i < H; i++)
    for (int j = 0; j < H; j++) {
      long double me, mo;
      if (2 * j == m) { me = A[(size_t)i * M + j]; mo = 0.0L; }
      else {
        const long double a = A[(size_t)i * M + j], b = A[(size_t)i * M + (m - j)];
        me = 0.5L * (a + b); mo = 0.5L * (a - b);
      }
      ME[(size_t)i * H + j] = me; MO[(size_t)i * H + j] = mo;
    }
  return diffmat_from_blocks(M, ME.data(), MO.data(), sym, out);
}

hipError_t diffmat_from_blocks(int M, const long double *ME, const long double *MO, int sym, DiffMat *out) {
  if (M < 1 || M > 256) return hipErrorInvalidValue;
  const int H = (M + 1) / 2;
  int KS = 4;
  while (4 * KS < H) KS *= 2;
  const int MTP = KS / 4;
  const size_t cnt = (size_t)MTP * KS * 64;
  std::vector<double> fe(cnt, 0.0), fo(cnt, 0.0);
  for (int mt = 0; mt < MTP; mt++)
    for (int s = 0; s < KS; s++)
      for (int l = 0; l < 64; l++) {
        const int i = mt * 16 + (l & 15), j = 4 * s + (l >> 4);
        if (i >= H || j >= H) continue;
        fe[((size_t)mt * KS + s) * 64 + l] = (double)ME[(size_t)i * H + j];
        fo[((size_t)mt * KS + s) * 64 + l] = (double)MO[(size_t)i * H + j];
      }
  DiffMat r;
  r.P = M; r.H = H; r.KS = KS; r.MTP = MTP; r.sym = sym;
  hipError_t e = hipMalloc((void **)&r.fragE, (cnt + 8 + 1024) * sizeof(double));
  if (e != hipSuccess) return e;
  e = hipMalloc((void **)&r.fragO, 3 * cnt * sizeof(double));
  if (e != hipSuccess) { (void)hipFree(r.fragE); return e; }
  r.fragE2 = r.fragO + cnt; r.fragO2 = r.fragO + 2 * cnt;
  r.zero = r.fragE + cnt;
  r.sink = r.zero + 8;
  e = hipMemset(r.zero, 0, 8 * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(r.fragE, fe.data(), cnt * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(r.fragO, fo.data(), cnt * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = upload_paired(fe, MTP, KS, r.fragE2);
  if (e == hipSuccess) e = upload_paired(fo, MTP, KS, r.fragO2);
  if (e != hipSuccess) { (void)hipFree(r.fragE); (void)hipFree(r.fragO); return e; }
  *out = r;
  return hipSuccess;
}

// Second-derivative operator of a zero-Dirichlet line, restricted to its interior:
//   L = (D D)[1..n-1, 1..n-1],  M = P-2 points.
// For constant coefficients the two sweeps of a direction, D_k (1 * D_k w0) with w0 = 0 at both ends
// (elliptic.C:305-334 with eta = 1, deta = 0), collapse into y = L x on the interior values.  L is
// centro-SYMMETRIC.  The product is formed in long double and rounded once.
hipError_t diffmat_create_lap(int P, DiffMat *out) {
  const int n = P - 1, M = P - 2;
  std::vector<long double> D((size_t)P * P), L((size_t)M * M);
  for (int i = 0; i < P; i++) for (int j = 0; j < P; j++) D[(size_t)i * P + j] = dentry(i, j, n);
  for (int i = 0; i < M; i++)
    for (int j = 0; j < M; j++) {
      long double s = 0.0L;
      for (int q = 0; q < P; q++) s += D[(size_t)(i + 1) * P + q] * D[(size_t)q * P + (j + 1)];
      L[(size_t)i * M + j] = s;
    }
  return diffmat_from_dense(M, L.data(), 1, out);
}

// D with the end values of a line replaced by their extrapolation from the interior (StokesPressureReduceOrder,
// stokes.C:1029-1080: the degree-(P-3) polynomial through the interior values, evaluated at both ends):
//     x_0 = sum_j w0_j x_j,  x_n = sum_j w1_j x_j   (j = 1 .. n-1, Lagrange weights)
//     D (x_0, x_1 .. x_{n-1}, x_n)^T = Dext x,   Dext[i][j] = D[i][j] + D[i][0] w0_j + D[i][n] w1_j,  Dext[i][0] = Dext[i][n] = 0.
// w1_j = w0_{n-j} (mirror nodes), so Dext is centro-antisymmetric like D and runs on the same kernels.  Formed in long
// double, rounded once.  The interior values of a pressure-gradient line never depend on the extrapolations along the
// OTHER directions, so a Stokes callback needs no extrapolation pass at all (stokes.hip).
hipError_t diffmat_create_pext(int P, DiffMat *out) {
  if (P < 3 || P > 256) return hipErrorInvalidValue;
  const int n = P - 1, m = P - 2;
  std::vector<long double> x(P), w0(P, 0.0L), w1(P, 0.0L), A((size_t)P * P, 0.0L);
  for (int i = 0; i < P; i++) x[i] = cosl(PI_L * i / n);
  for (int j = 1; j <= m; j++) {
    long double l0 = 1.0L, l1 = 1.0L;
    for (int q = 1; q <= m; q++) if (q != j) { l0 *= (x[0] - x[q]) / (x[j] - x[q]); l1 *= (x[n] - x[q]) / (x[j] - x[q]); }
    w0[j] = l0; w1[j] = l1;
  }
  for (int i = 0; i < P; i++)
    for (int j = 1; j <= m; j++) A[(size_t)i * P + j] = dentry(i, j, n) + dentry(i, 0, n) * w0[j] + dentry(i, n, n) * w1[j];
  return diffmat_from_dense(P, A.data(), 0, out);
}

// D D on all P points (rows and columns 0 .. n): formed in long double, rounded once.  D is centro-antisymmetric, so the
// product is centro-symmetric.
hipError_t diffmat_create_dd(int P, DiffMat *out) {
  if (P < 3 || P > 256) return hipErrorInvalidValue;
  const int n = P - 1;
  std::vector<long double> D((size_t)P * P), A((size_t)P * P);
  for (int i = 0; i < P; i++) for (int j = 0; j < P; j++) D[(size_t)i * P + j] = dentry(i, j, n);
  for (int i = 0; i < P; i++)
    for (int j = 0; j < P; j++) {
      long double s = 0.0L;
      for (int q = 0; q < P; q++) s += D[(size_t)i * P + q] * D[(size_t)q * P + j];
      A[(size_t)i * P + j] = s;
    }
  return diffmat_from_dense(P, A.data(), 1, out);
}

// ---------------------------------------------------------------------------------------------
// Fast diagonalisation of the finite-difference preconditioner (elliptic.C:556-579 with eta = 1, deta = 0;
// stokes.C:1181-1226 per velocity component): on the tensor grid that matrix is  sum_k I x .. x T_k x .. x I  with
// the 1-D three-point operator T on the interior Gauss-Lobatto nodes,
//     (T u)_i = -idx (idxM u_{i-1} + idxP u_{i+1}) + idx (idxP + idxM) u_i,
//     idxM = 1/(x_i - x_{i-1}), idxP = 1/(x_{i+1} - x_i), idx = 1/(xP - xM), xM, xP the midpoints.
// T = H^-1 K with H = diag(-(xP - xM)) > 0 and K symmetric positive definite, so T = S Lambda S^-1 with
// S = H^-1/2 W, S^-1 = W^T H^1/2, W the orthonormal eigenvectors of H^-1/2 K H^-1/2 (implicit QL, long double).
// ---------------------------------------------------------------------------------------------
// EISPACK tql2: eigenvalues d[] and eigenvectors z (row-major n x n, identity on entry) of the symmetric
// tridiagonal matrix with diagonal d[] and sub-diagonal e[1..n-1] (e[0] unused)
static bool tql2(int n, std::vector<long double> &d, std::vector<long double> &e, std::vector<long double> &z) {
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0L;
  long double f = 0.0L, tst1 = 0.0L;
  for (int l = 0; l < n; l++) {
    const long double h0 = fabsl(d[l]) + fabsl(e[l]);
    if (tst1 < h0) tst1 = h0;
    int m = l;
    while (m < n - 1) { if (tst1 + fabsl(e[m]) == tst1) break; m++; }
    if (m != l) {
      int iter = 0;
      do {
        if (++iter > 200) return false;
        const long double g0 = d[l];
        long double p = (d[l + 1] - g0) / (2.0L * e[l]);
        long double r = hypotl(p, 1.0L);
        d[l] = e[l] / (p + (p < 0 ? -r : r));
        d[l + 1] = e[l] * (p + (p < 0 ? -r : r));
        const long double dl1 = d[l + 1];
        long double h = g0 - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        long double c = 1.0L, c2 = c, c3 = c, s = 0.0L, s2 = 0.0L;
        const long double el1 = e[l + 1];
        for (int i = m - 1; i >= l; i--) {
          c3 = c2; c2 = c; s2 = s;
          const long double g = c * e[i];
          h = c * p;
          r = hypotl(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r; c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; k++) {
            h = z[(size_t)k * n + i + 1];
            z[(size_t)k * n + i + 1] = s * z[(size_t)k * n + i] + c * h;
            z[(size_t)k * n + i] = c * z[(size_t)k * n + i] - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (tst1 + fabsl(e[l]) > tst1);
    }
    d[l] += f;
  }
  return true;
}

// S (nodal <- modal), S^-1 and the eigenvalues of the 1-D operator T on the M = P-2 interior nodes of a line
bool fdm_line(int P, std::vector<long double> &S, std::vector<long double> &Sinv, std::vector<long double> &lam) {
  const int n = P - 1, M = P - 2;
  if (M < 1) return false;
  std::vector<long double> x(P), h(M), kd(M), ke(M, 0.0L);
  for (int i = 0; i < P; i++) x[i] = cosl(PI_L * i / n);
  for (int q = 0; q < M; q++) {
    const int i = q + 1;
    const long double idxM = 1.0L / (x[i] - x[i - 1]), idxP = 1.0L / (x[i + 1] - x[i]);
    h[q] = -0.5L * (x[i + 1] - x[i - 1]);              // -(xP - xM) > 0
    kd[q] = -(idxP + idxM);                            // K = -(reference's bracket): positive diagonal
    if (q > 0) ke[q] = idxM;                           // K[q][q-1] = idxM < 0
  }
  // A = H^-1/2 K H^-1/2: symmetric tridiagonal, diagonal d, sub-diagonal e[q] = A[q][q-1]
  std::vector<long double> d(M), e(M, 0.0L);
  for (int q = 0; q < M; q++) { d[q] = kd[q] / h[q]; if (q > 0) e[q] = ke[q] / sqrtl(h[q] * h[q - 1]); }
  // The nodes are symmetric about 0, so A is centro-symmetric and every eigenvector is even or odd under i -> M-1-i.
  // The modes localised at the two ends of the line come in even / odd pairs whose eigenvalues agree to far below
  // rounding, so a solver for the whole matrix returns arbitrary mixtures of each pair.  The two parity classes are
  // therefore diagonalised SEPARATELY: restricted to even (odd) vectors A is again tridiagonal, of half the size, with
  // simple well-separated eigenvalues.  Modes are laid out by parity -- position p < ceil(M/2): the p-th even mode,
  // position M-1-q: the q-th odd mode, both by ascending eigenvalue -- which is the layout the raw modes of the sweep
  // kernels read and write (sweep.h: SweepParams::raw).
  const int m = M - 1, He = (M + 1) / 2, Ho = M / 2;
  const bool has_mid = (M & 1) != 0;
  const long double r2 = sqrtl(2.0L);
  std::vector<long double> W((size_t)M * M, 0.0L);       // W[i][position]
  lam.assign(M, 0.0L);
  for (int parity = 0; parity < 2; parity++) {           // 0: even, 1: odd
    const int n = parity == 0 ? He : Ho;
    if (n == 0) continue;
    std::vector<long double> dd(n), ee(n, 0.0L), Z((size_t)n * n, 0.0L);
    for (int i = 0; i < n; i++) { dd[i] = d[i]; if (i > 0) ee[i] = e[i]; Z[(size_t)i * n + i] = 1.0L; }
    if (!has_mid) dd[n - 1] += (parity == 0 ? e[n] : -e[n]);        // the two middle points couple to each other
    else if (parity == 0 && n > 1) ee[n - 1] = r2 * e[n - 1];       // the middle point is its own mirror
    if (!tql2(n, dd, ee, Z)) return false;
    std::vector<int> ord(n);
    for (int j = 0; j < n; j++) ord[j] = j;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return dd[a] < dd[b]; });
    for (int q = 0; q < n; q++) {
      const int j = ord[q], pos = parity == 0 ? q : m - q;
      lam[pos] = dd[j];
      for (int i = 0; i < n; i++) {
        const long double v = Z[(size_t)i * n + j];
        if (has_mid && parity == 0 && i == n - 1) W[(size_t)i * M + pos] = v;
        else { W[(size_t)i * M + pos] = v / r2; W[(size_t)(m - i) * M + pos] = (parity == 0 ? v : -v) / r2; }
      }
    }
  }
  S.assign((size_t)M * M, 0.0L); Sinv.assign((size_t)M * M, 0.0L);
  for (int i = 0; i < M; i++)
    for (int pos = 0; pos < M; pos++) {
      S[(size_t)i * M + pos] = W[(size_t)i * M + pos] / sqrtl(h[i]);
      Sinv[(size_t)pos * M + i] = W[(size_t)i * M + pos] * sqrtl(h[i]);
    }
  return true;
}

// centro-symmetric (part = 1) or centro-antisymmetric (part = 0) part of a dense M x M matrix
void centro_part(int M, const std::vector<long double> &A, int part, std::vector<long double> &out) {
  out.resize((size_t)M * M);
  const int m = M - 1;
  for (int i = 0; i < M; i++)
    for (int j = 0; j < M; j++) {
      const long double a = A[(size_t)i * M + j], b = A[(size_t)(m - i) * M + (m - j)];
      out[(size_t)i * M + j] = part ? 0.5L * (a + b) : 0.5L * (a - b);
    }
}

void diffmat_destroy(DiffMat *m) {
  if (m->fragE) (void)hipFree(m->fragE);
  if (m->fragO) (void)hipFree(m->fragO);
  if (m->longDT) (void)hipFree(m->longDT);
  if (m->longD) (void)hipFree(m->longD);
  m->fragE = m->fragO = m->fragE2 = m->fragO2 = m->longDT = m->longD = nullptr;
}

}  // namespace chebhip
