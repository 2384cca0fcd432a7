/*
 * cheb_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See cheb_oracle.h for scope, pinning and the FFTW definitions restated here.
 * Build: make -C oracle   (gcc -O2 -fopenmp, no external libraries)
 */
#include "cheb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846 /* chebyshev.h:10 */
static const long double PI_L = 3.14159265358979323846264338327950288L;

/* ------------------------------------------------------------------------- */
/* Transform plan: tables shared read-only by all lines of one sweep.         */
/* N = number of intervals; both REDFT00 on N+1 points and RODFT00 on N-1     */
/* points are real-symmetric DFTs of logical length 2N (FFTW manual).         */
/* ------------------------------------------------------------------------- */
typedef struct { double re, im; } cpx;

typedef struct {
  int N;
  int use_fft;       /* fast path: complex FFT of length N, else table summation */
  int nf, fac[40];   /* radix, remaining length pairs (smallest prime first)       */
  cpx *tw;           /* e^{-2 pi i t/N}, t < N                                    */
  cpx *half;         /* e^{-i pi k/N},  k <= N                                    */
  double *ctab, *stab;        /* cos/sin(pi t/N), t < 2N (double, table-summation fallback) */
  long double *ctabl, *stabl; /* cos/sin(pi t/N), t < 2N (long double, ORC_DIRECT)          */
} xplan;

static void xplan_free(xplan *p) {
  if (!p) return;
  free(p->tw); free(p->half); free(p->ctab); free(p->ctabl); free(p->stab); free(p->stabl); free(p);
}

static xplan *xplan_make(int N, int mode) {
  xplan *p = (xplan *)calloc(1, sizeof(xplan));
  p->N = N;
  if (mode == ORC_DIRECT) {
    p->ctabl = (long double *)malloc(sizeof(long double) * 2 * (size_t)N);
    p->stabl = (long double *)malloc(sizeof(long double) * 2 * (size_t)N);
    for (int t = 0; t < 2 * N; t++) { p->ctabl[t] = cosl(PI_L * t / N); p->stabl[t] = sinl(PI_L * t / N); }
    return p;
  }
  /* factorise N */
  int n = N, maxf = 1;
  for (int f = 2; n > 1;) {
    if (n % f == 0) { p->fac[2 * p->nf] = f; n /= f; p->fac[2 * p->nf + 1] = n; p->nf++; if (f > maxf) maxf = f; }
    else { f++; if ((long)f * f > n) f = n; }
  }
  p->use_fft = (N >= 8 && maxf <= 32);
  if (p->use_fft) {
    p->tw = (cpx *)malloc(sizeof(cpx) * (size_t)N);
    p->half = (cpx *)malloc(sizeof(cpx) * (size_t)(N + 1));
    for (int t = 0; t < N; t++) {
      long double a = -2.0L * PI_L * t / N;
      p->tw[t].re = (double)cosl(a); p->tw[t].im = (double)sinl(a);
    }
    for (int k = 0; k <= N; k++) {
      long double a = -PI_L * k / N;
      p->half[k].re = (double)cosl(a); p->half[k].im = (double)sinl(a);
    }
  } else {
    p->ctab = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    p->stab = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    for (int t = 0; t < 2 * N; t++) { p->ctab[t] = (double)cosl(PI_L * t / N); p->stab[t] = (double)sinl(PI_L * t / N); }
  }
  return p;
}

/* Mixed-radix decimation-in-time complex FFT, generic butterflies. */
static void fft_work(const xplan *p, cpx *out, const cpx *in, long fstride, const int *fac) {
  const int r = fac[0], m = fac[1], N = p->N;
  cpx scratch[32]; /* radices are <= 32 when use_fft is set */
  if (m == 1) {
    for (int q = 0; q < r; q++) out[q] = in[q * fstride];
  } else {
    for (int q = 0; q < r; q++) fft_work(p, out + (long)q * m, in + q * fstride, fstride * r, fac + 2);
  }
  for (int u = 0; u < m; u++) {
    for (int q = 0; q < r; q++) scratch[q] = out[u + (long)q * m];
    for (int q1 = 0; q1 < r; q1++) {
      const long k = u + (long)q1 * m;
      double sr = scratch[0].re, si = scratch[0].im;
      long tix = 0;
      const long step = (fstride * k) % N;
      for (int q = 1; q < r; q++) {
        tix += step; if (tix >= N) tix -= N;
        const cpx w = p->tw[tix];
        sr += scratch[q].re * w.re - scratch[q].im * w.im;
        si += scratch[q].re * w.im + scratch[q].im * w.re;
      }
      out[k].re = sr; out[k].im = si;
    }
  }
}

/* Real DFT of length 2N of xt (real), returned as X_k, k = 0..N, via one
 * complex FFT of length N.  buf must hold 2N + 64 cpx. */
static void rdft2N(const xplan *p, const double *xt, cpx *X, cpx *buf) {
  const int N = p->N;
  cpx *z = buf, *Z = buf + N;
  for (int m = 0; m < N; m++) { z[m].re = xt[2 * m]; z[m].im = xt[2 * m + 1]; }
  fft_work(p, Z, z, 1, p->fac);
  for (int k = 0; k <= N; k++) {
    const cpx a = Z[k % N], b = Z[(N - k) % N];
    const double evr = 0.5 * (a.re + b.re), evi = 0.5 * (a.im - b.im);
    const double dr = a.re - b.re, di = a.im + b.im; /* a - conj(b) */
    const double odr = 0.5 * di, odi = -0.5 * dr;    /* (a-conj b)/(2i) */
    const cpx w = p->half[k];
    X[k].re = evr + w.re * odr - w.im * odi;
    X[k].im = evi + w.re * odi + w.im * odr;
  }
}

typedef struct { double *xt; cpx *X; cpx *buf; } xscratch;
static void xscratch_init(xscratch *s, int N) {
  s->xt = (double *)malloc(sizeof(double) * 2 * (size_t)N);
  s->X = (cpx *)malloc(sizeof(cpx) * (size_t)(N + 1));
  s->buf = (cpx *)malloc(sizeof(cpx) * (2 * (size_t)N + 64));
}
static void xscratch_free(xscratch *s) { free(s->xt); free(s->X); free(s->buf); }

/* FFTW manual REDFT00 on n = N+1 points. */
static void redft00_line(const xplan *p, xscratch *s, const double *in, long is, double *out, long os) {
  const int N = p->N;
  if (p->ctabl) {
    for (int k = 0; k <= N; k++) {
      long double acc = 0.0L;
      for (int j = 1; j < N; j++) acc += (long double)in[j * is] * p->ctabl[((long)j * k) % (2 * N)];
      acc = (long double)in[0] + ((k & 1) ? -1.0L : 1.0L) * (long double)in[(long)N * is] + 2.0L * acc;
      out[k * os] = (double)acc;
    }
  } else if (p->use_fft) {
    double *xt = s->xt;
    for (int j = 0; j <= N; j++) xt[j] = in[j * is];
    for (int j = 1; j < N; j++) xt[2 * N - j] = in[j * is];
    rdft2N(p, xt, s->X, s->buf);
    for (int k = 0; k <= N; k++) out[k * os] = s->X[k].re;
  } else {
    for (int k = 0; k <= N; k++) {
      double acc = 0.0;
      for (int j = 1; j < N; j++) acc += in[j * is] * p->ctab[((long)j * k) % (2 * N)];
      out[k * os] = in[0] + ((k & 1) ? -1.0 : 1.0) * in[(long)N * is] + 2.0 * acc;
    }
  }
}

/* FFTW manual RODFT00 on n = N-1 points. */
static void rodft00_line(const xplan *p, xscratch *s, const double *in, long is, double *out, long os) {
  const int N = p->N, n = N - 1;
  if (n <= 0) return;
  if (p->ctabl) {
    for (int k = 0; k < n; k++) {
      long double acc = 0.0L;
      for (int j = 0; j < n; j++) acc += (long double)in[j * is] * p->stabl[((long)(j + 1) * (k + 1)) % (2 * N)];
      out[k * os] = (double)(2.0L * acc);
    }
  } else if (p->use_fft) {
    double *xt = s->xt;
    xt[0] = 0.0; xt[N] = 0.0;
    for (int j = 0; j < n; j++) { xt[j + 1] = in[j * is]; xt[2 * N - 1 - j] = -in[j * is]; }
    rdft2N(p, xt, s->X, s->buf);
    for (int k = 1; k <= n; k++) out[(k - 1) * os] = -s->X[k].im;
  } else {
    for (int k = 0; k < n; k++) {
      double acc = 0.0;
      for (int j = 0; j < n; j++) acc += in[j * is] * p->stab[((long)(j + 1) * (k + 1)) % (2 * N)];
      out[k * os] = 2.0 * acc;
    }
  }
}

int orc_redft00(int n, const double *in, long is, double *out, long os, int mode) {
  if (n < 2) return 1;
  xplan *p = xplan_make(n - 1, mode);
  xscratch s; xscratch_init(&s, n - 1);
  redft00_line(p, &s, in, is, out, os);
  xscratch_free(&s); xplan_free(p);
  return 0;
}

int orc_rodft00(int n, const double *in, long is, double *out, long os, int mode) {
  if (n < 1) return 1;
  xplan *p = xplan_make(n + 1, mode);
  xscratch s; xscratch_init(&s, n + 1);
  rodft00_line(p, &s, in, is, out, os);
  xscratch_free(&s); xplan_free(p);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* chebyshev.c:89-138 argument checks + stride convention (:107-120).         */
/* ------------------------------------------------------------------------- */
static int cheb_geom(int rank, int tr, const int *dims, long *ntot, int *P, long *inner, long *outer) {
  if (rank < 1 || !(0 <= tr && tr < rank)) return 2;      /* chebyshev.c:106 */
  long tot = 1, in = 1, out = 1;
  for (int r = 0; r < rank; r++) {
    if (dims[r] < 1) return 3;
    tot *= dims[r];
    if (r > tr) in *= dims[r];
    if (r < tr) out *= dims[r];
  }
  if (tot < 2) return 1;                                  /* chebyshev.c:98  */
  if (dims[tr] < 2) return 1;
  *ntot = tot; *P = dims[tr]; *inner = in; *outer = out;
  return 0;
}

/* ---- genuine FFTW, if the box has it: libfftw3.so.3 is looked up at run time and used exactly as
 * chebyshev.c does -- fftw_plan_guru_r2r with FFTW_ESTIMATE, batched strided REDFT00 x -> work (PRESERVE_INPUT,
 * chebyshev.c:127) and RODFT00 on n-1 points work+os -> y+is (DESTROY_INPUT, :128-129), fftw_execute_r2r (:157,181).
 * mode ORC_FFTW = 2.  Not available in the build image (no FFTW): this path is exercised only where the library
 * exists; everything else about the call (passes 2 and 4) is the restatement below. */
#include <dlfcn.h>
typedef struct { int n, is, os; } orc_fftw_iodim;
static struct {
  int tried, ok;
  void *(*plan_guru_r2r)(int, const orc_fftw_iodim *, int, const orc_fftw_iodim *, double *, double *, const int *, unsigned);
  void (*execute_r2r)(void *, double *, double *);
  void (*destroy_plan)(void *);
} g_fftw;
int orc_fftw_available(void) {
  if (!g_fftw.tried) {
    g_fftw.tried = 1;
    void *L = dlopen("libfftw3.so.3", RTLD_NOW | RTLD_LOCAL);
    if (!L) L = dlopen("libfftw3.so", RTLD_NOW | RTLD_LOCAL);
    if (L) {
      *(void **)&g_fftw.plan_guru_r2r = dlsym(L, "fftw_plan_guru_r2r");
      *(void **)&g_fftw.execute_r2r = dlsym(L, "fftw_execute_r2r");
      *(void **)&g_fftw.destroy_plan = dlsym(L, "fftw_destroy_plan");
      g_fftw.ok = g_fftw.plan_guru_r2r && g_fftw.execute_r2r && g_fftw.destroy_plan;
    }
  }
  return g_fftw.ok;
}

int orc_cheb_mult(int rank, int tr, const int *dims, const double *x, double *y, int mode, int nthreads) {
  long ntot, inner, outer; int P;
  int err = cheb_geom(rank, tr, dims, &ntot, &P, &inner, &outer);
  if (err) return err;
  const int n = P - 1;                                    /* chebyshev.c:154 */
  const long ts = inner;                                  /* tdim.is         */
  const long nlines = outer * inner;
  double *work = (double *)malloc(sizeof(double) * (size_t)ntot); /* chebyshev.c:102 */
  void *fp1 = NULL, *fp2 = NULL;
  if (mode == 2) {                                        /* the reference's two plans, chebyshev.c:124-129 */
    if (!orc_fftw_available() || P < 3) { free(work); return 7; }
    orc_fftw_iodim tdim = { P, (int)ts, (int)ts };
    orc_fftw_iodim batch[2] = { { (int)outer, (int)(P * inner), (int)(P * inner) }, { (int)inner, 1, 1 } };
    const int redft00 = 3, rodft00 = 7;                   /* fftw3.h: FFTW_REDFT00, FFTW_RODFT00 */
    const unsigned estimate = 1u << 6, preserve = 1u << 4, destroy = 1u << 0;
    fp1 = g_fftw.plan_guru_r2r(1, &tdim, 2, batch, (double *)x, work, &redft00, estimate | preserve);
    tdim.n = P - 2;
    fp2 = g_fftw.plan_guru_r2r(1, &tdim, 2, batch, work + ts, y + ts, &rodft00, estimate | destroy);
    if (!fp1 || !fp2) { if (fp1) g_fftw.destroy_plan(fp1); if (fp2) g_fftw.destroy_plan(fp2); free(work); return 7; }
    nthreads = 1;                                         /* the reference is serial */
  }
  xplan *p = xplan_make(n, mode == 2 ? 1 : mode);
  if (nthreads < 1) nthreads = 1;
  const double N = (double)n;
  const double pin = ORC_PI / N;                          /* chebyshev.c:183 */
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    xscratch s; xscratch_init(&s, n);
    /* pass 1: forward REDFT00, x -> work (chebyshev.c:157) */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long l = 0; l < nlines; l++) {
      if (fp1) continue;
      const long off = (l / inner) * P * inner + (l % inner);
      redft00_line(p, &s, x + off, ts, work + off, ts);
    }
#ifdef _OPENMP
#pragma omp single
#endif
    { if (fp1) g_fftw.execute_r2r(fp1, (double *)x, work); }   /* chebyshev.c:157 */
    /* pass 2: coefficient scaling and endpoint sums (chebyshev.c:162-179) */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long l = 0; l < nlines; l++) {
      const long offset = (l / inner) * P * inner + (l % inner);
      const long ix0 = offset, ixn = offset + n * ts;
      y[ix0] = 0.0; y[ixn] = 0.0;
      double sgn = 1.0;
      for (int i = 1; i < n; i++) {
        const long ix = offset + i * ts;
        const double I = (double)i;
        work[ix] *= I;
        y[ix0] += I * work[ix];
        y[ixn] += sgn * I * work[ix];
        sgn = -sgn;
      }
      y[ix0] = 0.5 * work[ixn] * N + y[ix0] / n;
      y[ixn] = y[ixn] / N + 0.5 * sgn * N * work[ixn];
    }
    /* pass 3: backward RODFT00 on n-1 points, work+os -> y+is (chebyshev.c:181) */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long l = 0; l < nlines; l++) {
      if (fp2) continue;
      const long off = (l / inner) * P * inner + (l % inner);
      rodft00_line(p, &s, work + off + ts, ts, y + off + ts, ts);
    }
#ifdef _OPENMP
#pragma omp single
#endif
    { if (fp2) g_fftw.execute_r2r(fp2, work + ts, y + ts); }   /* chebyshev.c:181 */
    /* pass 4: metric scaling (chebyshev.c:186-193) */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long l = 0; l < nlines; l++) {
      const long offset = (l / inner) * P * inner + (l % inner);
      for (int i = 1; i < n; i++) {
        const long ix = offset + i * ts;
        const double I = (double)i;
        const double c = cos(I * pin);
        y[ix] /= 2 * n * sqrt(1.0 - c * c);
      }
    }
    xscratch_free(&s);
  }
  xplan_free(p);
  if (fp1) g_fftw.destroy_plan(fp1);
  if (fp2) g_fftw.destroy_plan(fp2);
  free(work);
  return 0;
}

int orc_cheb_mult_truth(int rank, int tr, const int *dims, const double *x, double *y) {
  long ntot, inner, outer; int P;
  int err = cheb_geom(rank, tr, dims, &ntot, &P, &inner, &outer);
  if (err) return err;
  const int n = P - 1;
  const long ts = inner, nlines = outer * inner;
  long double *Y = (long double *)malloc(sizeof(long double) * (size_t)(n + 1));
  long double *ct = (long double *)malloc(sizeof(long double) * 2 * (size_t)n);
  long double *st = (long double *)malloc(sizeof(long double) * 2 * (size_t)n);
  for (int t = 0; t < 2 * n; t++) { ct[t] = cosl(PI_L * t / n); st[t] = sinl(PI_L * t / n); }
  for (long l = 0; l < nlines; l++) {
    const long off = (l / inner) * P * inner + (l % inner);
    const double *u = x + off; double *v = y + off;
    for (int k = 0; k <= n; k++) {
      long double acc = 0.0L;
      for (int j = 1; j < n; j++) acc += (long double)u[j * ts] * ct[((long)j * k) % (2 * n)];
      Y[k] = (long double)u[0] + ((k & 1) ? -1.0L : 1.0L) * (long double)u[(long)n * ts] + 2.0L * acc;
    }
    long double y0 = 0.0L, yn = 0.0L, sgn = 1.0L;
    for (int k = 1; k < n; k++) {
      Y[k] *= k;
      y0 += k * Y[k]; yn += sgn * k * Y[k]; sgn = -sgn;
    }
    v[0] = (double)(0.5L * Y[n] * n + y0 / n);
    v[(long)n * ts] = (double)(yn / n + 0.5L * sgn * n * Y[n]);
    for (int j = 1; j < n; j++) {
      long double acc = 0.0L;
      for (int k = 1; k < n; k++) acc += Y[k] * st[((long)j * k) % (2 * n)];
      v[j * ts] = (double)(2.0L * acc / (2.0L * n * st[j]));
    }
  }
  free(Y); free(ct); free(st);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* elliptic.C:372-434 SetupBC: local / global(interior) / dirichlet(boundary) */
/* index sets in BlockIt (row-major) order.                                   */
/* ------------------------------------------------------------------------- */
long orc_local_size(int d, const int *dims) { long n = 1; for (int i = 0; i < d; i++) n *= dims[i]; return n; }
long orc_global_size(int d, const int *dims) { long n = 1; for (int i = 0; i < d; i++) n *= (dims[i] > 2 ? dims[i] - 2 : 0); return n; }
long orc_dirichlet_size(int d, const int *dims) { return orc_local_size(d, dims) - orc_global_size(d, dims); }

/* ixL[l] = global index or -1 on the boundary (elliptic.C:399-407). */
static int *build_ixL(int d, const int *dims, long N) {
  int *ixL = (int *)malloc(sizeof(int) * (size_t)N);
  int ind[16] = {0};
  long g = 0;
  for (long l = 0; l < N; l++) {
    int bdy = 0;
    for (int j = 0; j < d; j++) if (ind[j] == 0 || ind[j] == dims[j] - 1) bdy = 1;
    ixL[l] = bdy ? -1 : (int)g++;
    for (int j = d - 1; j >= 0; j--) { if (++ind[j] < dims[j]) break; ind[j] = 0; }
  }
  return ixL;
}

static void par_for_pointwise(long N, int nthreads, void (*body)(long, long, void *), void *ctx) {
  /* tiny helper so the pointwise passes honour nthreads too */
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int t = 0; t < nthreads; t++) {
    long lo = N * t / nthreads, hi = N * (t + 1) / nthreads;
    body(lo, hi, ctx);
  }
}

typedef struct { int d; long N; const double *eta, *deta, *u, *gradu0; double *w; } flux_ctx;
static void flux_body(long lo, long hi, void *vc) {
  flux_ctx *c = (flux_ctx *)vc;
  for (long i = lo; i < hi; i++)
    for (int k = 0; k < c->d; k++) {               /* elliptic.C:319-323 */
      double *uk = c->w + (long)(1 + k) * c->N;
      uk[i] = c->eta[i] * uk[i] + c->deta[i] * c->u[i] * c->gradu0[(long)k * c->N + i];
    }
}

typedef struct { long N; double *a; const double *b; } axpy_ctx;
static void axpym1_body(long lo, long hi, void *vc) {
  axpy_ctx *c = (axpy_ctx *)vc;
  for (long i = lo; i < hi; i++) c->a[i] += -1.0 * c->b[i]; /* VecAXPY(w0,-1,t) elliptic.C:333 */
}

static int cheb_dims_dir(int d, const int *dims, int k, const double *x, double *y, int mode, int nthreads) {
  return orc_cheb_mult(d, k, dims, x, y, mode, nthreads);  /* MatCreateCheb(comm,d,i,dim,...) elliptic.C:269-271 */
}

int orc_elliptic_mult(int d, const int *dims, const double *eta, const double *deta,
                      const double *gradu0, const double *U, double *V, int mode, int nthreads) {
  if (d < 1 || d > 10) return 4;
  const long N = orc_local_size(d, dims);
  int *ixL = build_ixL(d, dims, N);
  double *w = (double *)malloc(sizeof(double) * (size_t)N * (2 + d));  /* c->w[2+d] elliptic.C:260,263 */
  double *w0 = w;
  int err = 0;
  /* scatter GL then DL with dirichlet0 == 0 (elliptic.C:305-308) */
  for (long l = 0; l < N; l++) w0[l] = ixL[l] >= 0 ? U[ixL[l]] : 0.0;
  for (int k = 0; k < d && !err; k++) err = cheb_dims_dir(d, dims, k, w0, w + (long)(1 + k) * N, mode, nthreads);
  flux_ctx fc = { d, N, eta, deta, w0, gradu0, w };
  if (!err) par_for_pointwise(N, nthreads, flux_body, &fc);
  memset(w0, 0, sizeof(double) * (size_t)N);             /* VecZeroEntries elliptic.C:330 */
  double *t = w + (long)(1 + d) * N;
  for (int k = 0; k < d && !err; k++) {                  /* elliptic.C:331-334 */
    err = cheb_dims_dir(d, dims, k, w + (long)(1 + k) * N, t, mode, nthreads);
    axpy_ctx ac = { N, w0, t };
    par_for_pointwise(N, nthreads, axpym1_body, &ac);
  }
  for (long l = 0; l < N; l++) if (ixL[l] >= 0) V[ixL[l]] = w0[l];  /* scatter LG :336 */
  free(w); free(ixL);
  return err;
}

/* Timing protocol of BASELINE.md section 3 for the CPU baseline: `warm` untimed applies, then `reps` timed ones
 * (seconds[i] = wall time of timed apply i).  The reference allocates its work vectors once, at MatCreateCheb /
 * MatCreate_Elliptic time (chebyshev.c:102, elliptic.C:260-263); here they are allocated per call, so the allocator is
 * told to keep freed blocks (no mmap / trim): after the warm-up applies every buffer is recycled, already touched
 * memory and allocation and first-touch cost stay outside the timed applies, as plan construction does. */
#include <malloc.h>
#include <time.h>
int orc_elliptic_mult_timed(int d, const int *dims, const double *U, double *V, int mode, int nthreads,
                            int warm, int reps, double *seconds) {
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, -1);
  const long N = orc_local_size(d, dims);
  double *eta = (double *)malloc(sizeof(double) * (size_t)N), *deta = (double *)calloc((size_t)N, sizeof(double));
  double *g0 = (double *)calloc((size_t)N * d, sizeof(double));
  for (long i = 0; i < N; i++) eta[i] = 1.0;               /* gamma = 0: eta == 1, deta == 0 (elliptic.C:265-266) */
  int err = 0;
  for (int it = 0; it < warm + reps && !err; it++) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    err = orc_elliptic_mult(d, dims, eta, deta, g0, U, V, mode, nthreads);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (it >= warm) seconds[it - warm] = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  }
  free(eta); free(deta); free(g0);
  return err;
}

int orc_elliptic_function(int d, const int *dims, double gamma, double exponent,
                          const double *dirichlet, const double *U, const double *b,
                          double *rhs, double *eta_o, double *deta_o, double *gradu_o,
                          int mode, int nthreads) {
  if (d < 1 || d > 10) return 4;
  const long N = orc_local_size(d, dims);
  int *ixL = build_ixL(d, dims, N);
  double *w = (double *)malloc(sizeof(double) * (size_t)N * (2 + d));
  double *gradu = (double *)malloc(sizeof(double) * (size_t)N * d);
  double *eta = (double *)malloc(sizeof(double) * (size_t)N);
  double *deta = (double *)malloc(sizeof(double) * (size_t)N);
  double *w0 = w;
  int err = 0;
  long dd = 0;
  for (long l = 0; l < N; l++) {                          /* elliptic.C:490-493 */
    if (ixL[l] >= 0) w0[l] = U[ixL[l]];
    else { w0[l] = dirichlet ? dirichlet[dd] : 0.0; dd++; }
  }
  for (int k = 0; k < d && !err; k++) err = cheb_dims_dir(d, dims, k, w0, gradu + (long)k * N, mode, nthreads); /* :497-499 */
  for (long i = 0; i < N; i++) {                          /* elliptic.C:507-513 */
    eta[i] = 1.0 + gamma * pow(w0[i], exponent);
    deta[i] = exponent * gamma * pow(w0[i], exponent - 1.0);
    for (int k = 0; k < d; k++) w[(long)(1 + k) * N + i] = eta[i] * gradu[(long)k * N + i];
  }
  memset(w0, 0, sizeof(double) * (size_t)N);
  double *t = w + (long)(1 + d) * N;
  for (int k = 0; k < d && !err; k++) {                  /* elliptic.C:521-524 */
    err = cheb_dims_dir(d, dims, k, w + (long)(1 + k) * N, t, mode, nthreads);
    for (long i = 0; i < N; i++) w0[i] += -1.0 * t[i];
  }
  for (long l = 0; l < N; l++) if (ixL[l] >= 0) rhs[ixL[l]] = w0[l];
  const long G = orc_global_size(d, dims);
  if (b) for (long g = 0; g < G; g++) rhs[g] += -1.0 * b[g];  /* VecAXPY(rhs,-1,b) :530 */
  if (eta_o) memcpy(eta_o, eta, sizeof(double) * (size_t)N);
  if (deta_o) memcpy(deta_o, deta, sizeof(double) * (size_t)N);
  if (gradu_o) memcpy(gradu_o, gradu, sizeof(double) * (size_t)N * d);
  free(w); free(gradu); free(eta); free(deta); free(ixL);
  return err;
}

int orc_elliptic_exact(int d, const int *dims, int exact, double gamma, double exponent,
                       double cos_scale, double *u, double *u2, double *dirichlet) {
  if (d < 1 || d > 10) return 4;
  const long N = orc_local_size(d, dims);
  int *ixL = build_ixL(d, dims, N);
  int ind[16] = {0};
  double s = 0.5;                                         /* elliptic.C:605-610 */
  if (exact == 0 || exact == 3) s *= cos_scale;
  long dd = 0;
  for (long l = 0; l < N; l++) {
    double x[16];
    for (int j = 0; j < d; j++) x[j] = cos(ind[j] * ORC_PI / (dims[j] - 1)); /* elliptic.C:279 */
    double v = 0.0, w = 0.0, z;
    switch (exact) {
      case 0: {                                           /* elliptic.C:620-632 */
        v = 1.0; w = 0.0;
        for (int j = 0; j < d; j++) v *= cos(s * ORC_PI * x[j]);
        const double eta = 1.0 + gamma * pow(v, exponent);
        const double deta = (fabs(exponent) < 1e-10) ? 0.0 : gamma * exponent * pow(v, exponent - 1.0);
        for (int j = 0; j < d; j++) {
          double dv = 1.0;
          for (int k = 0; k < d; k++) dv *= (k == j) ? -s * ORC_PI * sin(s * ORC_PI * x[k]) : cos(s * ORC_PI * x[k]);
          const double d2v = -(s * ORC_PI) * (s * ORC_PI) * v;
          w += deta * dv * dv + eta * d2v;
        }
        w = -w;
      } break;
      case 1:                                             /* elliptic.C:633-643 */
        v = 1.0; w = 0.0;
        for (int j = 0; j < d; j++) {
          v *= (1 - x[j]) * (1 + x[j]);
          z = 1.0;
          for (int k = 0; k < d; k++) if (k != j) z *= 2.0 * (1 - x[k]) * (1 + x[k]);
          w += z;
        }
        break;
      case 2:                                             /* elliptic.C:644-655 */
        v = 1.0; w = 0.0;
        for (int j = 0; j < d; j++) {
          v *= pow(x[j], 4 + j);
          z = 1.0;
          for (int k = 0; k < d; k++) {
            if (k == j) z *= (4 + k) * (3 + k) * pow(x[k], 2 + k);
            else z *= pow(x[k], 4 + k);
          }
          w -= z;
        }
        break;
      default:
        free(ixL); return 5;
    }
    if (ixL[l] >= 0) { if (u) u[ixL[l]] = v; if (u2) u2[ixL[l]] = w; }
    else { if (dirichlet) dirichlet[dd] = v; dd++; }
    for (int j = d - 1; j >= 0; j--) { if (++ind[j] < dims[j]) break; ind[j] = 0; }
  }
  free(ixL);
  return 0;
}

/* ========================================================================= */
/* Stokes restatement (stokes.C), -boundary 0.                                */
/* ========================================================================= */
static void stokes_counts(int d, const int *dims, long *N, long *I) {
  *N = orc_local_size(d, dims); *I = orc_global_size(d, dims);
}

/* util.C:129-144 polyInterp, verbatim arithmetic (Neville table in w[i*4+..]). */
static void poly_interp(int n, const double *x, double *w, double x0, double x1, double *f0, double *f1) {
  int o = 0, e;
  for (int di = 1; di < n; di++) {
    o = di % 2; e = (o + 1) % 2;
    for (int i = 0; i < n - di; i++) {
      w[i * 4 + 2 * o]     = ((x0 - x[i + di]) * w[i * 4 + 2 * e]     + (x[i] - x0) * w[(i + 1) * 4 + 2 * e])     / (x[i] - x[i + di]);
      w[i * 4 + 2 * o + 1] = ((x1 - x[i + di]) * w[i * 4 + 2 * e + 1] + (x[i] - x1) * w[(i + 1) * 4 + 2 * e + 1]) / (x[i] - x[i + di]);
    }
  }
  *f0 = w[2 * o];
  *f1 = w[2 * o + 1];
}

int orc_stokes_pressure_reduce(int d, const int *dims, double *pres) {
  if (d < 2 || d > 3) return 6;                                   /* stokes.C:1036 */
  const int m = dims[0], n = dims[1], p = (d == 2) ? 1 : dims[2];
  int mnp = m > n ? m : n; if (p > mnp) mnp = p;
  double *work = (double *)malloc(sizeof(double) * 4 * (size_t)mnp), *x = (double *)malloc(sizeof(double) * (size_t)mnp);
  /* coordinates: x_i = cos(i pi/(dim-1)) (stokes.C:296) */
#define CX(i) cos((i) * ORC_PI / (m - 1))
#define CY(j) cos((j) * ORC_PI / (n - 1))
#define CZ(k) cos((k) * ORC_PI / (p - 1))
  for (int i = 1; i < m; i++) {                                   /* stokes.C:1042 */
    if (p > 1) {
      for (int j = 1; j < n; j++) {
        const long iM = ((long)i * n + j) * p + 0, iP = ((long)i * n + j) * p + p - 1;
        for (int k = 1; k < p - 1; k++) { x[k - 1] = CZ(k); work[(k - 1) * 4] = work[(k - 1) * 4 + 1] = pres[iM + k]; }
        if (p - 2 >= 1) poly_interp(p - 2, x, work, CZ(0), CZ(p - 1), &pres[iM], &pres[iP]);
      }
    }
    for (int k = 0; k < p; k++) {
      const long iM = ((long)i * n + 0) * p + k, iP = ((long)i * n + n - 1) * p + k;
      for (int j = 1; j < n - 1; j++) { x[j - 1] = CY(j); work[(j - 1) * 4] = work[(j - 1) * 4 + 1] = pres[iM + (long)j * p]; }
      if (n - 2 >= 1) poly_interp(n - 2, x, work, CY(0), CY(n - 1), &pres[iM], &pres[iP]);
    }
  }
  for (int j = 0; j < n; j++)
    for (int k = 0; k < p; k++) {
      const long iM = ((long)0 * n + j) * p + k, iP = ((long)(m - 1) * n + j) * p + k;
      for (int i = 1; i < m - 1; i++) { x[i - 1] = CX(i); work[(i - 1) * 4] = work[(i - 1) * 4 + 1] = pres[iM + (long)i * n * p]; }
      if (m - 2 >= 1) poly_interp(m - 2, x, work, CX(0), CX(m - 1), &pres[iM], &pres[iP]);
    }
#undef CX
#undef CY
#undef CZ
  free(work); free(x);
  return 0;
}

/* DV[i]: rank d+1, dims = {dim..., d}, tr = i (stokes.C:284-290); DP[i]: rank d scalar. */
static int stokes_DV(int d, const int *dims, int i, const double *x, double *y, int mode, int nt) {
  int cd[17];
  for (int k = 0; k < d; k++) cd[k] = dims[k];
  cd[d] = d;
  return orc_cheb_mult(d + 1, i, cd, x, y, mode, nt);
}

/* velocity local <- global (scatterVL, zero elsewhere), optional dirichlet (scatterDL) */
static void stokes_v_local(int d, long N, const int *ixL, const double *vG, const double *dirichlet, double *xL) {
  long dd = 0;
  for (long l = 0; l < N; l++)
    for (int k = 0; k < d; k++) {
      if (ixL[l] >= 0) xL[l * d + k] = vG[(long)ixL[l] * d + k];
      else { xL[l * d + k] = dirichlet ? dirichlet[dd] : 0.0; dd++; }
    }
}

int orc_stokes_mult_vv(int d, const int *dims, const double *eta, const double *deta, const double *Strain,
                       const double *vG_in, double *vG_out, int mode, int nt) {
  if (d < 1 || d > 3) return 6;
  long N, I; stokes_counts(d, dims, &N, &I);
  int *ixL = build_ixL(d, dims, N);
  const long nd = N * d;
  double *xL = (double *)malloc(sizeof(double) * (size_t)nd * (2 + 2 * d));
  double *yL = xL + nd, *V = yL + nd, *W = V + (long)d * nd;
  int err = 0;
  stokes_v_local(d, N, ixL, vG_in, NULL, xL);                      /* stokes.C:634-637 */
  for (int i = 0; i < d && !err; i++) err = stokes_DV(d, dims, i, xL, V + (long)i * nd, mode, nt);  /* :639 */
  for (long i = 0; i < N; i++) {                                   /* :647-662 */
    double strain[3][3], z = 0.0;
    for (int j = 0; j < d; j++)
      for (int k = 0; k < d; k++) {
        strain[j][k] = 0.5 * (V[(long)j * nd + i * d + k] + V[(long)k * nd + i * d + j]);
        z += strain[j][k] * Strain[(long)j * nd + i * d + k];
      }
    for (int j = 0; j < d; j++)
      for (int k = 0; k < d; k++) {
        const double s = eta[i] * strain[j][k];
        V[(long)j * nd + i * d + k] = s + deta[i] * Strain[(long)j * nd + i * d + k] * z;
      }
  }
  for (int i = 0; i < d && !err; i++) err = stokes_DV(d, dims, i, V + (long)i * nd, W + (long)i * nd, mode, nt);  /* :668 */
  memset(yL, 0, sizeof(double) * (size_t)nd);
  for (int i = 0; i < d; i++) for (long a = 0; a < nd; a++) yL[a] += -1.0 * W[(long)i * nd + a];    /* :670-671 */
  for (long l = 0; l < N; l++) if (ixL[l] >= 0) for (int k = 0; k < d; k++) vG_out[(long)ixL[l] * d + k] = yL[l * d + k];
  free(xL); free(ixL);
  return err;
}

int orc_stokes_divergence(int d, const int *dims, const double *dirichlet, const double *vG, double *pG, int mode, int nt) {
  if (d < 1 || d > 3) return 6;
  long N, I; stokes_counts(d, dims, &N, &I);
  int *ixL = build_ixL(d, dims, N);
  double *xL = (double *)malloc(sizeof(double) * (size_t)N * (d + 3));
  double *p0 = xL + N * d, *p1 = p0 + N, *p2 = p1 + N;
  int err = 0;
  stokes_v_local(d, N, ixL, vG, dirichlet, xL);                    /* stokes.C:575-582 */
  memset(p2, 0, sizeof(double) * (size_t)N);
  for (int i = 0; i < d && !err; i++) {                            /* :584-591 */
    for (long l = 0; l < N; l++) p0[l] = xL[l * d + i];            /* VecStrideGather */
    err = orc_cheb_mult(d, i, dims, p0, p1, mode, nt);
    for (long l = 0; l < N; l++) p2[l] += 1.0 * p1[l];
  }
  for (long l = 0; l < N; l++) if (ixL[l] >= 0) pG[ixL[l]] = p2[l];
  free(xL); free(ixL);
  return err;
}

int orc_stokes_mult_vp(int d, const int *dims, const double *pG, double *vG, int mode, int nt) {
  if (d < 2 || d > 3) return 6;
  long N, I; stokes_counts(d, dims, &N, &I);
  int *ixL = build_ixL(d, dims, N);
  double *p0 = (double *)malloc(sizeof(double) * (size_t)N * (2 + d));
  double *p1 = p0 + N, *vL = p1 + N;
  for (long l = 0; l < N; l++) p0[l] = ixL[l] >= 0 ? pG[ixL[l]] : 0.0;          /* stokes.C:606-608 */
  int err = orc_stokes_pressure_reduce(d, dims, p0);               /* :609 */
  memset(vL, 0, sizeof(double) * (size_t)N * d);
  for (int i = 0; i < d && !err; i++) {                            /* :611-614 */
    err = orc_cheb_mult(d, i, dims, p0, p1, mode, nt);
    for (long l = 0; l < N; l++) vL[l * d + i] = p1[l];            /* VecStrideScatter */
  }
  for (long l = 0; l < N; l++) if (ixL[l] >= 0) for (int k = 0; k < d; k++) vG[(long)ixL[l] * d + k] = vL[l * d + k];
  free(p0); free(ixL);
  return err;
}

int orc_stokes_mult(int d, const int *dims, const double *eta, const double *deta, const double *strain,
                    const double *xG, double *yG, int mode, int nt) {
  long N, I; stokes_counts(d, dims, &N, &I);
  double *vG0 = (double *)malloc(sizeof(double) * (size_t)I * (3 * d + 2));
  double *vG1 = vG0 + I * d, *vG2 = vG1 + I * d, *pG0 = vG2 + I * d, *pG1 = pG0 + I;
  for (long n = 0; n < I; n++) { for (int k = 0; k < d; k++) vG0[n * d + k] = xG[n * (d + 1) + k]; pG0[n] = xG[n * (d + 1) + d]; }  /* scatterGV/GP */
  int err = orc_stokes_mult_vv(d, dims, eta, deta, strain, vG0, vG1, mode, nt);                 /* stokes.C:508 */
  if (!err) err = orc_stokes_divergence(d, dims, NULL, vG0, pG1, mode, nt);                      /* :509 */
  if (!err) err = orc_stokes_mult_vp(d, dims, pG0, vG2, mode, nt);                               /* :512 */
  for (long a = 0; a < I * d; a++) vG1[a] += 1.0 * vG2[a];                                       /* :513 */
  for (long n = 0; n < I; n++) { for (int k = 0; k < d; k++) yG[n * (d + 1) + k] = vG1[n * d + k]; yG[n * (d + 1) + d] = pG1[n]; }
  free(vG0);
  return err;
}

static void rheology_eval(const orc_rheology *rh, double gamma, double *eta, double *deta) {
  if (!rh || rh->kind == 0) { *eta = 1.0; *deta = 0.0; return; }   /* stokes.C:1924 */
  const double n = rh->exponent, p = (1.0 - n) / (2.0 * n);          /* :1933-1934 */
  *eta = rh->hardness * pow(rh->regularization + gamma / rh->gamma0, p);
  if (fabs(n) > 1.0e-5) *deta = rh->hardness * p / rh->gamma0 * pow(rh->regularization + gamma / rh->gamma0, p - 1.0);
  else *deta = 0.0;
}

int orc_stokes_function(int d, const int *dims, const orc_rheology *rh, const double *dirichlet,
                        const double *force, const double *xG, double *yG,
                        double *eta_o, double *deta_o, double *strain_o, int mode, int nt) {
  if (d < 2 || d > 3) return 6;
  long N, I; stokes_counts(d, dims, &N, &I);
  int *ixL = build_ixL(d, dims, N);
  const long nd = N * d;
  double *buf = (double *)malloc(sizeof(double) * ((size_t)nd * (2 + 3 * d) + 2 * (size_t)N + (size_t)I * (3 * d + 2)));
  double *xL = buf, *yL = xL + nd, *V = yL + nd, *W = V + (long)d * nd, *strain = W + (long)d * nd;
  double *eta = strain + (long)d * nd, *deta = eta + N;
  double *vG0 = deta + N, *vG1 = vG0 + I * d, *vG2 = vG1 + I * d, *pG0 = vG2 + I * d, *pG1 = pG0 + I;
  int err = 0;
  for (long n = 0; n < I; n++) { for (int k = 0; k < d; k++) vG0[n * d + k] = xG[n * (d + 1) + k]; pG0[n] = xG[n * (d + 1) + d]; }  /* stokes.C:691-694 */
  stokes_v_local(d, N, ixL, vG0, dirichlet, xL);                   /* :695-699 */
  for (int i = 0; i < d && !err; i++) err = stokes_DV(d, dims, i, xL, strain + (long)i * nd, mode, nt);  /* :701 */
  for (long i = 0; i < N; i++) {                                   /* :710-725 */
    double s[3][3], gamma = 0.0;
    for (int j = 0; j < d; j++)
      for (int k = 0; k < d; k++) {
        s[j][k] = 0.5 * (strain[(long)j * nd + i * d + k] + strain[(long)k * nd + i * d + j]);
        gamma += 0.5 * (s[j][k] * s[j][k]);
      }
    rheology_eval(rh, gamma, &eta[i], &deta[i]);
    for (int j = 0; j < d; j++)
      for (int k = 0; k < d; k++) { V[(long)j * nd + i * d + k] = eta[i] * s[j][k]; strain[(long)j * nd + i * d + k] = s[j][k]; }
  }
  for (int i = 0; i < d && !err; i++) err = stokes_DV(d, dims, i, V + (long)i * nd, W + (long)i * nd, mode, nt);  /* :737 */
  memset(yL, 0, sizeof(double) * (size_t)nd);
  for (int i = 0; i < d; i++) for (long a = 0; a < nd; a++) yL[a] += -1.0 * W[(long)i * nd + a];    /* :739-740 */
  for (long l = 0; l < N; l++) if (ixL[l] >= 0) for (int k = 0; k < d; k++) vG1[(long)ixL[l] * d + k] = yL[l * d + k];  /* :743 */
  if (!err) err = orc_stokes_divergence(d, dims, dirichlet, vG0, pG1, mode, nt);   /* :746, withDirichlet */
  if (!err) err = orc_stokes_mult_vp(d, dims, pG0, vG2, mode, nt);                 /* :747 (overwrites vG0 in the reference) */
  for (long a = 0; a < I * d; a++) vG1[a] += 1.0 * vG2[a];                         /* :750 */
  for (long n = 0; n < I; n++) { for (int k = 0; k < d; k++) yG[n * (d + 1) + k] = vG1[n * d + k]; yG[n * (d + 1) + d] = pG1[n]; }
  if (force) for (long a = 0; a < I * (d + 1); a++) yG[a] += -1.0 * force[a];      /* :756 */
  if (eta_o) memcpy(eta_o, eta, sizeof(double) * (size_t)N);
  if (deta_o) memcpy(deta_o, deta, sizeof(double) * (size_t)N);
  if (strain_o) memcpy(strain_o, strain, sizeof(double) * (size_t)nd * d);
  free(buf); free(ixL);
  return err;
}

int orc_stokes_exact(int d, const int *dims, int exact, double *U, double *U2, double *dirichlet) {
  if (d < 2 || d > 3) return 6;
  long N, I; stokes_counts(d, dims, &N, &I);
  int *ixL = build_ixL(d, dims, N);
  int ind[4] = {0, 0, 0, 0};
  long dd = 0;
  for (long l = 0; l < N; l++) {
    double c[3] = {0, 0, 0}, val[4] = {0, 0, 0, 0}, rhs[4] = {0, 0, 0, 0};
    for (int j = 0; j < d; j++) c[j] = cos(ind[j] * ORC_PI / (dims[j] - 1));   /* stokes.C:296 */
    const double eta = 1.0;
    if (exact == 1 || exact == 2) {                                /* stokes.C:1963-2012 */
      const double u = sin(0.5 * ORC_PI * c[0]) * cos(0.5 * ORC_PI * c[1]);
      const double v = -cos(0.5 * ORC_PI * c[0]) * sin(0.5 * ORC_PI * c[1]);
      /* Exact2 leaves the 3-D pressure slot unset in the reference (value[3] never written, :2000-2001);
       * the restatement pins it to 0. */
      const double p = (exact == 1) ? 0.25 * (cos(ORC_PI * c[0]) + cos(ORC_PI * c[1])) + 10 * (c[0] + c[1]) : 0.0;
      val[0] = u; val[1] = v; if (d == 3) val[2] = 0.0; val[d] = p;
      rhs[0] = (0.5 * ORC_PI) * (0.5 * ORC_PI) * eta * u; rhs[1] = (0.5 * ORC_PI) * (0.5 * ORC_PI) * eta * v;
      if (exact == 1) { rhs[0] += -0.25 * ORC_PI * sin(ORC_PI * c[0]) + 10; rhs[1] += -0.25 * ORC_PI * sin(ORC_PI * c[1]) + 10; }
      if (d == 3) rhs[2] = 0.0;
      rhs[d] = 0.0;
    } else if (exact != 0) { free(ixL); return 5; }
    if (ixL[l] >= 0) {
      for (int k = 0; k <= d; k++) { if (U) U[(long)ixL[l] * (d + 1) + k] = val[k]; if (U2) U2[(long)ixL[l] * (d + 1) + k] = rhs[k]; }
    } else {
      for (int k = 0; k < d; k++) { if (dirichlet) dirichlet[dd] = val[k]; dd++; }
    }
    for (int j = d - 1; j >= 0; j--) { if (++ind[j] < dims[j]) break; ind[j] = 0; }
  }
  free(ixL);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * FormJacobian's preconditioning matrix P (elliptic.C:537-590) and, with gradu == NULL, one velocity
 * component of MatVVPC (StokesPCSetUp0, stokes.C:1181-1226): rows in global (interior) order, 2d+1 entries
 * each -- cols[r*(2d+1)+0] = r (diagonal), then the (-1, +1) neighbours per dimension; a neighbour on the
 * boundary has column -1, which MatSetValues ignores (value kept for inspection).
 * eta, deta: N local values; gradu: d*N (gradu[j][i]) or NULL.
 * ------------------------------------------------------------------------------------------- */
int orc_fd_matrix(int d, const int *dims, const double *eta, const double *deta, const double *gradu,
                  int *cols, double *vals) {
  if (d < 1 || d > 10) return 6;
  long N = 1; for (int j = 0; j < d; j++) N *= dims[j];
  int *ixL = build_ixL(d, dims, N);
  long ls[10]; { long s = 1; for (int j = d - 1; j >= 0; j--) { ls[j] = s; s *= dims[j]; } }
  int ind[10] = {0};
  const int W = 2 * d + 1;
  for (long i = 0; i < N; i++) {                               /* BlockIt order, elliptic.C:561 */
    if (ixL[i] >= 0) {                                         /* :563 */
      const long r = ixL[i];
      int *J = cols + r * W; double *v = vals + r * W;
      J[0] = (int)r; v[0] = 0.0; int k = 1;                    /* :564 */
      for (int j = 0; j < d; j++) {
        const long iM = i - ls[j], iP = i + ls[j];             /* it.shift(j, -1 / +1), :566-567 */
        const double x0 = cos(ind[j] * ORC_PI / (dims[j] - 1)), xMM = cos((ind[j] - 1) * ORC_PI / (dims[j] - 1)),
                     xPP = cos((ind[j] + 1) * ORC_PI / (dims[j] - 1));                                 /* :569 */
        const double xM = 0.5 * (xMM + x0), idxM = 1.0 / (x0 - xMM), xP = 0.5 * (x0 + xPP), idxP = 1.0 / (xPP - x0), idx = 1.0 / (xP - xM);   /* :570 */
        const double eM = 0.5 * (eta[iM] + eta[i]), eP = 0.5 * (eta[iP] + eta[i]);
        double deM = 0, du0M = 0, deP = 0, du0P = 0;
        if (gradu) {                                           /* :571-572 */
          deM = 0.5 * (deta[iM] + deta[i]); du0M = 0.5 * (gradu[(long)j * N + iM] + gradu[(long)j * N + i]);
          deP = 0.5 * (deta[iP] + deta[i]); du0P = 0.5 * (gradu[(long)j * N + iP] + gradu[(long)j * N + i]);
        }
        J[k] = ixL[iM]; v[k] = -idx * (idxM * eM - 0.5 * deM * du0M); k++;                            /* :573 */
        J[k] = ixL[iP]; v[k] = -idx * (idxP * eP + 0.5 * deP * du0P); k++;                            /* :574 */
        v[0] += idx * (idxP * eP + idxM * eM - 0.5 * (deP * du0P - deM * du0M));                       /* :575 */
      }
    }
    for (int j = d - 1; j >= 0; j--) { if (++ind[j] < dims[j]) break; ind[j] = 0; }
  }
  free(ixL);
  return 0;
}

