// Host-side check of the fast-diagonalisation data of the finite-difference preconditioner (csrc/diffmat.cpp: fdm_line):
// prints, per line length P, the residuals of S S^-1 = I and T S = S Lambda (T: the three-point operator of
// elliptic.C:556-579 with eta = 1 on the interior Gauss-Lobatto nodes) and the parity defect of the mode layout.
// Built and run by tests/test_host_fdm.py; needs no GPU.
#include "sweep.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
using namespace chebhip;

int main(int argc, char **argv) {
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int a = 1; a < argc; a++) {
    const int P = atoi(argv[a]), M = P - 2, n = P - 1, He = (M + 1) / 2;
    std::vector<long double> S, Si, lam;
    if (!fdm_line(P, S, Si, lam)) { printf("%d failed\n", P); continue; }
    long double e_inv = 0, e_eig = 0, e_par = 0, smax = 0, lmax = 0;
    for (int i = 0; i < M; i++)
      for (int j = 0; j < M; j++) {
        long double s = 0;
        for (int k = 0; k < M; k++) s += S[(size_t)i * M + k] * Si[(size_t)k * M + j];
        e_inv = fmaxl(e_inv, fabsl(s - (i == j ? 1.0L : 0.0L)));
        smax = fmaxl(smax, fabsl(S[(size_t)i * M + j]));
      }
    std::vector<long double> x(P);
    for (int i = 0; i < P; i++) x[i] = cosl(pi * i / n);
    for (int pos = 0; pos < M; pos++) lmax = fmaxl(lmax, fabsl(lam[pos]));
    for (int pos = 0; pos < M; pos++)
      for (int q = 0; q < M; q++) {
        const int i = q + 1;
        const long double idxM = 1 / (x[i] - x[i - 1]), idxP = 1 / (x[i + 1] - x[i]), idx = -1 / (0.5L * (x[i + 1] - x[i - 1]));
        const long double um = q > 0 ? S[(size_t)(q - 1) * M + pos] : 0, up = q < M - 1 ? S[(size_t)(q + 1) * M + pos] : 0, u0 = S[(size_t)q * M + pos];
        // (T u)_i of elliptic.C:556-579 with eta = 1: xP - xM < 0 on the decreasing Gauss-Lobatto grid
        const long double Tu = idx * (idxM * um + idxP * up) - idx * (idxP + idxM) * u0;
        e_eig = fmaxl(e_eig, fabsl(Tu - lam[pos] * u0) / (lmax * smax));
        const long double sg = pos < He ? 1.0L : -1.0L;
        e_par = fmaxl(e_par, fabsl(u0 - sg * S[(size_t)(M - 1 - q) * M + pos]) / smax);
      }
    // ascending eigenvalues inside each parity class; all positive
    int sorted = 1;
    for (int q = 1; q < He; q++) if (!(lam[q] > lam[q - 1])) sorted = 0;
    for (int q = 1; q < M / 2; q++) if (!(lam[M - 1 - q] > lam[M - q])) sorted = 0;
    for (int q = 0; q < M; q++) if (!(lam[q] > 0)) sorted = 0;
    printf("%d %.3Le %.3Le %.3Le %d\n", P, e_inv, e_eig, e_par, sorted);
  }
  return 0;
}
