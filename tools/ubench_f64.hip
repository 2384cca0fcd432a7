// ubench_f64.hip -- microbenchmarks that size the sweep kernel on gfx950:
//   (1) issue rate of v_mfma_f64_16x16x4_f64, (2) v_fma_f64 rate, (3) whether the two pipes
//   overlap when they run in different waves of one SIMD, (4) ... in one wave's stream.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_f64.hip -o ubench_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

// mode bit0: waves 0-3 run MFMA; bit1: waves 4-7 run VALU FMA; bit2: waves 4-7 run MFMA; bit3: waves 0-3 VALU
// mode 16: every wave interleaves MFMA and VALU in one stream
__global__ __launch_bounds__(512) void k(int mode, int iters, double *out) {
  const int w = threadIdx.x >> 6;
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double f0 = a, f1 = b, f2 = a + 1, f3 = b + 1, f4 = a + 2, f5 = b + 2, f6 = a + 3, f7 = b + 3;
  const bool lo = w < 4;
  const bool do_mfma = (lo && (mode & 1)) || (!lo && (mode & 4)) || (mode & 16);
  const bool do_valu = (!lo && (mode & 2)) || (lo && (mode & 8)) || (mode & 16);
  if (mode & 16) {
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      f0 = __builtin_fma(f0, b, a); f1 = __builtin_fma(f1, b, a); f2 = __builtin_fma(f2, b, a); f3 = __builtin_fma(f3, b, a);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      f4 = __builtin_fma(f4, b, a); f5 = __builtin_fma(f5, b, a); f6 = __builtin_fma(f6, b, a); f7 = __builtin_fma(f7, b, a);
    }
  } else if (do_mfma) {
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  } else if (do_valu) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
        f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

// layout probe: D = A(16x4) * B(4x16) with A[i][k] = i + 100k, B[k][j] = (k==K0) ? (j==J0) : 0
__global__ void layout(double *out) {
  const int l = threadIdx.x;
  for (int k0 = 0; k0 < 4; k0++) {
    const double a = (l & 15) + 100.0 * (l >> 4);          // claimed: A[i=l&15][k=l>>4]
    const double b = ((l >> 4) == k0) ? 1000.0 + (l & 15) : 0.0;  // claimed: B[k=l>>4][j=l&15]
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[(k0 * 4 + r) * 64 + l] = c[r];
  }
}

int main() {
  double *out; hipMalloc(&out, 1024 * 512 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 256;
  const char *names[] = {"mfma waves0-3 only", "valu waves4-7 only", "mfma w0-3 + valu w4-7", "mfma all 8 waves",
                         "valu all 8 waves", "interleaved in one stream (8 waves)", "interleaved in one stream, 4 waves"};
  int modes[] = {1, 2, 3, 5, 10, 16, 16};
  int threads[] = {512, 512, 512, 512, 512, 512, 256};
  for (int t = 0; t < 7; t++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(threads[t]), 0, 0, modes[t], iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) {
        const int m = modes[t];
        const int nw = threads[t] / 64;
        double mf = 0, vf = 0;  // flops
        const double per_mfma = 2048.0, per_fma = 128.0;
        if (m & 16) { mf = nw * 4.0 * per_mfma; vf = nw * 16.0 * per_fma; }
        else { mf = (((m & 1) ? 4 : 0) + ((m & 4) ? 4 : 0)) * 4.0 * per_mfma; vf = (((m & 2) ? 4 : 0) + ((m & 8) ? 4 : 0)) * 16.0 * per_fma; }
        mf *= (double)iters * grid; vf *= (double)iters * grid;
        printf("%-40s %8.3f ms  mfma %7.2f TF  valu %7.2f TF  total %7.2f TF\n", names[t], ms, mf / ms * 1e-9, vf / ms * 1e-9, (mf + vf) / ms * 1e-9);
      }
    }
  }
  // layout probe
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
  std::vector<double> h(16 * 64);
  hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  // expected D[i][j] = A[i][k0] * B[k0][j] = (i + 100 k0) * (1000 + j).  Test the claimed C/D map row=(l>>4)+4r, col=l&15
  int bad_claim = 0, bad_alt = 0;
  for (int k0 = 0; k0 < 4; k0++) for (int r = 0; r < 4; r++) for (int l = 0; l < 64; l++) {
    const double v = h[(k0 * 4 + r) * 64 + l];
    const int j = l & 15;
    const int i1 = (l >> 4) + 4 * r, i2 = 4 * (l >> 4) + r;
    if (v != (i1 + 100.0 * k0) * (1000.0 + j)) bad_claim++;
    if (v != (i2 + 100.0 * k0) * (1000.0 + j)) bad_alt++;
  }
  printf("C/D layout: row=(l>>4)+4r mismatches %d ; row=4(l>>4)+r mismatches %d\n", bad_claim, bad_alt);
  return 0;
}
