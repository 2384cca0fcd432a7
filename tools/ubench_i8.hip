// ubench_i8.hip -- is an Ozaki-style split of the f64 product onto the int8 matrix pipe worth building (DESIGN "next")?
//   (1) sustained rate of v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8 (gfx950),
//   (2) the same with FP64 VALU work interleaved in the instruction stream (does int8 MFMA co-execute with VALU, unlike FP64 MFMA?),
//   (3) the FP64 MFMA rate of the same loop shape for reference.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_i8.hip -o ubench_i8 ; run on the GPU box (tools/ubench_i8.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));

// mode 0: 32x32x32 i8; 1: 16x16x64 i8; 2: 32x32x32 i8 + 8 FP64 FMAs per MFMA in one stream; 3: f64 16x16x4
__global__ __launch_bounds__(512) void k(int mode, int iters, double *out) {
  const v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
  v16i c0 = {}, c1 = {};
  v4i d0 = {}, d1 = {};
  v4d e0 = {0, 0, 0, 0}, e1 = e0;
  double f0 = 1.0 + 1e-9 * threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  const double x = 1.0 + 1e-9 * threadIdx.x, y = 1.0 - 1e-9 * threadIdx.x;
  if (mode == 0) {
    for (int i = 0; i < iters; i++) { c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0); }
  } else if (mode == 1) {
    for (int i = 0; i < iters; i++) { d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0); }
  } else if (mode == 2) {
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      f0 = __builtin_fma(f0, x, y); f1 = __builtin_fma(f1, x, y); f2 = __builtin_fma(f2, x, y); f3 = __builtin_fma(f3, x, y);
      f4 = __builtin_fma(f4, x, y); f5 = __builtin_fma(f5, x, y); f6 = __builtin_fma(f6, x, y); f7 = __builtin_fma(f7, x, y);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      f0 = __builtin_fma(f0, y, x); f1 = __builtin_fma(f1, y, x); f2 = __builtin_fma(f2, y, x); f3 = __builtin_fma(f3, y, x);
      f4 = __builtin_fma(f4, y, x); f5 = __builtin_fma(f5, y, x); f6 = __builtin_fma(f6, y, x); f7 = __builtin_fma(f7, y, x);
    }
  } else {
    for (int i = 0; i < iters; i++) { e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, e0, 0, 0, 0); e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, e1, 0, 0, 0); }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + d0[0] + d1[1] + e0[0] + e1[1] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

int main() {
  double *out; if (hipMalloc(&out, 1024 * 512 * sizeof(double)) != hipSuccess) return 1;
  hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
  const int iters = 20000, grid = 256;
  const char *names[] = {"v_mfma_i32_32x32x32_i8", "v_mfma_i32_16x16x64_i8", "32x32x32 i8 + 8 f64 FMAs per MFMA (one stream)", "v_mfma_f64_16x16x4_f64"};
  const double ops[] = {2.0 * 32 * 32 * 32, 2.0 * 16 * 16 * 64, 2.0 * 32 * 32 * 32, 2.0 * 16 * 16 * 4};
  for (int m = 0; m < 4; m++)
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(t0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, m, iters, out);
      hipEventRecord(t1); hipEventSynchronize(t1);
      float ms; hipEventElapsedTime(&ms, t0, t1);
      if (rep == 2) {
        const double total = ops[m] * 2.0 * iters * 8 * grid;       // two MFMAs per iteration, 8 waves per workgroup
        const double valu = m == 2 ? 128.0 * 16 * iters * 8 * grid : 0.0;
        printf("%-52s %8.3f ms  %8.1f T(F)LOP/s matrix pipe%s\n", names[m], ms, total / ms / 1e9, "");
        if (valu > 0) printf("%-52s            %8.1f TFLOP/s FP64 VALU beside it\n", "", valu / ms / 1e9);
      }
    }
  return 0;
}
