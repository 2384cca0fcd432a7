// calib_traffic.hip -- known-byte-count kernels in the sweep kernel's access shape (8 B per lane,
// 256-B row pieces) to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (the guide: widths
// other than 16 B/lane are uncalibrated).  Each kernel moves exactly NBYTES = 1 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double *x, double *out, long n) {
  double s = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += x[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void write8(double *x, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = (double)i;
}
// strided rows like the COLFAST tile: 32 consecutive doubles per row piece, row stride 65536 doubles
__global__ void read8_rows(const double *x, double *out, long n) {
  double s = 0;
  const int lane32 = threadIdx.x & 31;
  for (long piece = blockIdx.x * (long)(blockDim.x / 32) + (threadIdx.x >> 5); piece < n / 32; piece += (long)gridDim.x * (blockDim.x / 32)) {
    const long col = piece % 2048, row = piece / 2048;     // 2048 pieces of 32 doubles per 65536-double row
    s += x[row * 65536 + col * 32 + lane32];
  }
  if (s == 12345.678) out[0] = s;
}
int main() {
  const long n = 1L << 27;  // 1 GiB of doubles
  double *x, *o; hipMalloc(&x, n * 8); hipMalloc(&o, 64);
  hipMemset(x, 0, n * 8);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, x, o, n);
    hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, x, n);
    hipLaunchKernelGGL(read8_rows, dim3(4096), dim3(256), 0, 0, x, o, n);
  }
  hipDeviceSynchronize();
  printf("calib done: each kernel moves %ld bytes\n", n * 8);
  return 0;
}
