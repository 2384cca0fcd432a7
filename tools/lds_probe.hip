// lds_probe.hip -- which LDS access of the 256-point sweep kernel's two tilings conflicts?  (second half:
// SQ_LDS_BANK_CONFLICT reads 44 % of the LDS-active cycles for the JFAST launch of a matvec, DESIGN called its accesses conflict-free.)
// One kernel per access pattern of cheb_sweep_vec4_kernel<32, ..> (csrc/sweep_vec.hip), the same index arithmetic, nothing else:
//   jfast_park      ds_write_b128  idx = ld_b * LDJ + 2 * ld_a + slot * 8 * LDJ          (LDJ = 130: pitch = 2 mod 32 doubles)
//   jfast_frag      ds_read_b64    (nb + l16) * LDJ + kq + 4 k                            (the MFMA operand reads of a chain)
//   jfast_frag<..>  the same reads with pitch 129 (odd: what round 6 ships), 131, 133, 132 (= 4 mod 32), for comparison
//   jfast_park2     the park of an odd-pitch image: two 8-byte halves per lane (ds_write2_b64)
//   colfast_park    ds_write_b128  idx = ld_b * NT + ((2 ld_a) ^ ((ld_b & 1) << 4)) + slot * 32 * NT
//   colfast_frag    ds_read_b64    kq * NT + ((nb + l16) ^ ((kq & 1) << 4)) + 4 k NT
//   resident_frag   ds_read_b64    the LDS-resident matrix fragments: w * NFL * 64 + lane + 64 s
// Run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS (tools/r06_lds_probe.sh):
// conflict cycles per LDS instruction of each kernel.  build: hipcc -O3 --offload-arch=gfx950 tools/lds_probe.hip -o tools/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int KS = 32, HP = 4 * KS, NT = 32, REPS = 2000;

template <int LDJ>
__global__ __launch_bounds__(512) void jfast_park(double *out) {
  __shared__ double smem[NT * LDJ + 16];
  const int tid = threadIdx.x, ld_a = tid % (HP / 2), ld_b = tid / (HP / 2);
  const int idx0 = ld_b * LDJ + 2 * ld_a;
  d2 v = d2{(double)tid, 1.0};
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int s = 0; s < 4; s++) *(d2 *)(smem + idx0 + s * 8 * LDJ) = v;
    v.x += 1.0;
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  out[blockIdx.x * 512 + tid] = smem[tid];
}

// the park of an image whose lines start on 8-byte boundaries (odd pitch): two 8-byte halves -> ds_write2_b64
template <int LDJ>
__global__ __launch_bounds__(512) void jfast_park2(double *out) {
  __shared__ double smem[NT * LDJ + 16];
  const int tid = threadIdx.x, ld_a = tid % (HP / 2), ld_b = tid / (HP / 2);
  const int idx0 = ld_b * LDJ + 2 * ld_a;
  double vx = (double)tid, vy = 1.0;
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int s = 0; s < 4; s++) { smem[idx0 + s * 8 * LDJ] = vx; smem[idx0 + s * 8 * LDJ + 1] = vy; }
    vx += 1.0;
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  out[blockIdx.x * 512 + tid] = smem[tid];
}

template <int LDJ>
__global__ __launch_bounds__(512) void jfast_frag(double *out) {
  __shared__ double smem[NT * LDJ + 16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kq = lane >> 4, l16 = lane & 15;
  for (int i = tid; i < NT * LDJ; i += 512) smem[i] = (double)i;
  __syncthreads();
  double acc = 0.0;
  for (int r = 0; r < REPS; r++) {
    const int nb = (r & 1) * 16;
    const double *f = smem + (nb + l16) * LDJ + kq;
#pragma unroll
    for (int k = 0; k < KS; k++) acc += f[4 * k];
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 512 + tid] = acc + w;
}

__global__ __launch_bounds__(512) void colfast_park(double *out) {
  __shared__ double smem[HP * NT];
  const int tid = threadIdx.x, ld_a = tid % (NT / 2), ld_b = tid / (NT / 2);
  const int idx0 = ld_b * NT + ((2 * ld_a) ^ ((ld_b & 1) << 4));
  d2 v = d2{(double)tid, 1.0};
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int s = 0; s < 4; s++) *(d2 *)(smem + idx0 + s * 32 * NT) = v;
    v.x += 1.0;
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  out[blockIdx.x * 512 + tid] = smem[tid];
}

__global__ __launch_bounds__(512) void colfast_frag(double *out) {
  __shared__ double smem[HP * NT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kq = lane >> 4, l16 = lane & 15;
  for (int i = tid; i < HP * NT; i += 512) smem[i] = (double)i;
  __syncthreads();
  double acc = 0.0;
  for (int r = 0; r < REPS; r++) {
    const int nb = (r & 1) * 16;
    const double *f = smem + kq * NT + ((nb + l16) ^ ((kq & 1) << 4));
#pragma unroll
    for (int k = 0; k < KS; k++) acc += f[4 * k * NT];
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 512 + tid] = acc + w;
}

__global__ __launch_bounds__(512) void resident_frag(double *out) {
  constexpr int NFL = 8;
  __shared__ double smem[8 * NFL * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 8 * NFL * 64; i += 512) smem[i] = (double)i;
  __syncthreads();
  double acc = 0.0;
  const double *f = smem + w * NFL * 64 + lane;
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int s = 0; s < NFL; s++) acc += f[64 * s];
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 512 + tid] = acc;
}

int main() {
  double *out;
  if (hipMalloc((void **)&out, 256 * 512 * sizeof(double)) != hipSuccess) { printf("no device\n"); return 1; }
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL((jfast_park<130>), dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL((jfast_park<132>), dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL((jfast_park2<129>), dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL((jfast_frag<131>), dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL((jfast_frag<133>), dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL((jfast_frag<130>), dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL((jfast_frag<132>), dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL((jfast_frag<129>), dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(colfast_park, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(colfast_frag, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(resident_frag, dim3(256), dim3(512), 0, 0, out);
  }
  hipError_t e = hipDeviceSynchronize();
  printf("lds_probe: %s (instructions per kernel and wave: park 4 x %d ds_write_b128, frag 32 x %d ds_read_b64, resident 8 x %d ds_read_b64)\n",
         hipGetErrorString(e), REPS, REPS, REPS);
  return e == hipSuccess ? 0 : 1;
}
