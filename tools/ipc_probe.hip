// ipc_probe.hip -- can two PROCESSES do what the LOCAL transport's thread ranks do (csrc/comm.hip: comm_rendezvous)?
//   1. hipIpcGetMemHandle of a pointer INSIDE an allocation (a sub-block of a caching allocator), opened by the other process;
//   2. order the consumer's kernel behind the producer's without host synchronisation, three ways (argv[1]):
//        0  an interprocess event (hipEventInterprocess): record on the producer's stream, hipStreamWaitEvent on the consumer's
//        1  a sequence number in host memory both processes registered: hipStreamWriteValue64 / hipStreamWaitValue64
//        2  the same number written by a one-thread kernel and awaited by a one-thread kernel that polls it (with a time limit)
//   3. what a post / wait pair costs on the host and per round, and whether the wait really orders the kernels (a slow producer).
// The parent forks BEFORE any HIP call; the two talk through an anonymous shared mapping.
// build: hipcc -O2 --offload-arch=gfx950 tools/ipc_probe.hip -o tools/ipc_probe ; run: tools/ipc_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

struct Shared {
  hipIpcMemHandle_t mh; hipIpcEventHandle_t eh;
  std::atomic<int> stage, seq, ack, fail;
  long offset;
  alignas(64) unsigned long long flag;                                  // the device-written sequence number of modes 1, 2
};
__global__ void k_post(unsigned long long *flag, unsigned long long v) {
  __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait(const unsigned long long *flag, unsigned long long v, unsigned long long limit_ticks, int *timed_out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
    if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) { *timed_out = 1; return; }
    __builtin_amdgcn_s_sleep(8);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%s] %s: %s\n", who, #x, hipGetErrorString(e_)); sh->fail = 1; return 1; } } while (0)

__global__ void k_fill(double *p, long n, double v, int spin) {
  // a deliberately slow producer: the values land late, so a consumer that is not ordered behind it reads the old ones
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double x = v;
    for (int s = 0; s < spin; s++) x = x * 1.0000000001 + 0.0;
    p[i] = x > 1e300 ? 0.0 : v;
  }
}
__global__ void k_check(const double *p, long n, double v, unsigned long long *bad) {
  unsigned long long b = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b += p[i] != v;
  if (b) atomicAdd(bad, b);
}

static bool wait_for(std::atomic<int> &a, int v, Shared *sh) {
  auto t0 = std::chrono::steady_clock::now();
  while (a.load() < v) {
    if (sh->fail.load()) return false;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) { sh->fail = 1; return false; }
  }
  return true;
}

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  Shared *sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (sh == MAP_FAILED) { perror("mmap"); return 1; }
  memset((void *)sh, 0, sizeof *sh);
  const long n = 1 << 20; const int iters = 200;
  pid_t pid = fork();
  if (pid < 0) { perror("fork"); return 1; }
  if (pid != 0) {                                                       // ---- producer
    const char *who = "producer";
    double *base; CK(hipMalloc((void **)&base, 64 << 20));
    double *ptr = base + (1 << 17);                                     // 1 MiB into the allocation
    hipError_t e = hipIpcGetMemHandle(&sh->mh, ptr);
    printf("[producer] hipIpcGetMemHandle(inner pointer): %s\n", hipGetErrorString(e));
    sh->offset = 1 << 17;                                               // (measured: the handle of an inner pointer opens at the allocation's BASE)
    if (e != hipSuccess) { (void)hipGetLastError(); CK(hipIpcGetMemHandle(&sh->mh, base)); }
    {  // what finding the allocation of a pointer and taking its handle again cost per call, and whether the handle is stable
      hipIpcMemHandle_t h2, h1; void *b = nullptr; size_t sz = 0; double ta = 0, tb = 0; int same = 1;
      for (int i = 0; i < 200; i++) {
        auto a = std::chrono::steady_clock::now();
        CK(hipMemGetAddressRange((hipDeviceptr_t *)&b, &sz, (hipDeviceptr_t)ptr));
        auto m = std::chrono::steady_clock::now();
        CK(hipIpcGetMemHandle(&h2, b));
        auto z = std::chrono::steady_clock::now();
        ta += std::chrono::duration<double>(m - a).count(); tb += std::chrono::duration<double>(z - m).count();
        if (i == 0) h1 = h2;
        same = same && memcmp(&h2, &h1, sizeof h2) == 0 && b == (void *)base && sz == (size_t)(64 << 20);
      }
      printf("[producer] the handle of the inner pointer %s the handle of the base\n", memcmp(&h1, &sh->mh, sizeof h1) == 0 ? "equals" : "differs from");
      printf("[producer] hipMemGetAddressRange %.2f us, hipIpcGetMemHandle %.2f us per call; base, size and handle stable: %s\n", ta / 200 * 1e6, tb / 200 * 1e6, same ? "yes" : "NO");
    }
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
    CK(hipIpcGetEventHandle(&sh->eh, ev));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *dflag = nullptr;
    if (mode) { CK(hipHostRegister((void *)sh, sizeof *sh, hipHostRegisterMapped)); CK(hipHostGetDevicePointer((void **)&dflag, (void *)&sh->flag, 0)); }
    sh->stage = 1;
    if (!wait_for(sh->stage, 2, sh)) { printf("[producer] consumer did not come\n"); return 1; }
    double t_rec = 0;
    auto T0 = std::chrono::steady_clock::now();
    for (int it = 1; it <= iters; it++) {
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, ptr, n, (double)it, it <= 20 ? 20000 : 0);
      auto a = std::chrono::steady_clock::now();
      if (mode == 0) CK(hipEventRecord(ev, st));
      else if (mode == 1) CK(hipStreamWriteValue64(st, dflag, (uint64_t)it, 0));
      else hipLaunchKernelGGL(k_post, dim3(1), dim3(1), 0, st, dflag, (unsigned long long)it);
      t_rec += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
      sh->seq = it;
      if (!wait_for(sh->ack, it, sh)) { printf("[producer] no ack for %d\n", it); return 1; }
      if (it == 20) T0 = std::chrono::steady_clock::now();
    }
    const double per = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count() / (iters - 20);
    printf("[producer] mode %d: post %.1f us per call; one produce -> consume round %.1f us\n", mode, t_rec / iters * 1e6, per * 1e6);
    int status = 0; waitpid(pid, &status, 0);
    CK(hipStreamSynchronize(st));
    return (sh->fail.load() || status) ? 1 : 0;
  }
  // ---- consumer
  const char *who = "consumer";
  if (!wait_for(sh->stage, 1, sh)) return 1;
  double *p; CK(hipIpcOpenMemHandle((void **)&p, sh->mh, hipIpcMemLazyEnablePeerAccess));
  p += sh->offset;
  hipEvent_t ev; CK(hipIpcOpenEventHandle(&ev, sh->eh));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long *bad; CK(hipMalloc((void **)&bad, 8)); CK(hipMemset(bad, 0, 8));
  unsigned long long *dflag = nullptr; int *tmo = nullptr;
  if (mode) { CK(hipHostRegister((void *)sh, sizeof *sh, hipHostRegisterMapped)); CK(hipHostGetDevicePointer((void **)&dflag, (void *)&sh->flag, 0)); }
  CK(hipMalloc((void **)&tmo, 4)); CK(hipMemset(tmo, 0, 4));
  sh->stage = 2;
  double t_wait = 0; unsigned long long total_bad = 0;
  for (int it = 1; it <= iters; it++) {
    if (!wait_for(sh->seq, it, sh)) return 1;
    auto a = std::chrono::steady_clock::now();
    if (mode == 0) CK(hipStreamWaitEvent(st, ev, 0));
    else if (mode == 1) CK(hipStreamWaitValue64(st, dflag, (uint64_t)it, hipStreamWaitValueGte, ~0ull));
    else hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, (const unsigned long long *)dflag, (unsigned long long)it, 500000000ull, tmo);
    t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, st, p, n, (double)it, bad);
    CK(hipStreamSynchronize(st));
    unsigned long long b = 0; CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
    if (b != total_bad) { if (total_bad == 0) printf("[consumer] iteration %d: %llu stale values (the wait did not order the kernels)\n", it, b - total_bad); total_bad = b; }
    sh->ack = it;
  }
  printf("[consumer] mode %d: wait %.1f us per call; stale values in %d rounds: %llu\n", mode, t_wait / iters * 1e6, iters, total_bad);
  (void)hipIpcCloseMemHandle(p - sh->offset);
  return total_bad ? 1 : 0;
}
