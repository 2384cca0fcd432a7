// comm.hip -- transports of the slab-partitioned operators (the reference is serial: elliptic.C:262, nk.c:63).
//
// chebhip_comm is what the distributed drivers (dist.hip: linear Poisson; slabx.hip: general elliptic and Stokes)
// exchange through.  Three kinds:
//   RCCL      one process per GPU: every exchange is ONE grouped ncclSend / ncclRecv launch (rccl.h:700,722,923),
//             G-1 direct xGMI messages per GPU; reductions are ncclAllReduce (rccl.h:611).  RCCL is looked up at run
//             time (the copy the process already holds, e.g. PyTorch's, else the system one) and never linked.
//   LOCAL     ranks are host threads of ONE process, each with its own stream (and its own device when the node's
//             GPUs are driven from one process: peer access is enabled at create).  An exchange is event-ordered
//             device copies: every rank PULLS its segments out of its peers' send buffers.  On one device this is the
//             full-size rehearsal of an N-rank run (tests/test_gpu_dist_emul.py).
//   CALLBACK  anything else (the gloo staging of the tests).
//   IPC       ranks are PROCESSES of one node (the launcher's one process per GPU): the LOCAL transport's rendezvous across address
//             spaces.  Pointers travel as (hipIpcMemHandle_t of the allocation, offset) through a POSIX shared-memory segment, the
//             "my arrays are complete here" events are sequence numbers in that segment written by hipStreamWriteValue64 and awaited
//             by ONE polling launch per rendezvous (interprocess hipEvents fail after 32 records on this runtime: tools/ipc_probe.hip),
//             the barrier is a counter in it.  A DIRECT transport layered over a message transport (RCCL, a callback) that carries the
//             segment exchanges and the reductions.
#include "comm.h"
#include "sweep.h"
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

int chebhip_fail(int code, const char *fmt, ...);   // chebhip.hip

#define CHIPCHK(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {

// ---- RCCL through dlopen (rccl.h:236-923; types restated so that no header of the library is needed) ---------
struct Id128 { char internal[128]; };     // ncclUniqueId
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(void *id) = nullptr;                                       // ncclGetUniqueId(ncclUniqueId*): 128 bytes
  int (*CommInitRank)(void **comm, int nranks, Id128 id, int rank) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*GroupStart)() = nullptr, (*GroupEnd)() = nullptr;
  int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
  int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
  int (*AllReduce)(const void *s, void *r, size_t count, int dtype, int op, void *comm, hipStream_t st) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;
RcclApi g_rccl;
std::once_flag g_rccl_once;

bool rccl_ready() {
  std::call_once(g_rccl_once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) if (!g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // the process's own copy first
    for (const char *n : names) if (!g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.lib) return;
    void *L = g_rccl.lib;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(L, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(L, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(L, "ncclCommDestroy");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(L, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(L, "ncclGroupEnd");
    g_rccl.Send = (decltype(g_rccl.Send))dlsym(L, "ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))dlsym(L, "ncclRecv");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(L, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(L, "ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.GroupStart && g_rccl.GroupEnd &&
                g_rccl.Send && g_rccl.Recv && g_rccl.AllReduce;
  });
  return g_rccl.ok;
}
int rccl_fail(const char *what, int rc) {
  return chebhip_fail(CHEBHIP_ERR_DEVICE, "%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
}

enum { KIND_RCCL = 1, KIND_LOCAL = 2, KIND_CALLBACK = 3, KIND_NULL = 4, KIND_IPC = 5 };
constexpr int MAXR = 64;

// sum of the G posted vectors, taken in rank order on every rank: all ranks get the same bits
struct RedPtrs { const double *p[MAXR]; };
__global__ void k_local_reduce(RedPtrs in, int G, int count, double *__restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    double s = in.p[0][i];
    for (int r = 1; r < G; r++) s = s + in.p[r][i];
    out[i] = s;
  }
}

}  // namespace

// ---- ranks as host threads of one process ----------------------------------------------------------------------
struct chebhip_local_group {
  int G = 0;
  std::mutex mu;
  std::condition_variable cv;
  int count = 0; long gen = 0; bool aborted = false;
  double timeout_s = 120.0;
  struct Slot { const chebhip::XSeg *segs = nullptr; int nseg = 0; hipEvent_t ready = nullptr, done = nullptr; const double *vals = nullptr; int device = -1; bool bound = false;
                hipEvent_t xev[chebhip::COMM_NEV] = {nullptr, nullptr, nullptr, nullptr}; const double *xptr[2][chebhip::COMM_NPTR] = {{nullptr, nullptr}, {nullptr, nullptr}};
                unsigned long rdv = 0; } slot[MAXR];   // rdv: rendezvous this rank has made (all communicators of the group: the ranks call them in lockstep)

  // all G threads arrive, or the group is aborted (a rank failed, or did not come within the time limit)
  int barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return chebhip_fail(CHEBHIP_ERR_DEVICE, "local group: aborted by another rank");
    const long my = gen;
    if (++count == G) { count = 0; gen++; cv.notify_all(); return 0; }
    const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return gen != my || aborted; });
    if (!ok) { aborted = true; cv.notify_all(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "local group: a rank did not arrive within %.0f s", timeout_s); }
    if (gen == my && aborted) return chebhip_fail(CHEBHIP_ERR_DEVICE, "local group: aborted by another rank");
    return 0;
  }
  void abort() { std::lock_guard<std::mutex> lk(mu); aborted = true; cv.notify_all(); }
};


// ---- ranks as processes of one node ------------------------------------------------------------------------------
namespace {
struct IpcPtr { hipIpcMemHandle_t h; unsigned long long base, off; int valid; int pad_; };     // base: the allocation's address in its owner's space
struct IpcSlot {
  int bound, pid; char bus[32];                                          // bus: PCI id of the rank's device ("do the ranks share a device?")
  unsigned long long seq;                                                // posts (of any event slot) ENQUEUED so far, and ...
  unsigned long long posted[chebhip::COMM_NEV];                          // ... the number the latest post of each slot got (written by the owning rank only)
  unsigned long long rdv;                                                // rendezvous this rank has made
  IpcPtr xptr[2][chebhip::COMM_NPTR];                                    // alternating tables, as in the LOCAL group
  unsigned long long want[2];                                            // ... with the number the peers wait for: the later of `slot`'s and `wait_slot`'s post
  alignas(64) unsigned long long flag;                                   // posts EXECUTED by the rank's stream (device-written, in stream order)
};
struct IpcShared {
  std::atomic<unsigned> magic; int G;
  std::atomic<int> aborted, attached;                                     // attached: ranks that hold the mapping (rank 0 unlinks the name after the last)
  alignas(64) std::atomic<unsigned> bar_count;
  alignas(64) std::atomic<unsigned> bar_gen;
  IpcSlot slot[MAXR];
};
constexpr unsigned IPC_MAGIC = 0x43484950u;

// One launch instead of 2 (G - 1) hipStreamWaitValue64 calls of 2.5 us each (at G = 8 the host would spend 70 us per matvec
// enqueueing waits): lane r polls the sequence number of rank r until it reaches what that rank announced.  Every lane leaves
// after `limit` ticks of the 100-MHz clock at the latest and then marks the group aborted (the ranks' next barrier fails) -- no
// wave waits forever for a peer that died.
// EIGHT workgroups, one per XCD (workgroups are dealt to the XCDs round-robin), each ending with a system-scope acquire: what this
// XCD's L2 still holds of the peers' arrays -- they rewrite the same addresses call after call -- is dropped before the kernels
// behind this one read them.  (Between LOCAL thread ranks the runtime's cross-device event wait does that for the whole device;
// here the dispatches that follow carry agent-scope acquires only.  The same fence as the first statement of every wave of the
// reading kernels cost the G = 8 rank 31 us of 45: once per XCD and rendezvous is what is needed.)
struct IpcWait { const unsigned long long *flag[MAXR]; unsigned long long want[MAXR]; };
__global__ __launch_bounds__(64) void k_ipc_wait(IpcWait a, int G, int me, unsigned long long limit, int *aborted) {
  const int r = threadIdx.x;
  if (r < G && r != me && a.want[r] != 0) {
    const unsigned long long want = a.want[r], t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(a.flag[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      if (__hip_atomic_load(aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;     // a rank gave up: its number will not come
      if (__builtin_amdgcn_s_memrealtime() - t0 > limit) { __hip_atomic_store(aborted, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
static_assert(std::atomic<unsigned>::is_always_lock_free, "the barrier of the IPC group lives in memory shared between processes");
}  // namespace

struct chebhip_ipc_group {
  IpcShared *S = nullptr; char *dS = nullptr;                             // the segment, and its device view (hipHostRegister)
  size_t bytes = 0; int G = 0, rank = -1; bool registered = false;
  double timeout_s = 120.0;
  hipEvent_t fence = nullptr;                                             // recorded in front of every post: a system-scope release of what the stream wrote
  bool pending = false; hipStream_t pending_st = nullptr;                 // a mark whose number has not been written yet (ipc_post)
  struct Opened { hipIpcMemHandle_t h; void *mapped; };
  std::map<std::pair<int, unsigned long long>, Opened> opened;            // peers' allocations mapped here, by (rank, base in its space)
  std::vector<void *> retired;                                            // mappings of allocations their owner has replaced: closed with the group

  unsigned long long *dflag(int r) const { return (unsigned long long *)(dS + ((char *)&S->slot[r].flag - (char *)S)); }
  int barrier() {
    if (S->aborted.load()) return chebhip_fail(CHEBHIP_ERR_DEVICE, "ipc group: aborted by another rank");
    const unsigned gen = S->bar_gen.load(std::memory_order_acquire);
    if (S->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)G) {
      S->bar_count.store(0, std::memory_order_relaxed);
      S->bar_gen.fetch_add(1, std::memory_order_release);
      return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long spins = 0; S->bar_gen.load(std::memory_order_acquire) == gen; spins++) {
      if (S->aborted.load(std::memory_order_relaxed)) return chebhip_fail(CHEBHIP_ERR_DEVICE, "ipc group: aborted by another rank");
      if ((spins & 1023) == 1023) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
          S->aborted.store(1);
          return chebhip_fail(CHEBHIP_ERR_DEVICE, "ipc group: a rank did not arrive within %.0f s", timeout_s);
        }
        if (spins > (1ul << 16)) sched_yield();                           // a late peer: stop burning the core it may need
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    return 0;
  }
  // (handle of the allocation that holds p, its base, offset of p in it).  The handle is taken at EVERY call (hipMemGetAddressRange
  // 0.1 us + hipIpcGetMemHandle 2.8 us, tools/ipc_probe.hip): an allocation that was freed and made again at the same address has a
  // new handle, and the peers then map the new memory instead of reading the old one through a mapping that keeps it alive.
  int publish(const double *p, IpcPtr *out) {
    out->valid = 0;
    if (!p) return 0;
    void *base = nullptr; size_t size = 0;
    hipError_t e = hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)p);
    if (e != hipSuccess || !base) { (void)hipGetLastError(); return chebhip_fail(CHEBHIP_ERR_ARG, "ipc transport: %p is not inside a device allocation (%s)", (const void *)p, hipGetErrorString(e)); }
    e = hipIpcGetMemHandle(&out->h, base);
    if (e != hipSuccess) { (void)hipGetLastError(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipIpcGetMemHandle: %s (allocations of a virtual-memory allocator cannot be shared)", hipGetErrorString(e)); }
    out->base = (unsigned long long)base; out->off = (unsigned long long)((const char *)p - (const char *)base); out->valid = 1;
    return 0;
  }
  // rank r's pointer in this process: its allocation is mapped once and the mapping kept while r publishes the same handle for that
  // base; a new handle (r freed the allocation and got the address again) replaces the mapping (the old one is closed with the group:
  // rare, and nothing has to be proved about kernels still in flight)
  int translate(int r, const IpcPtr &q, const double **out) {
    *out = nullptr;
    if (!q.valid) return 0;
    const auto key = std::make_pair(r, q.base);
    auto it = opened.find(key);
    if (it != opened.end() && memcmp(&it->second.h, &q.h, sizeof q.h) != 0) {
      retired.push_back(it->second.mapped);
      opened.erase(it); it = opened.end();
    }
    if (it == opened.end()) {
      void *m = nullptr;
      hipError_t e = hipIpcOpenMemHandle(&m, q.h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess || !m) { (void)hipGetLastError(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipIpcOpenMemHandle: %s", hipGetErrorString(e)); }
      it = opened.emplace(key, Opened{q.h, m}).first;
    }
    *out = (const double *)((const char *)it->second.mapped + q.off);
    return 0;
  }
};

// Collective among the G processes.  `name`: a POSIX shared-memory name ("/chebhip-<unique>") that rank 0 creates and unlinks again as
// soon as everybody holds the mapping (a crash later leaves nothing behind); it must be unique per group -- a stale segment of the
// same name would be taken for the new one.  Call with the rank's device current.
extern "C" int chebhip_ipc_group_open(const char *name, int nranks, int rank, chebhip_ipc_group **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!name || name[0] != '/' || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return chebhip_fail(CHEBHIP_ERR_ARG, "bad argument");
  chebhip_ipc_group *g = new (std::nothrow) chebhip_ipc_group;
  if (!g) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  g->G = nranks; g->rank = rank;
  { const int t = chebhip::opt(chebhip::OPT_LOCAL_TIMEOUT_S); if (t > 0) g->timeout_s = (double)t; }
  const long page = sysconf(_SC_PAGESIZE);
  g->bytes = (sizeof(IpcShared) + (size_t)page - 1) / (size_t)page * (size_t)page;
  auto fail = [&](int code, const char *what, const char *why) { chebhip_ipc_group_close(g); return chebhip_fail(code, "ipc group %s: %s: %s", name, what, why); };
  int fd = -1;
  const auto t0 = std::chrono::steady_clock::now();
  auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > g->timeout_s; };
  // rank 0 takes the name away again -- at once when everybody holds the mapping; after a failure of its own only when the others
  // have found the segment (and its abort mark: they fail at once instead of searching a vanished name), or after 5 s
  auto unlink_when_attached = [&] {
    const auto u0 = std::chrono::steady_clock::now();
    while (g->S && g->S->attached.load() < nranks && std::chrono::duration<double>(std::chrono::steady_clock::now() - u0).count() < 5.0) usleep(500);
    (void)shm_unlink(name);
  };
  if (rank == 0) {
    (void)shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail(CHEBHIP_ERR_DEVICE, "shm_open", strerror(errno));
    if (ftruncate(fd, (off_t)g->bytes) != 0) { close(fd); (void)shm_unlink(name); return fail(CHEBHIP_ERR_DEVICE, "ftruncate", strerror(errno)); }
  } else {
    for (;;) {                                                            // until rank 0 has made it and given it its size
      fd = shm_open(name, O_RDWR, 0600);
      struct stat sb;
      if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= g->bytes) break;
      if (fd >= 0) { close(fd); fd = -1; }
      if (late()) return fail(CHEBHIP_ERR_DEVICE, "shm_open", "rank 0 did not create the segment in time");
      usleep(1000);
    }
  }
  void *m = mmap(nullptr, g->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { if (rank == 0) (void)shm_unlink(name); return fail(CHEBHIP_ERR_DEVICE, "mmap", strerror(errno)); }
  g->S = (IpcShared *)m;
  if (rank == 0) {                                                        // (a fresh segment is zero-filled)
    g->S->G = nranks;
    g->S->magic.store(IPC_MAGIC, std::memory_order_release);
  } else {
    while (g->S->magic.load(std::memory_order_acquire) != IPC_MAGIC) { if (late()) return fail(CHEBHIP_ERR_DEVICE, "attach", "rank 0 did not initialise the segment in time"); usleep(200); }
    if (g->S->G != nranks) return fail(CHEBHIP_ERR_ARG, "attach", "the segment belongs to a group of another size");
  }
  g->S->attached.fetch_add(1);
  if (g->S->aborted.load()) { if (rank == 0) unlink_when_attached(); return fail(CHEBHIP_ERR_DEVICE, "attach", "another rank could not set the group up"); }
  hipError_t e = hipHostRegister(m, g->bytes, hipHostRegisterMapped);
  if (e == hipSuccess) { g->registered = true; e = hipHostGetDevicePointer((void **)&g->dS, m, 0); }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&g->fence, hipEventDisableTiming);
  int dev = 0;
  if (e == hipSuccess) e = hipGetDevice(&dev);
  IpcSlot &me = g->S->slot[rank];
  if (e == hipSuccess) e = hipDeviceGetPCIBusId(me.bus, (int)sizeof me.bus, dev);
  if (e != hipSuccess) { (void)hipGetLastError(); g->S->aborted.store(1); if (rank == 0) unlink_when_attached(); return fail(CHEBHIP_ERR_DEVICE, "device setup", hipGetErrorString(e)); }
  me.pid = (int)getpid(); me.bound = 1;
  int rc = g->barrier();                                                  // everybody holds the mapping and has filled its slot
  if (rank == 0) { if (rc) unlink_when_attached(); else (void)shm_unlink(name); }
  if (rc) { chebhip_ipc_group_close(g); return rc; }
  *out = g;
  return 0;
}
extern "C" int chebhip_ipc_group_close(chebhip_ipc_group *g) {
  if (!g) return 0;
  for (auto &kv : g->opened) (void)hipIpcCloseMemHandle(kv.second.mapped);
  for (void *m : g->retired) (void)hipIpcCloseMemHandle(m);
  if (g->fence) (void)hipEventDestroy(g->fence);
  if (g->registered) (void)hipHostUnregister((void *)g->S);
  if (g->S) (void)munmap((void *)g->S, g->bytes);
  (void)hipGetLastError();
  delete g;
  return 0;
}
extern "C" int chebhip_ipc_group_abort(chebhip_ipc_group *g) { if (g && g->S) g->S->aborted.store(1); return 0; }

struct chebhip_comm {
  int kind = 0, G = 1, rank = 0;
  void *nccl = nullptr;                         // RCCL communicator (not owned)
  chebhip_exchangev_fn xfn = nullptr; chebhip_reduce_fn rfn = nullptr; void *ctx = nullptr;
  chebhip_local_group *lg = nullptr;
  double *scratch = nullptr;                    // LOCAL: the reduction's private result (MAXR doubles)
  const double *shadow[chebhip::COMM_NPTR][MAXR] = {};   // NULL: arrays that stand for the peers' (chebhip_comm_null_set_shadow; not owned)
  chebhip_ipc_group *ig = nullptr;              // IPC: the process group (not owned) ...
  chebhip_comm *inner = nullptr;                // ... and the message transport under it (not owned): segment exchanges, reductions
};

extern "C" int chebhip_local_group_create(int nranks, chebhip_local_group **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (nranks < 1 || nranks > MAXR) return chebhip_fail(CHEBHIP_ERR_ARG, "nranks = %d must be in 1..%d", nranks, MAXR);
  chebhip_local_group *g = new (std::nothrow) chebhip_local_group;
  if (!g) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  g->G = nranks;
  { const int t = chebhip::opt(chebhip::OPT_LOCAL_TIMEOUT_S); if (t > 0) g->timeout_s = (double)t; }   // "local_timeout_s"
  *out = g;
  return 0;
}
// The slots' events live as long as the group: a rank that is done (and destroys its communicator) must not take them
// away from a slower peer that is still enqueueing its last waits on them.
extern "C" int chebhip_local_group_destroy(chebhip_local_group *g) {
  if (!g) return 0;
  for (int r = 0; r < MAXR; r++) {
    if (g->slot[r].ready) (void)hipEventDestroy(g->slot[r].ready);
    if (g->slot[r].done) (void)hipEventDestroy(g->slot[r].done);
    for (int k = 0; k < chebhip::COMM_NEV; k++) if (g->slot[r].xev[k]) (void)hipEventDestroy(g->slot[r].xev[k]);
  }
  delete g;
  return 0;
}
// A rank that cannot go on (an error outside the library) releases the ranks waiting for it: their calls fail.
extern "C" int chebhip_local_group_abort(chebhip_local_group *g) { if (g) g->abort(); return 0; }

extern "C" int chebhip_comm_create_local(chebhip_local_group *g, int rank, chebhip_comm **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!g || rank < 0 || rank >= g->G) return chebhip_fail(CHEBHIP_ERR_ARG, "bad group or rank");
  chebhip_comm *c = new (std::nothrow) chebhip_comm;
  if (!c) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  c->kind = KIND_LOCAL; c->G = g->G; c->rank = rank; c->lg = g;
  auto &s = g->slot[rank];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess && !s.ready) e = hipEventCreateWithFlags(&s.ready, hipEventDisableTiming);       // (kept by the group, see its destroy)
  if (e == hipSuccess && !s.done) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
  for (int k = 0; k < chebhip::COMM_NEV; k++)
    if (e == hipSuccess && !s.xev[k]) {
      e = hipEventCreateWithFlags(&s.xev[k], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(s.xev[k], nullptr);       // never-recorded events must not be waited for: start them complete
    }
  if (e == hipSuccess) e = hipMalloc((void **)&c->scratch, MAXR * sizeof(double));
  if (e != hipSuccess) { delete c; return chebhip_fail(CHEBHIP_ERR_DEVICE, "local comm: %s", hipGetErrorString(e)); }
  {
    std::lock_guard<std::mutex> lk(g->mu);
    s.device = dev; s.bound = true;
    // ranks on other devices of this process: the pulls read their buffers directly (xGMI peer access)
    // (multi-device LOCAL groups are untested on hardware: every test box has one GPU)
    int no_peer = -1;
    for (int r = 0; r < g->G; r++)
      if (g->slot[r].bound && g->slot[r].device != dev) {
        int can_a = 0, can_b = 0;
        if (hipDeviceCanAccessPeer(&can_a, dev, g->slot[r].device) != hipSuccess || hipDeviceCanAccessPeer(&can_b, g->slot[r].device, dev) != hipSuccess || !can_a || !can_b) { no_peer = g->slot[r].device; break; }
        (void)hipDeviceEnablePeerAccess(g->slot[r].device, 0);                                  // "already enabled" is fine
        int cur = dev; (void)hipSetDevice(g->slot[r].device); (void)hipDeviceEnablePeerAccess(cur, 0); (void)hipSetDevice(cur);
        (void)hipGetLastError();
      }
    if (no_peer >= 0) {
      s.bound = false;
      (void)hipFree(c->scratch); delete c;
      return chebhip_fail(CHEBHIP_ERR_DEVICE, "local comm: devices %d and %d cannot access each other's memory (the pulls of the LOCAL transport read peers' buffers directly)", dev, no_peer);
    }
  }
  *out = c;
  return 0;
}

extern "C" int chebhip_comm_create_rccl(void *nccl_comm, int nranks, int rank, chebhip_comm **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!nccl_comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return chebhip_fail(CHEBHIP_ERR_ARG, "bad argument");
  if (!rccl_ready()) return chebhip_fail(CHEBHIP_ERR_DEVICE, "librccl.so could not be loaded");
  chebhip_comm *c = new (std::nothrow) chebhip_comm;
  if (!c) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  c->kind = KIND_RCCL; c->G = nranks; c->rank = rank; c->nccl = nccl_comm;
  *out = c;
  return 0;
}

extern "C" int chebhip_comm_create_callback(int nranks, int rank, chebhip_exchangev_fn xfn, chebhip_reduce_fn rfn, void *ctx, chebhip_comm **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks || (!xfn && nranks > 1)) return chebhip_fail(CHEBHIP_ERR_ARG, "bad argument");
  chebhip_comm *c = new (std::nothrow) chebhip_comm;
  if (!c) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  c->kind = KIND_CALLBACK; c->G = nranks; c->rank = rank; c->xfn = xfn; c->rfn = rfn; c->ctx = ctx;
  *out = c;
  return 0;
}

// No wire at all: a rank's own blocks are copied, nothing else moves, reductions leave the values as they are.  For
// timing the compute side of ONE rank of an N-rank partition on one GPU (bench.py "dist_rank_compute"); results are
// meaningless for N > 1.
extern "C" int chebhip_comm_create_null(int nranks, int rank, chebhip_comm **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return chebhip_fail(CHEBHIP_ERR_ARG, "bad argument");
  chebhip_comm *c = new (std::nothrow) chebhip_comm;
  if (!c) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  c->kind = KIND_NULL; c->G = nranks; c->rank = rank;
  *out = c;
  return 0;
}

// The direct route for one process per GPU: rendezvous through `g`, everything else through `inner` (a communicator of the same
// ranks on a message transport; NULL is allowed for one rank).
extern "C" int chebhip_comm_create_ipc(chebhip_ipc_group *g, chebhip_comm *inner, chebhip_comm **out) {
  if (!out) return chebhip_fail(CHEBHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!g || !g->S) return chebhip_fail(CHEBHIP_ERR_ARG, "bad group");
  if (inner ? (inner->G != g->G || inner->rank != g->rank || inner->kind == KIND_IPC) : g->G > 1)
    return chebhip_fail(CHEBHIP_ERR_ARG, "the message transport under an IPC communicator must be a communicator of the same ranks");
  chebhip_comm *c = new (std::nothrow) chebhip_comm;
  if (!c) return chebhip_fail(CHEBHIP_ERR_MEMORY, "out of host memory");
  c->kind = KIND_IPC; c->G = g->G; c->rank = g->rank; c->ig = g; c->inner = inner;
  *out = c;
  return 0;
}

// NULL transport only: from now on peer r's k-th posted array is arrays[r] (NULL entries: the rank's own, as before) instead of the
// rank's own array -- the kernels of the direct route then read and write G distinct arrays, as among real ranks, instead of finding
// the "peers'" rows in the caches because they are its own.  The arrays must be as large as what the drivers post at that index
// (chebhip_dist_mult: k = 0 the slab vector(s), k = 1 the result array of the same size); contents are the caller's business.
extern "C" int chebhip_comm_null_set_shadow(chebhip_comm *c, int k, const double *const *arrays) {
  if (!c || c->kind != KIND_NULL || k < 0 || k >= chebhip::COMM_NPTR) return chebhip_fail(CHEBHIP_ERR_ARG, "chebhip_comm_null_set_shadow: a NULL-transport communicator and k = 0 .. %d", chebhip::COMM_NPTR - 1);
  for (int r = 0; r < c->G; r++) c->shadow[k][r] = arrays ? arrays[r] : nullptr;
  return 0;
}

extern "C" int chebhip_comm_destroy(chebhip_comm *c) {
  if (!c) return 0;
  if (c->kind == KIND_LOCAL && c->lg) {
    std::lock_guard<std::mutex> lk(c->lg->mu);
    c->lg->slot[c->rank].bound = false;          // (the slot's events stay with the group: peers may still be waiting on them)
  }
  if (c->scratch) (void)hipFree(c->scratch);
  delete c;
  return 0;
}
extern "C" int chebhip_comm_size(const chebhip_comm *c) { return c ? c->G : -1; }
extern "C" int chebhip_comm_rank(const chebhip_comm *c) { return c ? c->rank : -1; }

namespace chebhip {
int comm_size(const chebhip_comm *c) { return c ? c->G : 1; }
int comm_rank(const chebhip_comm *c) { return c ? c->rank : 0; }
void comm_abort(chebhip_comm *c) {
  if (c && c->kind == KIND_LOCAL && c->lg) c->lg->abort();
  if (c && c->kind == KIND_IPC && c->ig) c->ig->S->aborted.store(1);
}

bool comm_direct(const chebhip_comm *c) { return c && (c->kind == KIND_LOCAL || c->kind == KIND_NULL || c->kind == KIND_IPC); }
bool comm_is_null(const chebhip_comm *c) { return c && c->kind == KIND_NULL; }
bool comm_overlap_pays(const chebhip_comm *c) {
  if (!c || c->G == 1 || c->kind == KIND_NULL) return false;
  if (c->kind == KIND_IPC) {
    const IpcShared *S = c->ig->S;
    for (int r = 0; r < c->G; r++) if (strncmp(S->slot[r].bus, S->slot[c->rank].bus, sizeof S->slot[r].bus) != 0) return true;
    return false;                                 // every process drives the same device (the rehearsal on one GPU)
  }
  if (c->kind != KIND_LOCAL) return true;
  std::lock_guard<std::mutex> lk(c->lg->mu);
  const int dev = c->lg->slot[c->rank].device;
  for (int r = 0; r < c->G; r++) if (c->lg->slot[r].bound && c->lg->slot[r].device != dev) return true;
  return false;                                   // every rank of the group drives the same device: a side stream only adds dependencies
}
int comm_group_barrier(chebhip_comm *c) { return (c && c->kind == KIND_LOCAL) ? c->lg->barrier() : (c && c->kind == KIND_IPC) ? c->ig->barrier() : 0; }

// IPC: "everything this stream has done so far is complete" as a number the peers' streams can wait for.  ONE counter per rank for
// all event slots: the stream writes the numbers in order, so "the counter has reached the number of slot k's latest post" says that
// post has executed.  The event in front of the write is recorded for its system-scope release (hipEventRecord without
// hipEventDisableSystemFence): what the stream's kernels wrote is visible to other devices before the number is.
// A MARK ("my reads of the peers' arrays end here") only takes its number; the stream write (3.8 us of host time for the pair of
// calls) is left to the next post on the same stream, which is the next thing a driver does or the first thing of its next call --
// no peer waits for a mark before this rank has passed the barrier of a later rendezvous, whose post writes a larger number.
static int ipc_flush(chebhip_ipc_group *g, hipStream_t st, unsigned long long seq) {
  hipError_t e = hipEventRecord(g->fence, st);
  if (e == hipSuccess) e = hipStreamWriteValue64(st, g->dflag(g->rank), (uint64_t)seq, 0);
  if (e != hipSuccess) { g->S->aborted.store(1); return chebhip_fail(CHEBHIP_ERR_DEVICE, "ipc post: %s", hipGetErrorString(e)); }
  return 0;
}
static int ipc_post(chebhip_ipc_group *g, int slot, hipStream_t st, bool mark) {
  IpcSlot &me = g->S->slot[g->rank];
  if (g->pending && g->pending_st != st) {
    // the pending mark belongs to another stream: written there, and waited for, so that the counter never runs backwards (rare path)
    int rc = ipc_flush(g, g->pending_st, me.seq); if (rc) return rc;
    if (hipStreamSynchronize(g->pending_st) != hipSuccess) { g->S->aborted.store(1); return chebhip_fail(CHEBHIP_ERR_DEVICE, "ipc post: stream synchronisation failed"); }
    g->pending = false;
  }
  me.posted[slot] = ++me.seq;
  if (mark) { g->pending = true; g->pending_st = st; return 0; }
  g->pending = false;
  return ipc_flush(g, st, me.seq);
}

int comm_mark(chebhip_comm *c, int slot, hipStream_t st) {
  if (!c || (c->kind != KIND_LOCAL && c->kind != KIND_IPC)) return 0;
  if (slot < 0 || slot >= COMM_NEV) return chebhip_fail(CHEBHIP_ERR_ARG, "comm_mark: bad slot");
  if (c->kind == KIND_IPC) return ipc_post(c->ig, slot, st, true);
  hipError_t e = hipEventRecord(c->lg->slot[c->rank].xev[slot], st);
  if (e != hipSuccess) { c->lg->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(e)); }
  return 0;
}

int comm_rendezvous(chebhip_comm *c, const double *const *ptrs, int n, int slot, int wait_slot, hipStream_t st, PeerView *out) {
  if (!c || !out || n < 0 || n > COMM_NPTR || slot < 0 || slot >= COMM_NEV || wait_slot >= COMM_NEV) return chebhip_fail(CHEBHIP_ERR_ARG, "comm_rendezvous: bad argument");
  if (c->kind == KIND_NULL) {                    // one rank stands for all of them: every peer's arrays are this rank's own ...
    for (int r = 0; r < c->G; r++) for (int k = 0; k < COMM_NPTR; k++) {
      out->ptr[r][k] = k < n ? ptrs[k] : nullptr;
      if (r != c->rank && out->ptr[r][k] && c->shadow[k][r]) out->ptr[r][k] = c->shadow[k][r];   // ... or arrays of their own that the host set aside
    }
    return 0;
  }
  if (c->kind == KIND_IPC) {
    chebhip_ipc_group *g = c->ig;
    IpcShared *S = g->S;
    IpcSlot &me = S->slot[c->rank];
    int rc = ipc_post(g, slot, st, false); if (rc) return rc;            // my arrays are complete at this point of my stream
    const int tb = (int)(me.rdv++ & 1);                                  // (alternating tables: see the LOCAL rendezvous below)
    for (int k = 0; k < COMM_NPTR && !rc; k++) rc = g->publish(k < n ? ptrs[k] : nullptr, &me.xptr[tb][k]);
    me.want[tb] = me.posted[slot];                                       // (the post above is this rank's latest: wait_slot's number is smaller)
    if (rc) { S->aborted.store(1); return rc; }
    if ((rc = g->barrier())) return rc;                                  // every rank has posted and published
    IpcWait wa;
    for (int r = 0; r < MAXR; r++) { wa.flag[r] = nullptr; wa.want[r] = 0; }
    for (int r = 0; r < c->G; r++) {
      if (r == c->rank) { for (int k = 0; k < COMM_NPTR; k++) out->ptr[r][k] = k < n ? ptrs[k] : nullptr; continue; }
      const IpcSlot &ps = S->slot[r];
      for (int k = 0; k < COMM_NPTR && !rc; k++) rc = g->translate(r, ps.xptr[tb][k], &out->ptr[r][k]);
      if (rc) { S->aborted.store(1); return rc; }
      wa.flag[r] = g->dflag(r); wa.want[r] = ps.want[tb];
    }
    if (c->G > 1) {
      int *dab = (int *)(g->dS + ((char *)&S->aborted - (char *)S));
      hipLaunchKernelGGL(k_ipc_wait, dim3(8), dim3(64), 0, st, wa, c->G, c->rank, (unsigned long long)(g->timeout_s * 1e8), dab);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { S->aborted.store(1); return chebhip_fail(CHEBHIP_ERR_DEVICE, "k_ipc_wait: %s", hipGetErrorString(e)); }
    }
    return 0;
  }
  if (c->kind != KIND_LOCAL) return chebhip_fail(CHEBHIP_ERR_ARG, "comm_rendezvous: the transport has no directly addressable peers");
  chebhip_local_group *g = c->lg;
  auto &me = g->slot[c->rank];
  hipError_t e = hipEventRecord(me.xev[slot], st);                     // my arrays are complete at this point of my stream
  if (e != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(e)); }
  // ONE host barrier per rendezvous: the pointer tables alternate (a rank that is already posting rendezvous n + 1 writes the other
  // table than the one a slow rank still reads for n, and nobody reaches n + 2 before everybody has passed the barrier of n + 1);
  // an event slot is re-recorded two rendezvous later at the earliest, after every peer has enqueued its wait on this record.
  const int tb = (int)(me.rdv++ & 1);
  for (int k = 0; k < COMM_NPTR; k++) me.xptr[tb][k] = k < n ? ptrs[k] : nullptr;
  int rc = g->barrier(); if (rc) return rc;                            // every rank has posted and recorded
  for (int r = 0; r < g->G; r++) {
    for (int k = 0; k < COMM_NPTR; k++) out->ptr[r][k] = g->slot[r].xptr[tb][k];
    if (r == c->rank) continue;
    e = hipStreamWaitEvent(st, g->slot[r].xev[slot], 0);
    if (e == hipSuccess && wait_slot >= 0) e = hipStreamWaitEvent(st, g->slot[r].xev[wait_slot], 0);
    if (e != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipStreamWaitEvent: %s", hipGetErrorString(e)); }
  }
  return 0;
}

static int self_copies(const chebhip_comm *c, const XSeg *segs, int nseg, hipStream_t st) {
  for (int i = 0; i < nseg; i++)
    if (segs[i].peer == c->rank && segs[i].nrecv > 0) {
      if (segs[i].nsend != segs[i].nrecv) return chebhip_fail(CHEBHIP_ERR_ARG, "exchange: a rank's own segment has unequal send / receive counts");
      CHIPCHK(hipMemcpyAsync(segs[i].recv, segs[i].send, (size_t)segs[i].nrecv * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
  return 0;
}

static int exchange_rccl(chebhip_comm *c, const XSeg *segs, int nseg, hipStream_t st) {
  // the own blocks first: nothing that can fail sits between ncclGroupStart and ncclGroupEnd except RCCL itself.
  // option "rccl_self_messages" (one-rank smoke runs): the own block goes through ncclSend / ncclRecv as well.
  const bool self_rccl = chebhip::opt(chebhip::OPT_RCCL_SELF_MESSAGES) != 0;
  if (!self_rccl) { int rc = self_copies(c, segs, nseg, st); if (rc) return rc; }
  bool any = false;
  for (int i = 0; i < nseg; i++) if ((segs[i].peer != c->rank || self_rccl) && (segs[i].nsend > 0 || segs[i].nrecv > 0)) any = true;
  if (!any) return 0;
  int rc = g_rccl.GroupStart(); if (rc) return rccl_fail("ncclGroupStart", rc);
  for (int i = 0; i < nseg; i++) {
    const XSeg &s = segs[i];
    if (s.peer == c->rank && !self_rccl) continue;
    if (s.nsend > 0 && (rc = g_rccl.Send(s.send, (size_t)s.nsend, NCCL_DOUBLE, s.peer, c->nccl, st))) { g_rccl.GroupEnd(); return rccl_fail("ncclSend", rc); }
    if (s.nrecv > 0 && (rc = g_rccl.Recv(s.recv, (size_t)s.nrecv, NCCL_DOUBLE, s.peer, c->nccl, st))) { g_rccl.GroupEnd(); return rccl_fail("ncclRecv", rc); }
  }
  rc = g_rccl.GroupEnd(); if (rc) return rccl_fail("ncclGroupEnd", rc);
  return 0;
}

static int exchange_callback(chebhip_comm *c, const XSeg *segs, int nseg, hipStream_t st) {
  int rc = self_copies(c, segs, nseg, st); if (rc) return rc;
  std::vector<int> peers; std::vector<const double *> sp; std::vector<double *> rp; std::vector<long> sc, rcn;
  for (int i = 0; i < nseg; i++) if (segs[i].peer != c->rank) {
    peers.push_back(segs[i].peer); sp.push_back(segs[i].send); sc.push_back(segs[i].nsend); rp.push_back(segs[i].recv); rcn.push_back(segs[i].nrecv);
  }
  if (peers.empty()) return 0;
  if (!c->xfn) return chebhip_fail(CHEBHIP_ERR_ARG, "exchange: no transport");
  return c->xfn(c->ctx, (int)peers.size(), peers.data(), sp.data(), sc.data(), rp.data(), rcn.data(), (void *)st);
}

// every rank pulls: its k-th segment towards peer s is filled from the k-th segment s addresses to it
static int exchange_local(chebhip_comm *c, const XSeg *segs, int nseg, hipStream_t st) {
  chebhip_local_group *g = c->lg;
  auto &me = g->slot[c->rank];
  hipError_t e = hipEventRecord(me.ready, st);                         // my send buffers are complete at this point of my stream
  if (e != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(e)); }
  me.segs = segs; me.nseg = nseg;
  int rc = g->barrier(); if (rc) return rc;                            // every rank has posted and recorded
  int fail = 0;
  std::vector<int> taken(g->G, 0);                                     // how many of my segments towards s have been served
  for (int s = 0; s < g->G && !fail; s++) {
    bool waited = false;
    for (int i = 0; i < nseg && !fail; i++) {
      if (segs[i].peer != s) continue;
      const int k = taken[s]++;
      if (segs[i].nrecv <= 0) continue;
      const auto &ps = g->slot[s];
      const XSeg *match = nullptr; int seen = 0;
      for (int j = 0; j < ps.nseg; j++) if (ps.segs[j].peer == c->rank && seen++ == k) { match = &ps.segs[j]; break; }
      if (!match || match->nsend != segs[i].nrecv) { fail = chebhip_fail(CHEBHIP_ERR_ARG, "local exchange: rank %d has no matching segment %d for rank %d", s, k, c->rank); break; }
      if (!waited && s != c->rank) { e = hipStreamWaitEvent(st, ps.ready, 0); waited = true; if (e != hipSuccess) { fail = chebhip_fail(CHEBHIP_ERR_DEVICE, "hipStreamWaitEvent: %s", hipGetErrorString(e)); break; } }
      e = hipMemcpyAsync(segs[i].recv, match->send, (size_t)segs[i].nrecv * sizeof(double), hipMemcpyDefault, st);
      if (e != hipSuccess) fail = chebhip_fail(CHEBHIP_ERR_DEVICE, "hipMemcpyAsync: %s", hipGetErrorString(e));
    }
  }
  if (!fail) { e = hipEventRecord(me.done, st); if (e != hipSuccess) fail = chebhip_fail(CHEBHIP_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(e)); }
  if (fail) { g->abort(); return fail; }
  rc = g->barrier(); if (rc) return rc;                                // every rank has enqueued its pulls
  for (int s = 0; s < g->G; s++)                                       // my send buffers may be rewritten once the peers have read them
    if (s != c->rank) { e = hipStreamWaitEvent(st, g->slot[s].done, 0); if (e != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipStreamWaitEvent: %s", hipGetErrorString(e)); } }
  return 0;      // the peers read my list only between the two barriers: it may go once this call returns
}

int comm_exchange(chebhip_comm *c, const XSeg *segs, int nseg, hipStream_t st) {
  if (!c) {   // one rank, no transport: own blocks only
    chebhip_comm one; one.G = 1; one.rank = 0;
    for (int i = 0; i < nseg; i++) if (segs[i].peer != 0) return chebhip_fail(CHEBHIP_ERR_ARG, "exchange: no transport set");
    return self_copies(&one, segs, nseg, st);
  }
  switch (c->kind) {
    case KIND_RCCL: return exchange_rccl(c, segs, nseg, st);
    case KIND_LOCAL: return exchange_local(c, segs, nseg, st);
    case KIND_CALLBACK: return exchange_callback(c, segs, nseg, st);
    case KIND_NULL: return self_copies(c, segs, nseg, st);
    case KIND_IPC: return comm_exchange(c->inner, segs, nseg, st);      // (inner == NULL: one rank, own blocks only)
  }
  return chebhip_fail(CHEBHIP_ERR_ARG, "exchange: bad communicator");
}
}  // namespace chebhip

// chebhip_reduce_fn over any communicator (ctx = the chebhip_comm): the all-reduce of the few doubles a Krylov iteration
// needs, for chebhip_fgmres_set_reduce / stokes_op_set_inner_reduce.  Same result bits on every rank.
extern "C" int chebhip_comm_reduce(void *comm, double *vals_dev, int count, void *stream) {
  chebhip_comm *c = (chebhip_comm *)comm;
  if (!c || !vals_dev || count < 0) return chebhip_fail(CHEBHIP_ERR_ARG, "bad argument");
  if (count == 0 || c->G == 1 || c->kind == KIND_NULL) return 0;
  if (c->kind == KIND_IPC) return chebhip_comm_reduce(c->inner, vals_dev, count, stream);
  hipStream_t st = (hipStream_t)stream;
  if (c->kind == KIND_RCCL) {
    int rc = g_rccl.AllReduce(vals_dev, vals_dev, (size_t)count, NCCL_DOUBLE, NCCL_SUM, c->nccl, st);
    return rc ? rccl_fail("ncclAllReduce", rc) : 0;
  }
  if (c->kind == KIND_CALLBACK) {
    if (!c->rfn) return chebhip_fail(CHEBHIP_ERR_ARG, "reduce: the communicator has no reduction");
    return c->rfn(c->ctx, vals_dev, count, stream);
  }
  if (count > MAXR) return chebhip_fail(CHEBHIP_ERR_ARG, "local reduce: at most %d values", MAXR);
  chebhip_local_group *g = c->lg;
  auto &me = g->slot[c->rank];
  hipError_t e = hipEventRecord(me.ready, st);
  if (e != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(e)); }
  me.vals = vals_dev;
  int rc = g->barrier(); if (rc) return rc;
  RedPtrs in;
  for (int r = 0; r < g->G; r++) {
    in.p[r] = g->slot[r].vals;
    if (r != c->rank && (e = hipStreamWaitEvent(st, g->slot[r].ready, 0)) != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipStreamWaitEvent: %s", hipGetErrorString(e)); }
  }
  hipLaunchKernelGGL(k_local_reduce, dim3(1), dim3(64), 0, st, in, g->G, count, c->scratch);
  if ((e = hipGetLastError()) != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "k_local_reduce: %s", hipGetErrorString(e)); }
  e = hipEventRecord(me.done, st);
  if (e != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(e)); }
  rc = g->barrier(); if (rc) return rc;
  for (int r = 0; r < g->G; r++)
    if (r != c->rank && (e = hipStreamWaitEvent(st, g->slot[r].done, 0)) != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipStreamWaitEvent: %s", hipGetErrorString(e)); }
  e = hipMemcpyAsync(vals_dev, c->scratch, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, st);   // everybody has read my values
  if (e != hipSuccess) { g->abort(); return chebhip_fail(CHEBHIP_ERR_DEVICE, "hipMemcpyAsync: %s", hipGetErrorString(e)); }
  return 0;
}

// ---- communicator helpers for hosts that do not bring their own ncclComm_t ------------------------------------
extern "C" int chebhip_rccl_unique_id(void *id128) {
  if (!id128) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (!rccl_ready()) return chebhip_fail(CHEBHIP_ERR_DEVICE, "librccl.so could not be loaded");
  int rc = g_rccl.GetUniqueId(id128); if (rc) return rccl_fail("ncclGetUniqueId", rc);
  return 0;
}
extern "C" int chebhip_rccl_comm_create(int nranks, int rank, const void *id128, void **comm) {
  if (!id128 || !comm) return chebhip_fail(CHEBHIP_ERR_ARG, "NULL argument");
  if (!rccl_ready()) return chebhip_fail(CHEBHIP_ERR_DEVICE, "librccl.so could not be loaded");
  Id128 id; memcpy(&id, id128, sizeof id);
  int rc = g_rccl.CommInitRank(comm, nranks, id, rank); if (rc) return rccl_fail("ncclCommInitRank", rc);
  return 0;
}
extern "C" int chebhip_rccl_comm_destroy(void *comm) {
  if (!comm) return 0;
  if (!rccl_ready()) return chebhip_fail(CHEBHIP_ERR_DEVICE, "librccl.so could not be loaded");
  int rc = g_rccl.CommDestroy(comm); if (rc) return rccl_fail("ncclCommDestroy", rc);
  return 0;
}
// chebhip_reduce_fn over a bare RCCL communicator (ctx = the ncclComm_t)
extern "C" int chebhip_rccl_reduce(void *comm, double *vals_dev, int count, void *stream) {
  if (!comm || !vals_dev || count < 0) return chebhip_fail(CHEBHIP_ERR_ARG, "bad argument");
  if (!rccl_ready()) return chebhip_fail(CHEBHIP_ERR_DEVICE, "librccl.so could not be loaded");
  int rc = g_rccl.AllReduce(vals_dev, vals_dev, (size_t)count, NCCL_DOUBLE, NCCL_SUM, comm, (hipStream_t)stream);
  if (rc) return rccl_fail("ncclAllReduce", rc);
  return 0;
}
