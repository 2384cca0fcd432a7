// options.cpp -- the run-time options of the library (chebhip_set_option, include/chebhip.h): named integer switches,
// process-wide, read where they apply.  Nothing in the library reads the environment.  Kept free of HIP so that the host
// tests can link it beside diffmat.cpp.
#include "sweep.h"
#include <atomic>
#include <cstring>
#include <mutex>

namespace chebhip {
namespace {
struct OptDesc { const char *name; int def; };
const OptDesc g_opt_desc[OPT_COUNT] = {
  {"general_kernels", 0}, {"separate_launches", 0}, {"vendor_gemm", 0}, {"no_raw_transforms", 0}, {"equal_shares", 0}, {"force_gemm", 0},
  {"stokes_single_stream", 0}, {"eta_from_memory", 0}, {"gather_pass", 0}, {"rccl_self_messages", 0}, {"local_timeout_s", 120}, {"full_stress_storage", 0}, {"dist_single_stream", 0}, {"long_lines_gemm", 0}, {"pressure_passes", 0}, {"general_viscous", 0}, {"poisson_launches", 0}, {"dist_exact_order", 0}, {"fdm_passes", 0}, {"saddle_node_major", 0}, {"stokes_z_separate", 0}, {"fdm_z_separate", 0}, {"stokes_pressure_stream", 0}, {"krylov_exact_norm", 0}, {"stokes_pressure_sweeps", 0}, {"dist_packed_exchange", 0},
};
std::atomic<int> g_opt_val[OPT_COUNT];
std::once_flag g_opt_once;
void opt_init() { std::call_once(g_opt_once, [] { for (int i = 0; i < OPT_COUNT; i++) g_opt_val[i].store(g_opt_desc[i].def); }); }
}  // namespace

int opt(int id) { opt_init(); return (id >= 0 && id < OPT_COUNT) ? g_opt_val[id].load(std::memory_order_relaxed) : 0; }
void opt_set(int id, int value) { opt_init(); if (id >= 0 && id < OPT_COUNT) g_opt_val[id].store(value); }
const char *opt_name(int id) { return (id >= 0 && id < OPT_COUNT) ? g_opt_desc[id].name : ""; }
int opt_find(const char *name) { for (int i = 0; i < OPT_COUNT; i++) if (!strcmp(name, g_opt_desc[i].name)) return i; return -1; }
}  // namespace chebhip
