// sweep.hip -- the hot kernel: y = D x along one dimension of a row-major tensor, f64, gfx950.
//
// One launch replaces one ChebMult (chebyshev.c:142-199), with the neighbouring
// vector passes of the PDE callbacks (VecScatter GL/LG, the pointwise flux loop,
// VecZeroEntries + VecAXPY; elliptic.C:305-337) folded into its loads and stores.
//
// Structure (P <= 256; see DESIGN.md for the derivation and roofline):
//  - the two parity halves of D (ME, MO: H x H, H = ceil(P/2)) live in REGISTERS for the
//    lifetime of a workgroup: wave w of 8 owns the 16 output rows of m-tile (w % MTP) and
//    keeps their 2*KS MFMA operand fragments (128 VGPRs at P = 256);
//  - workgroups are persistent (grid = #CUs) and walk tiles of NT lines; the lines of tile
//    t+1 are loaded from HBM into registers while the MFMA chains of tile t run, then split
//    into e = x_j + x_{n-j}, o = x_j - x_{n-j} and parked in the other half of a
//    double-buffered LDS tile (2 x 64 KiB);
//  - each wave runs v_mfma_f64_16x16x4_f64 chains over the LDS tile and stores
//    y_i = a_i + b_i and y_{n-i} = b_i - a_i straight from the accumulators.
//  - two tilings: COLFAST (inner stride >= 16: neighbouring lanes = neighbouring lines,
//    matrix is the A operand) and JFAST (inner stride small, e.g. 1 or the d interleaved
//    Stokes components: neighbouring lanes = neighbouring points of a line, matrix is the B
//    operand so that the 16 lanes of an accumulator row are 16 consecutive outputs of a line).
//
// A wave issues about one vector instruction per 4-5 cycles, so everything that is not an
// MFMA is kept off the per-element path: offsets are 32-bit and tile-invariant, the mode
// switches are hoisted around the unrolled loops, LDS fragment reads run one group ahead.
#include "sweep.h"
#include <map>
#include <mutex>
#include <utility>
#include <dlfcn.h>
#include <atomic>
#include <type_traits>
#include <cstdlib>

namespace chebhip {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned u32;

static std::atomic<long> g_launches{0};
long sweep_launch_count() { return g_launches.load(); }
int sweep_num_cus(hipError_t *err) {
  static std::atomic<int> cache[64];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { *err = e; return 0; }
  if (dev >= 0 && dev < 64) { const int c = cache[dev].load(std::memory_order_relaxed); if (c > 0) { *err = hipSuccess; return c; } }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) { *err = e; return 0; }
  const int n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (dev >= 0 && dev < 64) cache[dev].store(n, std::memory_order_relaxed);
  *err = hipSuccess;
  return n;
}
void sweep_note_launch() { g_launches.fetch_add(1); }

template <int M> using mode_c = std::integral_constant<int, M>;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0): every
// wave would wait at each tile boundary for its own prefetch loads and result stores, which
// serialises the HBM stream with the MFMA phases.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One input element at local element offset `a` (point j of its line; gb = global index of the
// line's j = 1 node in the interior vector, or -1).
// Branch-free fetch.  For the single-array modes there is NO use of the loaded value here: an
// invalid slot reads from a zero word (p.zero) instead of being masked afterwards -- arithmetic
// on the value would make the compiler wait for the load right after issuing it and expose the
// whole HBM latency.  The flux modes combine several arrays and so do wait at issue.
template <int MODE>
__device__ __forceinline__ double fetch_in(const SweepParams &p, u32 a, int j, int gb, bool ok) {
  if (MODE == IN_PLAIN) return *(ok ? p.in0 + a : p.zero);
  if (MODE == IN_GATHER) {
    ok = ok && gb >= 0 && j >= 1 && j <= p.P - 2;
    return *(ok ? p.in0 + ((long)gb + (long)(j - 1) * p.gstride) : p.zero);
  }
  const u32 b = ok ? a : 0u;
  double v;
  if (MODE == IN_FLUX_ETA) v = p.in1[b] * p.in0[b];
  else v = p.in1[b] * p.in0[b] + p.in2[b] * p.in3[b] * p.in4[b];
  return ok ? v : 0.0;
}

template <int KS, bool JFAST>
__global__ __launch_bounds__(512) void cheb_sweep_kernel(const SweepParams p) {
  constexpr int MTP = KS / 4;                      // m-tiles of 16 output rows (padded)
  constexpr int NG = 8 / MTP;                      // wave groups along the line index
  constexpr int HP = 4 * KS;                       // padded half length
  constexpr int NSUB = (KS >= 16) ? 2 : 1;         // 16-line sub-tiles per wave per tile
  constexpr int NT = 16 * NG * NSUB;               // lines per tile: 32, 64, 64, 128
  constexpr int LDJ = HP + 1;                      // JFAST row pitch: ODD -> conflict-free operand reads (round 6: tools/lds_probe.hip; == 2 mod 32 was a 2-way conflict)
  constexpr int LDS_ELEMS = JFAST ? NT * LDJ : HP * NT;
  constexpr int ITEMS = HP * NT / 512;             // (j-pair, line) slots per thread per tile
  constexpr int CH = ITEMS / NSUB;                 // slots per chunk (one chunk rides under one sub-tile)
  constexpr int QSTEP = JFAST ? 512 / HP : 512 / NT;  // line step (JFAST) / j-pair step (COLFAST) between slots
  __shared__ double smem[4 * LDS_ELEMS];           // two buffers of (E, O)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mt = w % MTP, ng = w / MTP;
  const int kq = lane >> 4, l16 = lane & 15;
  const int nn = p.P - 1, H = p.H;
  const u32 inner = p.inner, ncols = p.ncols;
  const u32 lineLen = (u32)p.P * inner;
  const bool need_g = (p.in_mode == IN_GATHER) || (p.out_mode == OUT_ACC_SCATTER);

  // Matrix fragments -> registers (coalesced 512 B per wave load).
  double ae[KS], ao[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) {
    ae[s] = p.fragE[((long)(mt * KS + s)) * 64 + lane];
    ao[s] = p.fragO[((long)(mt * KS + s)) * 64 + lane];
  }
  // Retire the fragment loads here: otherwise the compiler's wait-count pass keeps them "pending"
  // around the tile loop and threads vmcnt(62..0) waits through the MFMA chain, which would also
  // drain the prefetch of the next tile mid-chain.  0x0F70 = vmcnt(0) only.
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // ---- tile geometry -------------------------------------------------------------------------
  // COLFAST: a tile is NT consecutive lines q0..q0+NT-1 of ONE outer block o (tiles never
  //          straddle blocks, so a line's offset is uniform base + lane term: no per-lane divide).
  // JFAST  : a tile is NT consecutive lines c of the flattened (outer, inner) line index.
  const u32 tpo = JFAST ? 1u : (inner + NT - 1) / NT;    // tiles per outer block (COLFAST)
  // XCD-aware tile walk: workgroups b and b+8 share an XCD (and its L2).  Give each XCD one
  // contiguous range of tiles and let its CUs take neighbouring tiles at the same time, so that a
  // 128-B line straddled by two neighbouring row pieces is fetched from HBM once, not once per XCD.
  const u32 nxcd = (gridDim.x % 8 == 0) ? 8u : 1u;
  const u32 t_per = (p.ntiles + nxcd - 1) / nxcd;
  const u32 t_lo = (blockIdx.x % nxcd) * t_per;
  const u32 t_hi = (t_lo + t_per < p.ntiles) ? t_lo + t_per : p.ntiles;
  const u32 t_step = gridDim.x / nxcd;

  // loader slots of this thread
  const int ld_n = JFAST ? tid / HP : tid % NT;          // first line slot (JFAST) / line (COLFAST)
  const int ld_j = JFAST ? tid % HP : tid / NT;          // j-pair (JFAST) / first j-pair slot (COLFAST)
  const int ld_lds0 = JFAST ? ld_n * LDJ + ld_j : ld_j * NT + (ld_n ^ ((ld_j & 1) << 4));
  constexpr int LDS_QSTEP = JFAST ? QSTEP * LDJ : QSTEP * NT;   // QSTEP even -> swizzle parity unchanged

  double rj[CH], rm[CH];                   // x_j and x_{n-j} of the chunk in flight

  // Issue the loads of chunk `chunk` of tile `tile` into rj/rm.
  auto issue_loads = [&](auto MODE, u32 tile, int chunk) {
    constexpr int IM = decltype(MODE)::value;
    if (!JFAST) {
      const u32 o = tile / tpo, q0 = (tile - o * tpo) * NT;
      const u32 q = q0 + ld_n;
      const bool cv = q < inner;
      const u32 base = o * lineLen + q;
      int gb = -1;
      if (IM == IN_GATHER) gb = p.gcol[cv ? o * inner + q : 0u];
      // running offsets, made opaque so that the optimiser does not hoist one precomputed
      // address pair per slot out of the tile loop (that costs ~60 VGPRs and spills)
      int jp = ld_j + chunk * CH * QSTEP;
      u32 rel = (u32)jp * inner;
      const u32 top = base + (u32)nn * inner;
      asm volatile("" : "+v"(rel), "+v"(jp));
#pragma unroll
      for (int s = 0; s < CH; s++, jp += QSTEP, rel += QSTEP * inner) {
        const int jm = nn - jp;
        const bool ok = cv && jp < H;
        rj[s] = fetch_in<IM>(p, base + rel, jp, gb, ok);
        rm[s] = fetch_in<IM>(p, top - rel, jm, gb, ok && jm != jp);
      }
    } else {
      const int jp = ld_j, jm = nn - jp;
#pragma unroll
      for (int s = 0; s < CH; s++) {
        const u32 c = tile * NT + ld_n + (chunk * CH + s) * QSTEP;
        const bool ok = c < ncols && jp < H;
        const u32 cc = ok ? c : 0u;
        const u32 base = (inner == 1) ? cc * lineLen : (cc / inner) * lineLen + (cc % inner);
        const int gb = (IM == IN_GATHER) ? p.gcol[cc] : -1;
        rj[s] = fetch_in<IM>(p, base + (u32)jp * inner, jp, gb, ok);
        rm[s] = fetch_in<IM>(p, base + (u32)jm * inner, jm, gb, ok && jm != jp);
      }
    }
  };
  auto issue_loads_any = [&](u32 tile, int chunk) {
    switch (p.in_mode) {
      case IN_PLAIN: issue_loads(mode_c<IN_PLAIN>{}, tile, chunk); break;
      case IN_GATHER: issue_loads(mode_c<IN_GATHER>{}, tile, chunk); break;
      case IN_FLUX_ETA: issue_loads(mode_c<IN_FLUX_ETA>{}, tile, chunk); break;
      default: issue_loads(mode_c<IN_FLUX_FULL>{}, tile, chunk); break;
    }
  };

  // Parity split of the chunk in registers -> LDS buffer `buf`.  For the self-paired middle
  // point (2j == n) xm was left 0, so e = x_j; o is forced to 0 there.
  auto park_chunk = [&](int buf, int chunk) {
    double *dE = smem + buf * (2 * LDS_ELEMS), *dO = dE + LDS_ELEMS;
    const bool mid = JFAST && (2 * ld_j == nn);
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const int idx = ld_lds0 + (chunk * CH + s) * LDS_QSTEP;
      const bool m2 = JFAST ? mid : (2 * (ld_j + (chunk * CH + s) * QSTEP) == nn);
      dE[idx] = rj[s] + rm[s];
      dO[idx] = m2 ? 0.0 : rj[s] - rm[s];
    }
  };

  // compute-side invariants of this lane
  const int i0 = mt * 16 + (JFAST ? l16 : kq);        // output row of accumulator element r: i0 + (JFAST ? 0 : 4r)

  u32 tile = t_lo + blockIdx.x / nxcd;
  if (tile < t_hi) {
#pragma unroll 1
    for (int ch = 0; ch < NSUB; ch++) { issue_loads_any(tile, ch); park_chunk(0, ch); }
  }
  lds_barrier();
  int cur = 0;
  for (; tile < t_hi; tile += t_step) {
    const u32 nxt = tile + t_step;
    const bool has_next = nxt < t_hi;
    const double *sE = smem + cur * (2 * LDS_ELEMS), *sO = sE + LDS_ELEMS;
    // tile base (COLFAST): uniform
    const u32 t_o = tile / tpo, t_q0 = (tile - t_o * tpo) * NT;
#pragma unroll 1
    for (int sub = 0; sub < NSUB; sub++) {
      if (has_next) issue_loads_any(nxt, sub);             // in flight during this sub-tile's MFMA chain
      const int nb = (ng * NSUB + sub) * 16;

      // ---- where this lane's 4 accumulator rows go ----
      u32 ob[4];        // element offset of output (line, i = 0)
      bool ov[4];
      int og[4];
      double accv[8];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        bool lv; u32 b; u32 cidx;
        if (!JFAST) {
          const u32 q = t_q0 + nb + l16;
          lv = q < inner; b = t_o * lineLen + q; cidx = t_o * inner + q;
          ov[r] = lv && (i0 + 4 * r < H);
        } else {
          const u32 c = tile * NT + nb + 4 * r + kq;
          lv = c < ncols; cidx = c;
          b = (inner == 1) ? c * lineLen : (c / inner) * lineLen + (c % inner);
          ov[r] = lv && (i0 < H);
        }
        ob[r] = b;
        og[r] = -1;
        accv[2 * r] = 0.0; accv[2 * r + 1] = 0.0;
        if (ov[r]) {
          if (need_g) og[r] = p.gcol[cidx];
          if (p.out_mode != OUT_STORE && p.acc) {          // VecAXPY operand: fetched ahead of the MFMA chain
            const int i = i0 + (JFAST ? 0 : 4 * r);
            accv[2 * r] = p.acc[b + (u32)i * inner];
            accv[2 * r + 1] = p.acc[b + (u32)(nn - i) * inner];
          }
        }
      }

      // ---- MFMA chains; LDS fragment reads run one group (2 k-steps) ahead ----
      v4d ce = {0.0, 0.0, 0.0, 0.0}, co = {0.0, 0.0, 0.0, 0.0};
      const int frag = JFAST ? (nb + l16) * LDJ + kq : kq * NT + ((nb + l16) ^ ((kq & 1) << 4));
      const double *fE = sE + frag, *fO = sO + frag;
      constexpr int KSTR = JFAST ? 4 : 4 * NT;             // LDS stride of one k-step
      {
        double fb[2][4];
        fb[0][0] = fE[0]; fb[0][1] = fE[KSTR]; fb[0][2] = fO[0]; fb[0][3] = fO[KSTR];
#pragma unroll
        for (int g = 0; g < KS / 2; g++) {
          const int cb = g & 1, nbuf = cb ^ 1;
          if (g + 1 < KS / 2) {
            fb[nbuf][0] = fE[(2 * g + 2) * KSTR]; fb[nbuf][1] = fE[(2 * g + 3) * KSTR];
            fb[nbuf][2] = fO[(2 * g + 2) * KSTR]; fb[nbuf][3] = fO[(2 * g + 3) * KSTR];
          }
          if (!JFAST) {   // rows = outputs i, cols = lines
            ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[2 * g], fb[cb][0], ce, 0, 0, 0);
            co = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[2 * g], fb[cb][2], co, 0, 0, 0);
            ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[2 * g + 1], fb[cb][1], ce, 0, 0, 0);
            co = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[2 * g + 1], fb[cb][3], co, 0, 0, 0);
          } else {        // rows = lines, cols = outputs i
            ce = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][0], ae[2 * g], ce, 0, 0, 0);
            co = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][2], ao[2 * g], co, 0, 0, 0);
            ce = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][1], ae[2 * g + 1], ce, 0, 0, 0);
            co = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][3], ao[2 * g + 1], co, 0, 0, 0);
          }
        }
      }

      // Park the prefetched chunk BEFORE issuing this sub-tile's stores: the wait in front of the
      // parity split then covers loads only (vmcnt retires in order; after the stores it would
      // also wait for them to reach memory).
      if (has_next) park_chunk(cur ^ 1, sub);
      // ---- stores: accumulator element r of a lane is row 4r + (lane >> 4), col lane & 15 ----
      if (p.sym) { const v4d t = ce; ce = co; co = t; }   // centro-symmetric matrix: y_{n-i} = a - b, i.e. the roles of a and b swap in (b - a)
      {
        const double alpha = p.alpha;
        if (p.out_mode == OUT_STORE) {
#pragma unroll
          for (int r = 0; r < 4; r++) if (ov[r]) {
            const int i = i0 + (JFAST ? 0 : 4 * r);
            p.out[ob[r] + (u32)i * inner] = alpha * (ce[r] + co[r]);
            if (nn - i != i) p.out[ob[r] + (u32)(nn - i) * inner] = alpha * (co[r] - ce[r]);
          }
        } else if (p.out_mode == OUT_ACC) {
#pragma unroll
          for (int r = 0; r < 4; r++) if (ov[r]) {
            const int i = i0 + (JFAST ? 0 : 4 * r);
            p.out[ob[r] + (u32)i * inner] = accv[2 * r] + alpha * (ce[r] + co[r]);
            if (nn - i != i) p.out[ob[r] + (u32)(nn - i) * inner] = accv[2 * r + 1] + alpha * (co[r] - ce[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) if (ov[r] && og[r] >= 0) {
            const int i = i0 + (JFAST ? 0 : 4 * r);
            const int im = nn - i;
            if (i >= 1 && i <= nn - 1) p.out[(long)og[r] + (long)(i - 1) * p.gstride] = accv[2 * r] + alpha * (ce[r] + co[r]);
            if (im != i && im >= 1 && im <= nn - 1)
              p.out[(long)og[r] + (long)(im - 1) * p.gstride] = accv[2 * r + 1] + alpha * (co[r] - ce[r]);
          }
        }
      }
    }
    lds_barrier();
    cur ^= 1;
  }
}

template <int KS, bool JFAST>
static hipError_t launch_t(const SweepParams &p0, hipStream_t stream) {
  constexpr int MTP = KS / 4, NG = 8 / MTP, NSUB = (KS >= 16) ? 2 : 1, NT = 16 * NG * NSUB;
  SweepParams p = p0;
  if (JFAST) p.ntiles = (p.ncols + NT - 1) / NT;
  else p.ntiles = (p.ncols / p.inner) * ((p.inner + NT - 1) / NT);
  hipError_t cu_err; const int ncu = sweep_num_cus(&cu_err);
  if (cu_err != hipSuccess) return cu_err;
  // persistent: one 512-thread workgroup per CU (the register-resident matrix allows no more)
  const unsigned grid = p.ntiles < (unsigned)ncu ? p.ntiles : (unsigned)ncu;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL((cheb_sweep_kernel<KS, JFAST>), dim3(grid), dim3(512), 0, stream, p);
  g_launches.fetch_add(1);
  return hipGetLastError();
}

}  // namespace chebhip
const double *chebhip_stamp_buf();
int chebhip_stamp_next();
namespace chebhip {

// ---------------------------------------------------------------------------------------------
// Lines of more than 256 points (the reference accepts any extent, chebyshev.c:98).  The matrix no longer
// fits the registers of a workgroup, so this kernel is a plain dense product on the FP64 VALU: LN lines per
// workgroup staged in LDS, D^T streamed from L2 (coalesced over the output index), the same load and store
// modes as cheb_sweep_kernel.  A correctness path, not a tuned one.
template <int LN>
__global__ __launch_bounds__(256) void cheb_sweep_long_kernel(const SweepParams p) {
  extern __shared__ double xs[];                 // [LN][P]
  const int P = p.P;
  const u32 inner = p.inner, ncols = p.ncols;
  const bool need_g = (p.in_mode == IN_GATHER) || (p.out_mode == OUT_ACC_SCATTER);
  for (u32 c0 = blockIdx.x * LN; c0 < ncols; c0 += gridDim.x * LN) {
    for (int t = threadIdx.x; t < LN * P; t += 256) {
      const u32 c = c0 + t / P; const int j = t % P;
      const bool ok = c < ncols;
      const u32 cc = ok ? c : 0u;
      const u32 a = (cc / inner) * (u32)P * inner + (cc % inner) + (u32)j * inner;
      const int gb = need_g ? p.gcol[cc] : -1;
      double v;
      switch (p.in_mode) {
        case IN_PLAIN: v = fetch_in<IN_PLAIN>(p, a, j, gb, ok); break;
        case IN_GATHER: v = fetch_in<IN_GATHER>(p, a, j, gb, ok); break;
        case IN_FLUX_ETA: v = fetch_in<IN_FLUX_ETA>(p, a, j, gb, ok); break;
        default: v = fetch_in<IN_FLUX_FULL>(p, a, j, gb, ok); break;
      }
      xs[t] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += 256) {
      double acc[LN];
#pragma unroll
      for (int l = 0; l < LN; l++) acc[l] = 0.0;
      for (int j = 0; j < P; j++) {
        const double d = p.longDT[(long)j * P + i];
#pragma unroll
        for (int l = 0; l < LN; l++) acc[l] += d * xs[l * P + j];
      }
#pragma unroll
      for (int l = 0; l < LN; l++) {
        const u32 c = c0 + l;
        if (c >= ncols) continue;
        const u32 a = (c / inner) * (u32)P * inner + (c % inner) + (u32)i * inner;
        const double r = p.alpha * acc[l];
        if (p.out_mode == OUT_STORE) p.out[a] = r;
        else if (p.out_mode == OUT_ACC) p.out[a] = p.acc[a] + r;
        else {
          const int gb = p.gcol[c];
          if (gb >= 0 && i >= 1 && i <= P - 2) p.out[(long)gb + (long)(i - 1) * p.gstride] = (p.acc ? p.acc[a] : 0.0) + r;
        }
      }
    }
    __syncthreads();
  }
}

// Plain sweeps (IN_PLAIN, STORE / ACC) of long lines are plain strided-batched DGEMMs: Y_o = D X_o per outer block.
// Behind the options "vendor_gemm" (lines beyond 1024 points) and "long_lines_gemm" / "force_gemm" (A/B routes) they can go
// to rocBLAS, looked up at run time (the copy the process already has, e.g. PyTorch's, else the system one) and never
// linked.  By default no vendor GEMM runs: cheb_sweep_long_kernel above takes every line beyond 1024 points.
namespace {
struct RocblasApi {
  void *lib = nullptr;
  int (*create)(void **) = nullptr;
  int (*set_stream)(void *, hipStream_t) = nullptr;
  int (*dgemm_sb)(void *, int, int, int, int, int, const double *, const double *, int, long long, const double *, int, long long,
                  const double *, double *, int, long long, int) = nullptr;
  bool ok = false;
  std::once_flag once;
  std::mutex mu;
  std::map<std::pair<int, hipStream_t>, void *> handles;    // one handle per (device, stream): a handle is bound to one stream
};
RocblasApi g_rb;

bool rocblas_ready() {
  std::call_once(g_rb.once, [] {
    const char *names[] = {"librocblas.so.5", "librocblas.so.4", "librocblas.so"};
    for (const char *n : names) if (!g_rb.lib) g_rb.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // already in the process?
    for (const char *n : names) if (!g_rb.lib) g_rb.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!g_rb.lib) return;
    g_rb.create = (int (*)(void **))dlsym(g_rb.lib, "rocblas_create_handle");
    g_rb.set_stream = (int (*)(void *, hipStream_t))dlsym(g_rb.lib, "rocblas_set_stream");
    g_rb.dgemm_sb = (decltype(g_rb.dgemm_sb))dlsym(g_rb.lib, "rocblas_dgemm_strided_batched");
    g_rb.ok = g_rb.create && g_rb.set_stream && g_rb.dgemm_sb;
  });
  return g_rb.ok;
}

// The handle of (current device, stream), created on first use.  Two streams of one operator (the Stokes pressure chain
// runs beside the viscous chain) therefore never re-target each other's handle.
void *rocblas_handle_for(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_rb.mu);
  auto key = std::make_pair(dev, stream);
  auto it = g_rb.handles.find(key);
  if (it != g_rb.handles.end()) return it->second;
  void *h = nullptr;
  if (g_rb.create(&h) != 0 || !h) return nullptr;
  if (g_rb.set_stream(h, stream) != 0) return nullptr;
  g_rb.handles[key] = h;
  return h;
}
}  // namespace

static hipError_t launch_long_gemm(const SweepParams &p, hipStream_t stream, bool *done) {
  *done = false;
  // No vendor GEMM on a default path: rocBLAS is touched only when an option asks for it ("vendor_gemm": lines beyond 1024
  // points; "long_lines_gemm" / "force_gemm": the A/B routes of shorter lines)
  if (!opt(OPT_VENDOR_GEMM) && !opt(OPT_LONG_LINES_GEMM) && !opt(OPT_FORCE_GEMM)) return hipSuccess;
  if (p.in_mode != IN_PLAIN || (p.out_mode != OUT_STORE && p.out_mode != OUT_ACC) || !p.longD || !rocblas_ready()) return hipSuccess;
  constexpr int OP_N = 111, OP_T = 112;                     // rocblas_operation_none / _transpose
  if (p.out_mode == OUT_ACC && p.acc != p.out) {
    hipError_t e = hipMemcpyAsync(p.out, p.acc, (size_t)p.ncols * p.P * sizeof(double), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
  }
  const double alpha = p.alpha, beta = (p.out_mode == OUT_ACC) ? 1.0 : 0.0;
  void *handle = rocblas_handle_for(stream);
  if (!handle) return hipErrorUnknown;
  const int P = p.P;
  int st;
  if (p.inner == 1)        // lines contiguous: Y (P x ncols, column-major) = D X
    st = g_rb.dgemm_sb(handle, OP_T, OP_N, P, (int)p.ncols, P, &alpha, p.longD, P, 0, p.in0, P, 0, &beta, p.out, P, 0, 1);
  else {                   // per outer block o: Y_o^T (inner x P, column-major) = X_o^T D^T
    const long long blk = (long long)P * p.inner;
    st = g_rb.dgemm_sb(handle, OP_N, OP_N, (int)p.inner, P, P, &alpha, p.in0, (int)p.inner, blk, p.longD, P, 0, &beta,
                       p.out, (int)p.inner, blk, (int)(p.ncols / p.inner));
  }
  if (st != 0) return hipErrorUnknown;
  sweep_note_launch();
  *done = true;
  return hipSuccess;
}

static hipError_t launch_long(const SweepParams &p, hipStream_t stream) {
  {
    bool done = false;
    hipError_t e = launch_long_gemm(p, stream, &done);
    if (e != hipSuccess || done) return e;
  }
  const int LN = p.P <= 2048 ? 4 : 2;            // LN * P * 8 B of LDS, at most 64 KiB
  unsigned grid = (p.ncols + LN - 1) / LN; if (grid > 2048) grid = 2048;
  if (grid == 0) return hipSuccess;
  const size_t lds = (size_t)LN * p.P * sizeof(double);
  if (LN == 4) hipLaunchKernelGGL((cheb_sweep_long_kernel<4>), dim3(grid), dim3(256), lds, stream, p);
  else hipLaunchKernelGGL((cheb_sweep_long_kernel<2>), dim3(grid), dim3(256), lds, stream, p);
  sweep_note_launch();
  return hipGetLastError();
}

hipError_t sweep_launch(const DiffMat &m, SweepParams p, hipStream_t stream) {
  if (p.in_mode == IN_SUM3) return hipErrorInvalidValue;    // exists in the multi-job launch of short lines only (sweep_vec.hip)
  p.P = m.P; p.H = m.H; p.fragE = m.fragE; p.fragO = m.fragO; p.fragE2 = m.fragE2; p.fragO2 = m.fragO2; p.zero = m.zero; p.sink = m.sink; p.sym = m.sym;
  p.longDT = m.longDT; p.longD = m.longD;
  if (m.KS == 0) {
    if (!m.longDT || p.out_mode == OUT_MUL || p.out_mode == OUT_ACC2 || p.in_mode == IN_MUL) return hipErrorInvalidValue;
    // lines of 257 .. 1024 points: the library's own matrix-core kernel; option "long_lines_gemm": the rocBLAS route (A/B)
    if (!opt(OPT_LONG_LINES_GEMM) && sweep_xl_eligible(m, p)) return sweep_xl_launch(m, p, stream);
    return launch_long(p, stream);
  }
  {
    if (!opt(OPT_GENERAL_KERNELS) && sweep_vec_eligible(m, p)) {
      // diagnostic builds: three stamp areas of 256 x 8 x 16 words, used round-robin (one per launch of a 3-D matvec)
      if (chebhip_stamp_buf()) p.in4 = chebhip_stamp_buf() + (size_t)(chebhip_stamp_next() % 3) * (256 * 8 * 16);
      return sweep_vec_launch(m, p, stream);
    }
  }
  if (p.raw || p.in_fblocks || p.out_mode == OUT_MUL || p.out_mode == OUT_ACC2 || p.in_mode == IN_MUL) return hipErrorInvalidValue;   // the raw modes, OUT_MUL and spaced-out input fields exist in the 16-byte kernels only
  const bool jfast = p.inner < 16;
  switch (m.KS) {
    case 4: return jfast ? launch_t<4, true>(p, stream) : launch_t<4, false>(p, stream);
    case 8: return jfast ? launch_t<8, true>(p, stream) : launch_t<8, false>(p, stream);
    case 16: return jfast ? launch_t<16, true>(p, stream) : launch_t<16, false>(p, stream);
    case 32: return jfast ? launch_t<32, true>(p, stream) : launch_t<32, false>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t sweep_launch_gather(const DiffMat &m, SweepParams p, const GatherSrc &g, hipStream_t stream, bool *done) {
  *done = false;
  if (m.KS == 0 || opt(OPT_GENERAL_KERNELS)) return hipSuccess;
  p.P = m.P; p.H = m.H; p.fragE = m.fragE; p.fragO = m.fragO; p.fragE2 = m.fragE2; p.fragO2 = m.fragO2; p.zero = m.zero; p.sink = m.sink; p.sym = m.sym;
  p.longDT = m.longDT; p.longD = m.longD;
  return sweep_vec_launch_gather(m, p, g, stream, done);
}

static void sweep_fill(const DiffMat &m, SweepParams &p) {
  p.P = m.P; p.H = m.H; p.fragE = m.fragE; p.fragO = m.fragO; p.fragE2 = m.fragE2; p.fragO2 = m.fragO2; p.zero = m.zero; p.sink = m.sink;
  p.sym = m.sym; p.longDT = m.longDT; p.longD = m.longD;
}

hipError_t sweep_launch_multi_try(int n, const DiffMat *const *m, const SweepParams *p, hipStream_t stream, bool *done) {
  *done = false;
  if (n < 2 || n > 9 || opt(OPT_SEPARATE_LAUNCHES) || opt(OPT_GENERAL_KERNELS)) return hipSuccess;
  SweepParams jobs[9];
  for (int j = 0; j < n; j++) { jobs[j] = p[j]; sweep_fill(*m[j], jobs[j]); if (m[j]->KS == 0) return hipSuccess; }
  return sweep_vec_launch_multi(n, m, jobs, stream, done);
}

hipError_t sweep_launch_multi_gather_try(int n, const DiffMat *const *m, const SweepParams *p, unsigned gmask, const GatherSrc &g, hipStream_t stream, bool *done) {
  *done = false;
  if (n < 2 || n > 9 || opt(OPT_SEPARATE_LAUNCHES) || opt(OPT_GENERAL_KERNELS)) return hipSuccess;
  SweepParams jobs[9];
  for (int j = 0; j < n; j++) { jobs[j] = p[j]; sweep_fill(*m[j], jobs[j]); if (m[j]->KS == 0) return hipSuccess; }
  return sweep_vec_launch_multi_gather(n, m, jobs, gmask, g, stream, done);
}

hipError_t sweep_launch_multi(int n, const DiffMat *const *m, const SweepParams *p, hipStream_t stream) {
  if (n >= 2 && n <= 9 && !opt(OPT_SEPARATE_LAUNCHES) && !opt(OPT_GENERAL_KERNELS)) {      // "general_kernels": EVERY sweep runs the general kernel
    SweepParams jobs[9];
    bool ok = true;
    for (int j = 0; j < n; j++) { jobs[j] = p[j]; sweep_fill(*m[j], jobs[j]); ok = ok && m[j]->KS != 0; }
    if (ok) {
      bool done = false;
      hipError_t e = sweep_vec_launch_multi(n, m, jobs, stream, &done);
      if (e != hipSuccess || done) return e;
    }
  }
  for (int j = 0; j < n; j++) { hipError_t e = sweep_launch(*m[j], p[j], stream); if (e != hipSuccess) return e; }
  return hipSuccess;
}

}  // namespace chebhip
