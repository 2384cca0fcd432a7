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
//  - workgroups are persistent (grid = #CUs) and walk tiles of NT lines; per tile the
//    workgroup loads the lines from HBM, forms e = x_j + x_{n-j}, o = x_j - x_{n-j} on the
//    fly and parks them in LDS (64 KiB);
//  - each wave runs v_mfma_f64_16x16x4_f64 chains over the LDS tile and stores
//    y_i = a_i + b_i and y_{n-i} = b_i - a_i straight from the accumulators.
//  - two tilings: COLFAST (inner stride >= 16: neighbouring lanes = neighbouring lines,
//    matrix is the A operand) and JFAST (inner stride small, e.g. 1 or the d interleaved
//    Stokes components: neighbouring lanes = neighbouring points of a line, matrix is the B
//    operand so that the 16 lanes of an accumulator row are 16 consecutive outputs of a line).
#include "sweep.h"
#include <atomic>

namespace chebhip {

typedef double v4d __attribute__((ext_vector_type(4)));

static std::atomic<long> g_launches{0};
long sweep_launch_count() { return g_launches.load(); }

__device__ __forceinline__ double fetch_in(const SweepParams &p, long a, int j, int gb) {
  switch (p.in_mode) {
    case IN_PLAIN: return p.in0[a];
    case IN_GATHER: return (gb >= 0 && j >= 1 && j <= p.P - 2) ? p.in0[(long)gb + (long)(j - 1) * p.gstride] : 0.0;
    case IN_FLUX_ETA: return p.in1[a] * p.in0[a];
    default: return p.in1[a] * p.in0[a] + p.in2[a] * p.in3[a] * p.in4[a];
  }
}

__device__ __forceinline__ void emit_out(const SweepParams &p, long a, int i, int gb, double r) {
  switch (p.out_mode) {
    case OUT_STORE: p.out[a] = p.alpha * r; break;
    case OUT_ACC: p.out[a] = p.acc[a] + p.alpha * r; break;
    default:
      if (gb >= 0 && i >= 1 && i <= p.P - 2)
        p.out[(long)gb + (long)(i - 1) * p.gstride] = (p.acc ? p.acc[a] : 0.0) + p.alpha * r;
      break;
  }
}

template <int KS, bool JFAST>
__global__ __launch_bounds__(512) void cheb_sweep_kernel(const SweepParams p) {
  constexpr int MTP = KS / 4;                      // m-tiles of 16 output rows (padded)
  constexpr int NG = 8 / MTP;                      // wave groups along the line index
  constexpr int HP = 4 * KS;                       // padded half length
  constexpr int NSUB = (KS >= 16) ? 2 : 1;         // 16-line sub-tiles per wave per tile
  constexpr int NT = 16 * NG * NSUB;               // lines per tile: 32, 64, 64, 128
  constexpr int LDJ = HP + 2;                      // JFAST row pitch: == 2 (mod 32) -> conflict-free b64 reads
  constexpr int LDS_ELEMS = JFAST ? NT * LDJ : HP * NT;
  __shared__ double smem[2 * LDS_ELEMS];
  double *sE = smem, *sO = smem + LDS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mt = w % MTP, ng = w / MTP;
  const int nn = p.P - 1, H = p.H;
  const unsigned inner = p.inner, ncols = p.ncols;
  const long lineLen = (long)p.P * inner;
  const bool need_g = (p.in_mode == IN_GATHER) || (p.out_mode == OUT_ACC_SCATTER);

  // Matrix fragments -> registers (coalesced 512 B per wave load).
  double ae[KS], ao[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) {
    ae[s] = p.fragE[((long)(mt * KS + s)) * 64 + lane];
    ao[s] = p.fragO[((long)(mt * KS + s)) * 64 + lane];
  }

  for (unsigned tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const unsigned c0 = tile * NT;
    // ---------------- load: HBM -> (e, o) -> LDS ----------------
    if (!JFAST) {
      const int n = tid % NT;
      const unsigned c = c0 + n;
      const bool cv = c < ncols;
      long base = 0; int gb = -1;
      if (cv) { base = (long)(c / inner) * lineLen + (c % inner); if (need_g) gb = p.gcol[c]; }
      for (int jp = tid / NT; jp < HP; jp += 512 / NT) {
        double e = 0.0, o = 0.0;
        if (cv && jp < H) {
          const int jm = nn - jp;
          const double xj = fetch_in(p, base + (long)jp * inner, jp, gb);
          if (jm != jp) { const double xm = fetch_in(p, base + (long)jm * inner, jm, gb); e = xj + xm; o = xj - xm; }
          else e = xj;
        }
        const int idx = jp * NT + (n ^ ((jp & 1) << 4));   // odd rows swap 16-column halves: bank spread
        sE[idx] = e; sO[idx] = o;
      }
    } else {
      const int jp = tid % HP;
      const int jm = nn - jp;
      for (int n = tid / HP; n < NT; n += 512 / HP) {
        const unsigned c = c0 + n;
        double e = 0.0, o = 0.0;
        if (c < ncols && jp < H) {
          const long base = (long)(c / inner) * lineLen + (c % inner);
          const int gb = need_g ? p.gcol[c] : -1;
          const double xj = fetch_in(p, base + (long)jp * inner, jp, gb);
          if (jm != jp) { const double xm = fetch_in(p, base + (long)jm * inner, jm, gb); e = xj + xm; o = xj - xm; }
          else e = xj;
        }
        sE[n * LDJ + jp] = e; sO[n * LDJ + jp] = o;
      }
    }
    __syncthreads();

    // ---------------- compute + store ----------------
#pragma unroll 1
    for (int sub = 0; sub < NSUB; sub++) {
      const int nb = (ng * NSUB + sub) * 16;
      v4d ce = {0.0, 0.0, 0.0, 0.0}, co = {0.0, 0.0, 0.0, 0.0};
      const int kq = lane >> 4, nl = nb + (lane & 15);
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const int k = 4 * s + kq;
        double be, bo;
        if (!JFAST) {
          const int idx = k * NT + (nl ^ ((k & 1) << 4));
          be = sE[idx]; bo = sO[idx];
          ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[s], be, ce, 0, 0, 0);   // rows = outputs i, cols = lines
          co = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[s], bo, co, 0, 0, 0);
        } else {
          const int idx = nl * LDJ + k;
          be = sE[idx]; bo = sO[idx];
          ce = __builtin_amdgcn_mfma_f64_16x16x4f64(be, ae[s], ce, 0, 0, 0);   // rows = lines, cols = outputs i
          co = __builtin_amdgcn_mfma_f64_16x16x4f64(bo, ao[s], co, 0, 0, 0);
        }
      }
      // accumulator element r of a lane: row = 4*r + (lane >> 4), col = lane & 15
      if (!JFAST) {
        const unsigned c = c0 + nb + (lane & 15);
        if (c < ncols) {
          const long base = (long)(c / inner) * lineLen + (c % inner);
          const int gb = need_g ? p.gcol[c] : -1;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int i = mt * 16 + 4 * r + kq;
            if (i < H) {
              const double a = ce[r], b = co[r];
              emit_out(p, base + (long)i * inner, i, gb, a + b);
              if (nn - i != i) emit_out(p, base + (long)(nn - i) * inner, nn - i, gb, b - a);
            }
          }
        }
      } else {
        const int i = mt * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned c = c0 + nb + 4 * r + kq;
          if (c < ncols && i < H) {
            const long base = (long)(c / inner) * lineLen + (c % inner);
            const int gb = need_g ? p.gcol[c] : -1;
            const double a = ce[r], b = co[r];
            emit_out(p, base + (long)i * inner, i, gb, a + b);
            if (nn - i != i) emit_out(p, base + (long)(nn - i) * inner, nn - i, gb, b - a);
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int KS, bool JFAST>
static hipError_t launch_t(const SweepParams &p0, hipStream_t stream) {
  constexpr int MTP = KS / 4, NG = 8 / MTP, NSUB = (KS >= 16) ? 2 : 1, NT = 16 * NG * NSUB;
  SweepParams p = p0;
  p.ntiles = (p.ncols + NT - 1) / NT;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0; hipDeviceProp_t prop;
    hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return e;
    e = hipGetDeviceProperties(&prop, dev); if (e != hipSuccess) return e;
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // persistent: one 512-thread workgroup per CU (the register-resident matrix allows no more)
  const unsigned grid = p.ntiles < (unsigned)ncu ? p.ntiles : (unsigned)ncu;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL((cheb_sweep_kernel<KS, JFAST>), dim3(grid), dim3(512), 0, stream, p);
  g_launches.fetch_add(1);
  return hipGetLastError();
}

hipError_t sweep_launch(const DiffMat &m, SweepParams p, hipStream_t stream) {
  p.P = m.P; p.H = m.H; p.fragE = m.fragE; p.fragO = m.fragO;
  const bool jfast = p.inner < 16;
  switch (m.KS) {
    case 4: return jfast ? launch_t<4, true>(p, stream) : launch_t<4, false>(p, stream);
    case 8: return jfast ? launch_t<8, true>(p, stream) : launch_t<8, false>(p, stream);
    case 16: return jfast ? launch_t<16, true>(p, stream) : launch_t<16, false>(p, stream);
    case 32: return jfast ? launch_t<32, true>(p, stream) : launch_t<32, false>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace chebhip
