// sweep_xl.hip -- ChebMult on lines of 257 .. 1024 points on the FP64 matrix cores (the reference accepts any extent,
// chebyshev.c:98; README:21 runs arbitrary extents).
//
// The matrix halves of such a line (H = ceil(P/2) up to 512: 2 MiB each) fit neither the registers nor the LDS of a
// workgroup, so the roles of sweep_vec.hip are swapped: the LINES of a tile sit in LDS -- the parity-split image
// e_j = x_j + x_{n-j}, o_j = x_j - x_{n-j} of 16 or 32 lines, 128 points of the half at a time (32 / 64 KiB: two workgroups
// per CU, one loads its next chunk while the other multiplies) -- and the MATRIX streams past them from L2 in MFMA-operand
// order ([m-tile][k-step][64 lanes]: one coalesced 512-byte load per fragment, no LDS for it), a few k-steps ahead of its
// use, in register queues that run on across the chunks.  A workgroup is (tile of lines) x (block of 128 or 256 output
// rows of each half): a wave owns one or two m-tiles of 16 rows and runs, per m-tile and 16 lines, an even and an odd
// accumulation chain of H/4 v_mfma_f64_16x16x4_f64.  With 32 lines per tile every fragment feeds two MFMAs, which halves
// the L2 traffic (at one MFMA per fragment the matrix stream alone would need ~20 TB/s chip-wide at the MFMA peak, more
// than the L2s deliver); with two m-tiles per wave every image operand read from LDS feeds two MFMAs as well and a tile is
// fetched by half as many workgroups (1024 x 8192: 194 -> 176 us).  Tiles of 16 lines are for grids too small to fill
// the chip otherwise.
// Same even/odd arithmetic as the short-line kernels (P flop per point instead of the 2 P of a dense product), same two
// tilings: COLFAST (line stride >= 16: lanes across 16 neighbouring lines, matrix = A operand) and JFAST (contiguous
// lines: lanes along the line, matrix = B operand, so that a wave stores 128 contiguous bytes per line).
// Plain input, STORE or ACC output.  Every tile is read once per row block (2 .. 4 times, mostly from L2 / MALL); at
// these lengths the product costs >= 257 flop per 8 bytes read.
#include "sweep.h"
#include <atomic>

namespace chebhip {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned u32;

constexpr int XL_MB = 8;           // waves per workgroup as launched (the kernel takes 4..8); each owns one or two m-tiles (of 16 rows)
constexpr int XL_KC = 128;         // points of the half per image chunk (32 k-steps)
constexpr int XL_LDJ_PAD = 1;   // pitch of a JFAST image line beyond XL_KC doubles (odd; the kernel's LDS accesses are all 8-byte)

// NT2: 16-line tiles per workgroup (1 or 2); MT2: m-tiles per wave (1 or 2: wave w owns m-tiles w and w + 8 of the
// workgroup's 16 -- every fragment still feeds NT2 MFMAs, every image operand now MT2 of them, and a tile is read by half
// as many workgroups); PF: k-steps of matrix fragments in flight per m-tile
template <bool JFAST, int NT2, int MT2, int PF>
__global__ __launch_bounds__(512, 4) void cheb_sweep_xl_kernel   // (4 waves per SIMD: two workgroups per CU)
(const SweepParams p) {
  extern __shared__ double smem[];
  constexpr int NL = 16 * NT2;                   // lines per tile
  constexpr int LDJ = XL_KC + XL_LDJ_PAD;        // JFAST image pitch: ODD -> conflict-free operand reads (round 6: tools/lds_probe.hip)
  constexpr int IMG = JFAST ? NL * LDJ : XL_KC * NL;
  constexpr int HIT = 4;                         // image elements a thread has in flight
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nthr = blockDim.x, W = nthr >> 6;    // waves of this workgroup
  const int kq = lane >> 4, l16 = lane & 15;
  const int P = p.P, nn = P - 1, H = p.H;
  const int KS = (int)p.in_rs;                   // k-steps (of 4) per chain, a multiple of 8   (launcher: set from the DiffMat)
  double *imgE = smem, *imgO = smem + IMG;
  const u32 inner = p.inner, ncols = p.ncols;
  const u32 lineLen = (u32)P * inner;
  const int mt0 = (int)blockIdx.y * (W * MT2) + w;       // this wave's m-tiles: mt0 + W u; rows 16 mt .. 16 mt + 15 of both halves

  // ---- tile -> first line / line stride -----------------------------------------------------------------------
  u32 base, lstride, qlim;                       // element offset of (line 0 of the tile, point 0); stride between the tile's lines
  if (JFAST) { base = blockIdx.x * NL * lineLen; lstride = lineLen; qlim = ncols - blockIdx.x * NL; }
  else {
    const u32 tpo = (inner + NL - 1) / NL, o = blockIdx.x / tpo, q0 = (blockIdx.x - o * tpo) * NL;
    base = o * lineLen + q0; lstride = 1; qlim = inner - q0;
  }
  if (qlim > (u32)NL) qlim = NL;                 // lines of the tile that exist

  // ---- fragment queues: PF k-steps ahead, across the chunks --------------------------------------------------------
  // (the m-tiles are padded to whole workgroups: a wave whose rows lie beyond the half only helps with the images)
  bool live[MT2];
  const double *fE[MT2], *fO[MT2];
  double ae[MT2][PF], ao[MT2][PF];
  v4d ce[MT2][NT2], co[MT2][NT2];
#pragma unroll
  for (int u = 0; u < MT2; u++) {
    const int mt = mt0 + W * u;
    live[u] = 16 * mt < H;
    fE[u] = p.fragE + ((long)mt * KS) * 64 + lane; fO[u] = p.fragO + ((long)mt * KS) * 64 + lane;
#pragma unroll
    for (int s = 0; s < PF; s++) { ae[u][s] = live[u] ? fE[u][(long)s * 64] : 0.0; ao[u][s] = live[u] ? fO[u][(long)s * 64] : 0.0; }
#pragma unroll
    for (int t = 0; t < NT2; t++) { ce[u][t] = (v4d){0.0, 0.0, 0.0, 0.0}; co[u][t] = (v4d){0.0, 0.0, 0.0, 0.0}; }
  }

  for (int k0 = 0; k0 < KS; k0 += XL_KC / 4) {
    // ---- parity-split image of points 4 k0 .. 4 k0 + 127 of the half (the mirror of row j is n - j; a self-mirrored
    //      middle row keeps e = x, o = 0; rows >= H and missing lines are zero) -----------------------------------------
    if (k0) __syncthreads();
    for (int h0 = 0; h0 * nthr < XL_KC * NL; h0 += HIT) {
      double va[HIT], vb[HIT];
#pragma unroll
      for (int it = 0; it < HIT; it++) {
        const int t = tid + (h0 + it) * nthr;
        int jj, l;
        if (JFAST) { l = t / XL_KC; jj = t - l * XL_KC; } else { jj = t / NL; l = t - jj * NL; }
        const int j = 4 * k0 + jj;
        const bool ok = t < XL_KC * NL && j < H && (u32)l < qlim;
        const u32 a = base + (u32)l * lstride;
        va[it] = ok ? p.in0[a + (u32)j * inner] : 0.0;
        vb[it] = (ok && 2 * j != nn) ? p.in0[a + (u32)(nn - j) * inner] : 0.0;
      }
#pragma unroll
      for (int it = 0; it < HIT; it++) {
        const int t = tid + (h0 + it) * nthr;
        if (t >= XL_KC * NL) continue;
        int jj, l;
        if (JFAST) { l = t / XL_KC; jj = t - l * XL_KC; } else { jj = t / NL; l = t - jj * NL; }
        const int j = 4 * k0 + jj;
        // COLFAST, 32 lines: rows of odd k are stored with the two 16-line halves swapped, so that the lanes of k and k + 1
        // (one half-wave of a B-operand read) fall on different banks
        const int idx = JFAST ? l * LDJ + jj : jj * NL + (NT2 == 2 ? (l ^ ((jj & 1) << 4)) : l);
        imgE[idx] = va[it] + vb[it];
        imgO[idx] = (2 * j == nn) ? 0.0 : va[it] - vb[it];
      }
    }
    __syncthreads();

    const int kend = live[0] ? min(KS - k0, XL_KC / 4) : 0;            // (m-tile 0 of a wave is the lower one: dead => both dead)
    for (int g = 0; g < kend; g += PF) {
#pragma unroll
      for (int s = 0; s < PF; s++) {
        const int ks = g + s;                      // k-step within the chunk: points 4 ks + kq
        const int nx = min(k0 + ks + PF, KS - 1);              // (the tail re-reads the last fragment instead of branching)
        double me[MT2], mo[MT2];
#pragma unroll
        for (int u = 0; u < MT2; u++) {
          me[u] = ae[u][s]; mo[u] = ao[u][s];
          if (u == 0 || live[u]) { ae[u][s] = fE[u][(long)nx * 64]; ao[u][s] = fO[u][(long)nx * 64]; }
        }
#pragma unroll
        for (int t = 0; t < NT2; t++) {
          const int bi = JFAST ? (16 * t + l16) * LDJ + 4 * ks + kq
                               : (4 * ks + kq) * NL + (NT2 == 2 ? ((16 * t + l16) ^ ((kq & 1) << 4)) : l16);
          const double be = imgE[bi], bo = imgO[bi];
#pragma unroll
          for (int u = 0; u < MT2; u++) {
            if (u > 0 && !live[u]) continue;
            if (!JFAST) {
              ce[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(me[u], be, ce[u][t], 0, 0, 0);
              co[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(mo[u], bo, co[u][t], 0, 0, 0);
            } else {
              ce[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(be, me[u], ce[u][t], 0, 0, 0);
              co[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(bo, mo[u], co[u][t], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  // ---- y_i = a + b,  y_{n-i} = b - a (D: centro-antisymmetric) / a - b (centro-symmetric) -----------------------------
  // accumulator element r of lane-quarter kq is row 4 r + kq of the 16 x 16 product
  const double alpha = p.alpha;
  const bool accm = p.out_mode == OUT_ACC;
#pragma unroll
  for (int u = 0; u < MT2; u++)
#pragma unroll
    for (int t = 0; t < NT2; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int mt = mt0 + W * u;
        const int i = JFAST ? 16 * mt + l16 : 16 * mt + 4 * r + kq;              // output row of the half
        const u32 l = JFAST ? (u32)(16 * t + 4 * r + kq) : (u32)(16 * t + l16);  // line of the tile
        if (i >= H || l >= qlim) continue;
        const double hi = ce[u][t][r] + co[u][t][r], lo = p.sym ? ce[u][t][r] - co[u][t][r] : co[u][t][r] - ce[u][t][r];
        const u32 a_hi = base + l * lstride + (u32)i * inner, a_lo = base + l * lstride + (u32)(nn - i) * inner;
        p.out[a_hi] = accm ? p.acc[a_hi] + alpha * hi : alpha * hi;
        if (2 * i != nn) p.out[a_lo] = accm ? p.acc[a_lo] + alpha * lo : alpha * lo;
      }
}

// Lines of 257 .. 1024 points, plain input, STORE / ACC output, fragments present (diffmat_create_long builds them).
bool sweep_xl_eligible(const DiffMat &m, const SweepParams &p) {
  if (m.KS != 0 || !m.fragE || !m.fragO || m.xl_ks <= 0 || m.P <= 256 || m.P > 1024) return false;   // (P <= 256 with KS == 0: option force_gemm)
  if (p.in_mode != IN_PLAIN || (p.out_mode != OUT_STORE && p.out_mode != OUT_ACC) || p.raw || p.in_fblocks || p.qmax || p.in_os) return false;
  if ((unsigned long long)p.ncols * (unsigned long long)m.P >= 0x100000000ull) return false;   // element offsets are 32-bit in the kernel
  if (p.inner >= 16) return true;
  return p.inner == 1;                          // JFAST needs contiguous lines; other small strides stay with the VALU kernel
}

template <bool JFAST, int NT2, int MT2, int PF>
static hipError_t xl_launch_t(const SweepParams &p, dim3 grid, unsigned waves, hipStream_t stream) {
  const size_t lds = (size_t)2 * (JFAST ? 16 * NT2 * (XL_KC + XL_LDJ_PAD) : XL_KC * 16 * NT2) * sizeof(double);
  if (lds > 64 * 1024) {                         // (more than 64 KiB of dynamic LDS needs the attribute, once per device)
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !(done.load() >> dev & 1ull)) {
      e = hipFuncSetAttribute((const void *)cheb_sweep_xl_kernel<JFAST, NT2, MT2, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if (dev < 64) done.fetch_or(1ull << dev);
    }
  }
  hipLaunchKernelGGL((cheb_sweep_xl_kernel<JFAST, NT2, MT2, PF>), grid, dim3(64 * waves), lds, stream, p);
  sweep_note_launch();
  return hipGetLastError();
}

hipError_t sweep_xl_launch(const DiffMat &m, SweepParams p, hipStream_t stream) {
  const bool jfast = p.inner < 16;
  p.in_rs = (unsigned)m.xl_ks;                   // (the kernel reads its k-step count here: the per-array geometry is unused on this path)
  const unsigned mtiles = (unsigned)((m.H + 15) / 16);
  const unsigned W = XL_MB, mb1 = (mtiles + W - 1) / W, mb2 = (mtiles + 2 * W - 1) / (2 * W), pad2 = mb2 * 2 * W - mtiles;
  auto tiles = [&](unsigned nl) { return jfast ? (p.ncols + nl - 1) / nl : (p.ncols / p.inner) * ((p.inner + nl - 1) / nl); };
  const unsigned t32 = tiles(32);
  // Two m-tiles per wave (32 lines per tile) when the 16 m-tiles of a workgroup are (nearly) all real and every CU still gets
  // its two workgroups; else one m-tile per wave; tiles of 16 lines for small grids.  (Workgroups of 5 or 6 waves with two
  // m-tiles each, so that 10 or 12 m-tiles fill them exactly, measured slower than 8 waves with idle ones: 300^3 313 against
  // 296 us, 384^3 622 against 592 us.)
  if (pad2 <= 3 && t32 * mb2 >= 512) {
    const dim3 grid(t32, mb2);
    return jfast ? xl_launch_t<true, 2, 2, 2>(p, grid, W, stream) : xl_launch_t<false, 2, 2, 2>(p, grid, W, stream);
  }
  if (t32 * mb1 >= 512) {
    const dim3 grid(t32, mb1);
    return jfast ? xl_launch_t<true, 2, 1, 8>(p, grid, W, stream) : xl_launch_t<false, 2, 1, 8>(p, grid, W, stream);
  }
  const unsigned t16 = tiles(16);
  if (t16 == 0) return hipSuccess;
  const dim3 grid(t16, mb1);
  return jfast ? xl_launch_t<true, 1, 1, 8>(p, grid, W, stream) : xl_launch_t<false, 1, 1, 8>(p, grid, W, stream);
}

}  // namespace chebhip
