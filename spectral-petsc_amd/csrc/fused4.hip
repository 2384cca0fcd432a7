// fused4.hip -- out (+)= alpha * D_k( coef( D_k u ) ) in one launch, written the way sweep_vec.hip's v4 kernel is.
//
// Same algorithm as fused.hip (stage 1: g = D u on the MFMA chains, flux f = eta g (+ c u) in the accumulator
// registers, parity-split f into a second LDS image; stage 2: t = D f, out = acc + alpha t), for the launches of
// MatMult_Elliptic with variable coefficients (elliptic.C:297-339) and of FormFunction (:481-533) on lines of 66..256
// points with even extents.  What changed is everything around the chains, because on gfx950 a vector instruction
// is paid in matrix-pipe time (FP64 MFMA and VALU do not co-execute) and fused.hip issued 340 of them
// next to every 64 MFMAs:
//   * every global access is a raw buffer access: 32-bit offset = per-lane constant + one scalar per tile, the
//     hardware range check is the mask (no selects, no 64-bit address arithmetic);
//   * no gather table: the Jacobian mode reads the interior-layout vector directly (points 0 and n of a line are
//     implicit zeros), the coefficient pairs through a shifted base into their local layout, the accumulator in the
//     padded interior layout of the constant-coefficient path;
//   * 16-byte accesses wherever the layout allows (input pairs, coefficient pairs, COLFAST accumulate / store);
//   * the tile loop is straight-line code, so that hipcc can count vmcnt exactly: coefficient and accumulator loads
//     are issued inside the chain whose epilogue uses them, the next tile's input rides under three chains.
// fused.hip stays for short lines (KS < 16), odd extents, d = 1 and the slab mode.
// (Round 4 tried 16-byte pieces for the stored gradient and w0 of the strided FormFunction launches -- the lane exchange the result
// stores use, 8 DPP broadcasts + selects per sub-tile in place of 8 of the 16 store instructions: FormFunction 256^3 507-512 us
// against 494-510 us with the 8-byte stores, A/B in one process.  The stores were not what the launch waits for; removed.)
#include "sweep.h"
#include <type_traits>

namespace chebhip {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_barrier4() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// pitch of a JFAST tile-image line beyond HP doubles (odd: see sweep_vec.hip V_LDJ_PAD)
#ifndef F4_LDJ_PAD
#define F4_LDJ_PAD 1
#endif
template <int KS, bool JFAST>
constexpr int f4_lds_doubles() {
  constexpr int MTP = KS / 4, NG = 8 / MTP, HP = 4 * KS, NT = 32 * NG, LDJ = HP + F4_LDJ_PAD;
  constexpr int LDS_ELEMS = JFAST ? NT * LDJ : HP * NT;
  constexpr int NFL = (KS == 32) ? (JFAST ? 7 : 8) : 0;
  return 4 * LDS_ELEMS + 8 * NFL * 64;
}

// Diagnostic builds only (-DF4_ABLATE=bits; tools/f4_ablate.sh): 1 = no coefficient loads, 2 = no accumulator loads,
// 4 = no next-tile input loads, 8 = no global stores, 16 = no MFMA chains.  Results are wrong; the timing shows what
// each stream costs.
#ifndef F4_ABLATE
#define F4_ABLATE 0
#endif

#define AO4(s_) (((s_) < KR) ? ao[((s_) < KR) ? (s_) : 0] : aoL[((s_) - KR) * 64])

template <int KS, bool JFAST, bool FULL, bool ACC, bool WIN, bool ETASQ, bool TRIMF>
__global__ __launch_bounds__(512) void cheb_fused4_kernel(const Fused4Params p) {
  constexpr int MTP = KS / 4;
  constexpr int NG = 8 / MTP;
  constexpr int HP = 4 * KS;
  constexpr int NSUB = 2;
  constexpr int NT = 16 * NG * NSUB;
  constexpr int LDJ = HP + F4_LDJ_PAD;   // odd: conflict-free operand reads (sweep_vec.hip V_LDJ_PAD); lines start on 8-byte boundaries
  constexpr int LDS_ELEMS = JFAST ? NT * LDJ : HP * NT;
  constexpr int ITEMS = HP * NT / 2 / 512;            // 16-B input slots per thread per tile
  constexpr int CH = ITEMS / NSUB;
  constexpr int QSTEP = JFAST ? 512 / (HP / 2) : 512 / (NT / 2);
  constexpr int LDS_QSTEP = JFAST ? QSTEP * LDJ : QSTEP * NT;
  constexpr int KSTR = JFAST ? 4 : 4 * NT;
  constexpr int NFL = (KS == 32) ? (JFAST ? 7 : 8) : 0;
  constexpr int KR = KS - NFL;
  constexpr u32 INVALID = 0x80000000u, T_INVALID = 0x40000000u;     // see sweep_vec.hip: arrays stay below 0x38000000 bytes
  constexpr int IT = (FULL || TRIMF) ? 1 : 0;         // input / accumulator hold interior points only
  constexpr int OT = (FULL || WIN || TRIMF) ? 1 : 0;  // so does the output
  constexpr bool SUB = WIN || (TRIMF && JFAST);       // the last launch of FormFunction: rhs -= b
  constexpr u32 CE = FULL ? 16u : 8u;                 // bytes of one coefficient element
  static_assert(KS >= 16 && CH >= 1, "two sub-tiles per tile");
  static_assert(!WIN || (JFAST && ACC && !FULL), "the window mode is the last launch of FormFunction");
  static_assert(!JFAST || ACC, "the contiguous direction is never the first one");
  static_assert(!ETASQ || !FULL, "eta = 1 + gamma u^2 formed on chip: FormFunction only (the line being differentiated is u)");
  static_assert(!TRIMF || (ETASQ && !WIN && !FULL), "FormFunction on the interior line space: eta on chip, homogeneous Dirichlet rows");

  __shared__ double smem[f4_lds_doubles<KS, JFAST>()];
  double *inE = smem, *inO = smem + LDS_ELEMS, *fE_ = smem + 2 * LDS_ELEMS, *fO_ = smem + 3 * LDS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mt = w % MTP, ng = w / MTP;
  const int kq = lane >> 4, l16 = lane & 15;
  const bool odd = l16 & 1; const int l16e = l16 & ~1;
  const int nn = p.P - 1, H = p.H;
  const u32 qmax = p.qmax;

  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_coef = __builtin_amdgcn_make_buffer_rsrc((void *)p.coef, 0, p.coef_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_gout = __builtin_amdgcn_make_buffer_rsrc((void *)(FULL ? (void *)p.in : (void *)p.gout), 0, FULL ? 0u : p.gout_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_acc = __builtin_amdgcn_make_buffer_rsrc((void *)(ACC ? p.acc : p.in), 0, ACC ? p.acc_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_sub = __builtin_amdgcn_make_buffer_rsrc((void *)((SUB && p.sub) ? p.sub : p.in), 0, (SUB && p.sub) ? p.sub_bytes : 0u, 0x00020000);
  // TRIMF, first direction: the local copy w0 of the state (scatter GL, elliptic.C:486-493) is stored on the way, same geometry as gout
  const __amdgpu_buffer_rsrc_t r_w0 = __builtin_amdgcn_make_buffer_rsrc((void *)((TRIMF && p.w0out) ? (void *)p.w0out : (void *)p.in), 0, (TRIMF && p.w0out) ? p.w0_bytes : 0u, 0x00020000);
  auto ld16 = [](__amdgpu_buffer_rsrc_t r, u32 off) { return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); };
  auto st16 = [](__amdgpu_buffer_rsrc_t r, u32 off, d2 v) { if (!(F4_ABLATE & 8) || v.x == 1.2345e300) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, (int)off, 0, 0); };
  auto ld8 = [](__amdgpu_buffer_rsrc_t r, u32 off) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); };
  auto st8 = [](__amdgpu_buffer_rsrc_t r, u32 off, double v) { if (!(F4_ABLATE & 8) || v == 1.2345e300) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, (int)off, 0, 0); };

  double ae[KS], ao[KR > 0 ? KR : 1];
  double *aoL = smem + 4 * LDS_ELEMS + (w * NFL) * 64 + lane;

  // XCD-aware tile walk; tile = (block o, NT lines starting at q0)
  const u32 tpo = p.tpo;
  const u32 nxcd = (gridDim.x % 8 == 0) ? 8u : 1u;
  const u32 t_per = (p.ntiles + nxcd - 1) / nxcd;
  const u32 t_lo = (blockIdx.x % nxcd) * t_per;
  const u32 t_hi = (t_lo + t_per < p.ntiles) ? t_lo + t_per : p.ntiles;
  const u32 t_step = gridDim.x / nxcd;
  auto tile_o = [&](u32 tl) -> u32 { return __umulhi(tl, p.tpo_inv); };
  // byte offset of a tile's origin in an array of geometry g with elements of esz bytes
  auto tile_off = [&](u32 tl, const F4Geom &g, u32 esz) -> u32 {
    const u32 o = tile_o(tl), q0 = (tl - o * tpo) * NT;
    return o * (g.os * esz) + q0 * (g.ls * esz);
  };

  // ---- loader: COLFAST (line pair 2*ld_a, row ld_b + s*QSTEP and its mirror); JFAST (point pair of line ld_b + s*QSTEP)
  const int ld_a = JFAST ? tid % (HP / 2) : tid % (NT / 2);
  const int ld_b = JFAST ? tid / (HP / 2) : tid / (NT / 2);
  const int ld_lds0 = JFAST ? ld_b * LDJ + 2 * ld_a + IT : ld_b * NT + ((2 * ld_a) ^ ((ld_b & 1) << 4));
  const u32 in_ls8 = p.gi.ls * 8u, in_rs8 = p.gi.rs * 8u;
  // JFAST with a trimmed line: the pair is (j, j+1) = (2a+1, 2a+2), stored points (2a, 2a+1): 16-B aligned in memory,
  // 8-B aligned in the LDS image; row 0 of the image stays zero
  const u32 lj = JFAST ? (u32)ld_b * in_ls8 + (u32)(2 * ld_a) * 8u : (u32)(2 * ld_a) * 8u + (u32)(ld_b - IT) * in_rs8;
  const u32 lm = JFAST ? (u32)ld_b * in_ls8 + (u32)(nn - 2 * ld_a - 1 - 2 * IT) * 8u : (u32)(2 * ld_a) * 8u + (u32)(nn - ld_b - IT) * in_rs8;
  const u32 lj0 = (!JFAST && IT && ld_b == 0) ? INVALID : lj;          // rows 0 and n of a trimmed line: zeros
  const u32 lm0 = (!JFAST && IT && ld_b == 0) ? INVALID : lm;
  const u32 slot8 = JFAST ? (u32)QSTEP * in_ls8 : (u32)QSTEP * in_rs8;

  auto issue_loads = [&](u32 tl, bool valid, int chunk, d2 (&rj)[CH], d2 (&rm)[CH]) {
    const u32 t0 = tile_off(tl, p.gi, 8u);
    if constexpr (F4_ABLATE & 4) { if (tl != p.ntiles + 12345u) {
#pragma unroll
      for (int s = 0; s < CH; s++) { rj[s] = d2{1.0 + s, 2.0}; rm[s] = d2{0.5, 0.25 * chunk}; }
      return; } }
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const int sg = chunk * CH + s;
      const u32 so = (u32)sg * slot8;
      rj[s] = ld16(r_in, (sg == 0 ? lj0 : lj) + (valid ? t0 + so : T_INVALID));
      rm[s] = ld16(r_in, (sg == 0 ? lm0 : lm) + (valid ? (JFAST ? t0 + so : t0 - so) : T_INVALID));
    }
  };
  auto park_chunk = [&](int chunk, const d2 (&rj)[CH], const d2 (&rm)[CH]) {
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const int idx = ld_lds0 + (chunk * CH + s) * LDS_QSTEP;
      if (!JFAST) {
        *(d2 *)(inE + idx) = rj[s] + rm[s];
        *(d2 *)(inO + idx) = rj[s] - rm[s];
      } else if (IT || (LDJ & 1)) {                                      // odd index or odd pitch: two 8-B halves (ds_write2_b64)
        inE[idx] = rj[s].x + rm[s].y; inE[idx + 1] = rj[s].y + rm[s].x;
        inO[idx] = rj[s].x - rm[s].y; inO[idx + 1] = rj[s].y - rm[s].x;
      } else {
        *(d2 *)(inE + idx) = d2{rj[s].x + rm[s].y, rj[s].y + rm[s].x};
        *(d2 *)(inO + idx) = d2{rj[s].x - rm[s].y, rj[s].y - rm[s].x};
      }
    }
  };

  // ---- stage 1: a lane's accumulator holds rows i0 + 4 r of line l16 (COLFAST) / row i0 of lines 4 r + kq (JFAST)
  const int i0 = mt * 16 + (JFAST ? l16 : kq);
  const u32 c_ls = p.gc.ls * CE, c_rs = p.gc.rs * CE;
  u32 k_hi[4], k_lo[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (!JFAST) {
      const int i = i0 + 4 * r; const bool ok = i < H;
      k_hi[r] = ok ? (u32)l16 * CE + (u32)i * c_rs : INVALID;
      k_lo[r] = ok ? (u32)l16 * CE + (u32)(nn - i) * c_rs : INVALID;
    } else {
      const bool ok = i0 < H;
      k_hi[r] = ok ? (u32)(4 * r + kq) * c_ls + (u32)i0 * CE : INVALID;
      k_lo[r] = ok ? (u32)(4 * r + kq) * c_ls + (u32)(nn - i0) * CE : INVALID;
    }
  }

  // LDS index of (row, line) of sub-tile 0 for r = 0; r adds 4 NT (COLFAST) / 4 LDJ (JFAST), the sub-tile 16 lines
  const int f_r = JFAST ? 4 * LDJ : 4 * NT;
  auto f_idx0 = [&](int sub) -> int {
    const int nb = (ng * NSUB + sub) * 16;
    return JFAST ? (nb + kq) * LDJ + i0 : i0 * NT + ((nb + l16) ^ ((i0 & 1) << 4));
  };

  typename std::conditional<FULL, d2, double>::type cv_hi[4], cv_lo[4];  // coefficients of the sub-tile in flight
  u32 tv1[JFAST ? 4 : 1];                                               // their masked tile offsets (gradient store)
  double ue[(FULL || ETASQ) ? 4 : 1], uo[(FULL || ETASQ) ? 4 : 1];

  auto coef_issue = [&](u32 tl, int sub) {
    const u32 o = tile_o(tl), q0 = (tl - o * tpo) * NT;
    const int nb = (ng * NSUB + sub) * 16;
    const u32 t0 = o * (p.gc.os * CE) + (q0 + (u32)nb) * c_ls;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (JFAST || r == 0) {
        const u32 q = q0 + (u32)nb + (JFAST ? (u32)(4 * r + kq) : (u32)l16);
        tv1[JFAST ? r : 0] = (q < qmax) ? t0 : T_INVALID;               // lines past the block's end
      }
      const u32 tv = tv1[JFAST ? r : 0];
      if constexpr (ETASQ) { (void)tv; }                                // eta comes from the tile image, see epi1
      else if constexpr (F4_ABLATE & 1) {
        if constexpr (FULL) { cv_hi[r] = d2{1.0, 0.5}; cv_lo[r] = d2{1.0, 0.25}; } else { cv_hi[r] = 1.0; cv_lo[r] = 1.5; }
        (void)tv;
      } else if constexpr (FULL) { cv_hi[r] = ld16(r_coef, k_hi[r] + tv); cv_lo[r] = ld16(r_coef, k_lo[r] + tv); }
      else { cv_hi[r] = ld8(r_coef, k_hi[r] + tv); cv_lo[r] = ld8(r_coef, k_lo[r] + tv); }
    }
  };
  auto u_read = [&](int sub) {
    if constexpr (FULL || ETASQ) {
      const int f0 = f_idx0(sub);
#pragma unroll
      for (int r = 0; r < 4; r++) { ue[r] = inE[f0 + r * f_r]; uo[r] = inO[f0 + r * f_r]; }
    }
  };
  auto epi1 = [&](int sub, const v4d &ce, const v4d &co) {
    const int f0 = f_idx0(sub);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const double gi = ce[r] + co[r], gm = co[r] - ce[r];              // D is centro-antisymmetric
      double fi, fm;
      if constexpr (FULL) {
        // the tile image holds e = u_i + u_{n-i}, o = u_i - u_{n-i}; the stored c is deta du0 / 2 (elliptic.C:321)
        fi = __builtin_fma(cv_hi[r].y, ue[r] + uo[r], cv_hi[r].x * gi);
        fm = __builtin_fma(cv_lo[r].y, ue[r] - uo[r], cv_lo[r].x * gm);
      } else {
        st8(r_gout, k_hi[r] + tv1[JFAST ? r : 0], gi);                  // c->gradu[k], elliptic.C:498
        st8(r_gout, k_lo[r] + tv1[JFAST ? r : 0], gm);
        if constexpr (ETASQ) {
          // eta = 1 + gamma u^2 (elliptic.C:508 with the default exponent) from the u the tile image holds (2 u_i = e + o,
          // 2 u_{n-i} = e - o): two multiply-adds instead of 8 bytes from HBM per value -- a byte costs the package about
          // twenty times what a flop does
          const double si = ue[r] + uo[r], sm = ue[r] - uo[r];
          if constexpr (TRIMF) {                                           // w0 = the line itself (2 u_i = e + o); dropped when w0out is null
            st8(r_w0, k_hi[r] + tv1[JFAST ? r : 0], 0.5 * si);
            st8(r_w0, k_lo[r] + tv1[JFAST ? r : 0], 0.5 * sm);
          }
          fi = __builtin_fma(p.gamma4 * si, si, 1.0) * gi; fm = __builtin_fma(p.gamma4 * sm, sm, 1.0) * gm;
        } else { fi = cv_hi[r] * gi; fm = cv_lo[r] * gm; }              // eta * g, elliptic.C:511
      }
      fE_[f0 + r * f_r] = fi + fm;
      fO_[f0 + r * f_r] = fi - fm;
    }
  };

  // ---- stage 2, COLFAST: after the lane exchange a lane owns, for rp = 0, 1, row i0 + 4 (2 rp + odd) and of it the two
  // adjacent columns starting at the even lane of its pair (16-B pieces, sweep_vec.hip)
  const u32 o_ls8 = p.go.ls * 8u, o_rs8 = p.go.rs * 8u, a_ls8 = p.ga.ls * 8u, a_rs8 = p.ga.rs * 8u;
  u32 o_hi[JFAST ? 1 : 2], o_lo[JFAST ? 1 : 2], c_hi[JFAST ? 1 : 2], c_lo[JFAST ? 1 : 2];
  if (!JFAST) {
#pragma unroll
    for (int rp = 0; rp < 2; rp++) {
      const int i = i0 + 4 * (2 * rp + (odd ? 1 : 0));
      const bool oko = (i < H) & !(OT && i == 0), oka = (i < H) & !(IT && i == 0);
      o_hi[rp] = oko ? (u32)l16e * 8u + (u32)(i - OT) * o_rs8 : INVALID;
      o_lo[rp] = oko ? (u32)l16e * 8u + (u32)(nn - i - OT) * o_rs8 : INVALID;
      c_hi[rp] = oka ? (u32)l16e * 8u + (u32)(i - IT) * a_rs8 : INVALID;
      c_lo[rp] = oka ? (u32)l16e * 8u + (u32)(nn - i - IT) * a_rs8 : INVALID;
    }
  } else {
    // JFAST: trimmed lines put the points at odd 8-B offsets, so the lane keeps its own row i0 of the four lines 4 r + kq
    // and moves 8 bytes at a time (a wave still covers whole 128-B segments)
    const bool oko = (i0 < H) & !(OT && i0 == 0), oka = (i0 < H) & !(IT && i0 == 0);
    o_hi[0] = oko ? (u32)kq * o_ls8 + (u32)(i0 - OT) * 8u : INVALID;
    o_lo[0] = oko ? (u32)kq * o_ls8 + (u32)(nn - i0 - OT) * 8u : INVALID;
    c_hi[0] = oka ? (u32)kq * a_ls8 + (u32)(i0 - IT) * 8u : INVALID;
    c_lo[0] = oka ? (u32)kq * a_ls8 + (u32)(nn - i0 - IT) * 8u : INVALID;
  }
  const double alpha = p.alpha, alpha_lo = -p.alpha;
  const bool win_o = WIN && p.nouter > 1;

  d2 acc_hi[2], acc_lo[2];                             // COLFAST: 2 x 16 B each; JFAST: the same registers as 4 + 4 doubles
  d2 sub_hi[SUB ? 2 : 1], sub_lo[SUB ? 2 : 1];
  u32 tv2[JFAST ? 4 : 1];                              // masked tile offsets of `out`
  auto acc_issue = [&](u32 tl, int sub) {
    const u32 o = tile_o(tl), q0 = (tl - o * tpo) * NT;
    const u32 nb = (u32)(ng * NSUB + sub) * 16u;
    if (!JFAST) {
      const u32 q = q0 + nb + (u32)l16e;
      tv2[0] = (q < qmax) ? o * (p.go.os * 8u) + (q0 + nb) * 8u : T_INVALID;
      if (ACC && !(F4_ABLATE & 2)) {
        const u32 ta = o * (p.ga.os * 8u) + (q0 + nb) * 8u;
#pragma unroll
        for (int rp = 0; rp < 2; rp++) { acc_hi[rp] = ld16(r_acc, c_hi[rp] + ta); acc_lo[rp] = ld16(r_acc, c_lo[rp] + ta); }
      }
    } else {
      const u32 ta = o * (p.ga.os * 8u) + (q0 + nb) * a_ls8;
      // window mode: `out` is indexed by (o - 1, q - 1) and has no slot for the boundary lines
      const u32 oo = win_o ? o - 1u : o, qt = WIN ? 1u : 0u;
      const bool o_ok = win_o ? (oo < p.nouter - 2u) : true;
      const u32 to = o_ok ? oo * (p.go.os * 8u) + (q0 + nb - qt) * o_ls8 : T_INVALID;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const u32 q = q0 + nb + (u32)(4 * r + kq) - qt;
        tv2[r] = (q < qmax - 2u * qt) ? to + (u32)(4 * r) * o_ls8 : T_INVALID;
        const double ah = (F4_ABLATE & 2) ? 1.0 : ld8(r_acc, c_hi[0] + ta + (u32)(4 * r) * a_ls8), al = (F4_ABLATE & 2) ? 2.0 : ld8(r_acc, c_lo[0] + ta + (u32)(4 * r) * a_ls8);
        if (r & 1) { acc_hi[r >> 1].y = ah; acc_lo[r >> 1].y = al; } else { acc_hi[r >> 1].x = ah; acc_lo[r >> 1].x = al; }
        if constexpr (SUB) {
          const double sh = ld8(r_sub, o_hi[0] + tv2[r]), sl = ld8(r_sub, o_lo[0] + tv2[r]);
          if (r & 1) { sub_hi[r >> 1].y = sh; sub_lo[r >> 1].y = sl; } else { sub_hi[r >> 1].x = sh; sub_lo[r >> 1].x = sl; }
        }
      }
    }
  };
  auto bc_even = [](double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xA0, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xA0, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  };
  auto bc_odd = [](double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xF5, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xF5, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  };
  auto epi2 = [&](const v4d &ce, const v4d &co) {
    double hi[4], lo[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { hi[r] = ce[r] + co[r]; lo[r] = ce[r] - co[r]; }
    if (!JFAST) {
#pragma unroll
      for (int rp = 0; rp < 2; rp++) {
        const double ha = hi[2 * rp], hb = hi[2 * rp + 1], la = lo[2 * rp], lb = lo[2 * rp + 1];
        // the broadcasts run with every lane active (a DPP read of an EXEC-disabled lane returns 0), the selects after
        const double hbe = bc_even(hb), hao = bc_odd(ha), lbe = bc_even(lb), lao = bc_odd(la);
        d2 vh = d2{odd ? hbe : ha, odd ? hb : hao};
        d2 vl = d2{odd ? lbe : la, odd ? lb : lao};
        if (ACC) { vh = acc_hi[rp] + alpha * vh; vl = acc_lo[rp] + alpha_lo * vl; }
        else { vh = alpha * vh; vl = alpha_lo * vl; }
        st16(r_out, o_hi[rp] + tv2[0], vh);
        st16(r_out, o_lo[rp] + tv2[0], vl);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double vh = ((r & 1) ? acc_hi[r >> 1].y : acc_hi[r >> 1].x) + alpha * hi[r];
        double vl = ((r & 1) ? acc_lo[r >> 1].y : acc_lo[r >> 1].x) + alpha_lo * lo[r];
        if constexpr (SUB) {                                            // rhs -= b, elliptic.C:530
          vh -= (r & 1) ? sub_hi[r >> 1].y : sub_hi[r >> 1].x;
          vl -= (r & 1) ? sub_lo[r >> 1].y : sub_lo[r >> 1].x;
        }
        st8(r_out, o_hi[0] + tv2[r], vh);
        st8(r_out, o_lo[0] + tv2[r], vl);
      }
    }
  };

  // three hooks at fixed k-steps of a chain: global loads, LDS parking, late LDS reads
  auto chain = [&](const double *sE, const double *sO, int sub, int g_a, int g_b, int g_c, v4d &ce, v4d &co,
                   auto &&fn_a, auto &&fn_b, auto &&fn_c) {
    const int nb = (ng * NSUB + sub) * 16;
    ce = v4d{0.0, 0.0, 0.0, 0.0}; co = v4d{0.0, 0.0, 0.0, 0.0};
    const int frag = JFAST ? (nb + l16) * LDJ + kq : kq * NT + ((nb + l16) ^ ((kq & 1) << 4));
    const double *fE = sE + frag, *fO = sO + frag;
    if constexpr (F4_ABLATE & 16) {                    // no matrix work: what the memory streams cost alone
      fn_a(); fn_b(); fn_c();
      ce[0] = fE[0] + ae[0]; co[0] = fO[KSTR] + AO4(KS - 1); ce[1] = fE[2 * KSTR]; co[2] = fO[3 * KSTR];
      return;
    }
    double fb[2][4];
    fb[0][0] = fE[0]; fb[0][1] = fE[KSTR]; fb[0][2] = fO[0]; fb[0][3] = fO[KSTR];
#pragma unroll
    for (int g = 0; g < KS / 2; g++) {
      const int cb = g & 1, nbuf = cb ^ 1;
      if (g + 1 < KS / 2) {
        fb[nbuf][0] = fE[(2 * g + 2) * KSTR]; fb[nbuf][1] = fE[(2 * g + 3) * KSTR];
        fb[nbuf][2] = fO[(2 * g + 2) * KSTR]; fb[nbuf][3] = fO[(2 * g + 3) * KSTR];
      }
      __builtin_amdgcn_sched_barrier(0);               // fragment reads stay one group ahead of their MFMAs
      if (g == g_a) fn_a();
      if (g == g_b) fn_b();
      if (g == g_c) fn_c();
      if (!JFAST) {
        ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[2 * g], fb[cb][0], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f64_16x16x4f64(AO4(2 * g), fb[cb][2], co, 0, 0, 0);
        ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[2 * g + 1], fb[cb][1], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f64_16x16x4f64(AO4(2 * g + 1), fb[cb][3], co, 0, 0, 0);
      } else {
        ce = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][0], ae[2 * g], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][2], AO4(2 * g), co, 0, 0, 0);
        ce = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][1], ae[2 * g + 1], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][3], AO4(2 * g + 1), co, 0, 0, 0);
      }
    }
  };

  u32 tile = t_lo + blockIdx.x / nxcd;
  if (tile >= t_hi) return;                            // whole workgroup: no barrier is skipped by part of it
  d2 rj[CH], rm[CH];
  {
    // first tile: both chunks requested BEFORE the matrix fragments (one memory round trip instead of two)
    d2 rjB[CH], rmB[CH];
    issue_loads(tile, true, 0, rj, rm); issue_loads(tile, true, 1, rjB, rmB);
    if (JFAST && IT) { if (tid < NT) { inE[tid * LDJ] = 0.0; inO[tid * LDJ] = 0.0; } }
    // the CUs of an XCD start at four different places of the fragment set (sweep_vec.hip)
    auto load_frags = [&](auto ROT_) {
      constexpr int ROT = decltype(ROT_)::value;
#pragma unroll
      for (int g0 = 0; g0 < KS / 2; g0++) {
        const int g = (g0 + ROT) % (KS / 2);
        const d2 ve = ((const d2 *)p.fragE2)[((long)(mt * (KS / 2) + g)) * 64 + lane];
        const d2 vo = ((const d2 *)p.fragO2)[((long)(mt * (KS / 2) + g)) * 64 + lane];
        ae[2 * g] = ve.x; ae[2 * g + 1] = ve.y;
        if (2 * g < KR) ao[2 * g] = vo.x; else aoL[(2 * g - KR) * 64] = vo.x;
        if (2 * g + 1 < KR) ao[2 * g + 1] = vo.y; else aoL[(2 * g + 1 - KR) * 64] = vo.y;
      }
    };
    switch ((blockIdx.x / nxcd) & 3u) {
      case 0: load_frags(std::integral_constant<int, 0>{}); break;
      case 1: load_frags(std::integral_constant<int, KS / 8>{}); break;
      case 2: load_frags(std::integral_constant<int, KS / 4>{}); break;
      default: load_frags(std::integral_constant<int, 3 * KS / 8>{}); break;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): see sweep.hip
    park_chunk(0, rj, rm); park_chunk(1, rjB, rmB);
  }

  auto run = [&](auto GRPB_) {
    // the two waves of a SIMD (w, w + 4) do their vector work at different k-steps of their chains
    constexpr bool GRPB = decltype(GRPB_)::value;
    constexpr int G_A = GRPB ? 0 : KS / 8, G_B = GRPB ? KS / 4 : 3 * KS / 8, G_C = KS / 2 - 2;
    lds_barrier4();
    for (; tile < t_hi; tile += t_step) {
      const u32 nxt = tile + t_step;
      const bool v1 = nxt < t_hi;
      v4d ce, co;
      // ---- stage 1: g = D u, f = coef(g) -> F
      chain(inE, inO, 0, G_A, G_B, G_C, ce, co, [&] { coef_issue(tile, 0); }, [] {}, [&] { u_read(0); });
      epi1(0, ce, co);
      chain(inE, inO, 1, G_A, G_B, G_C, ce, co, [&] { coef_issue(tile, 1); issue_loads(nxt, v1, 0, rj, rm); }, [] {}, [&] { u_read(1); });
      epi1(1, ce, co);
      lds_barrier4();                                  // F complete, IN dead
      // ---- stage 2: t = D f, out = acc + alpha t; the next tile goes into IN
      chain(fE_, fO_, 0, G_A, G_B, G_C, ce, co, [&] { acc_issue(tile, 0); },
            [&] { park_chunk(0, rj, rm); issue_loads(nxt, v1, 1, rj, rm); }, [] {});
      epi2(ce, co);
      chain(fE_, fO_, 1, G_A, G_B, G_C, ce, co, [&] { acc_issue(tile, 1); }, [&] { park_chunk(1, rj, rm); }, [] {});
      epi2(ce, co);
      lds_barrier4();                                  // IN complete, F dead
    }
  };
  if (w >= 4) run(std::true_type{}); else run(std::false_type{});
}

bool fused4_eligible(const DiffMat &m) { return (m.KS == 16 || m.KS == 32) && (m.P & 1) == 0 && m.fragE2 && m.sym == 0; }

template <int KS, bool JFAST, bool FULL, bool ACC, bool WIN, bool ETASQ = false, bool TRIMF = false>
static hipError_t launch4(const Fused4Params &p, unsigned grid, hipStream_t stream) {
  hipLaunchKernelGGL((cheb_fused4_kernel<KS, JFAST, FULL, ACC, WIN, ETASQ, TRIMF>), dim3(grid), dim3(512), 0, stream, p);
  sweep_note_launch();
  return hipGetLastError();
}

template <int KS>
static hipError_t launch4_ks(Fused4Params &p, bool jfast, bool full, bool acc, bool win, hipStream_t stream) {
  constexpr int NT = 32 * (8 / (KS / 4));
  p.tpo = (p.qmax + NT - 1) / NT;
  p.tpo_inv = (unsigned)((0x100000000ull + p.tpo - 1) / p.tpo);
  p.ntiles = p.nouter * p.tpo;
  if (p.ntiles == 0) return hipSuccess;
  if ((unsigned long long)p.ntiles * p.tpo >= 0x100000000ull) return hipErrorInvalidValue;   // exactness of tpo_inv
  hipError_t cu_err; const int ncu = sweep_num_cus(&cu_err);
  if (cu_err != hipSuccess) return cu_err;
  const unsigned grid = p.ntiles < (unsigned)ncu ? p.ntiles : (unsigned)ncu;
  const bool sq = p.eta_square != 0;                      // FormFunction with exponent 2: eta formed on chip
  if (full && sq) return hipErrorInvalidValue;
  if (p.trimf) {                                          // FormFunction on the interior line space (homogeneous Dirichlet rows, eta on chip)
    if (!sq || full || win) return hipErrorInvalidValue;
    if (jfast) return acc ? launch4<KS, true, false, true, false, true, true>(p, grid, stream) : hipErrorInvalidValue;
    return acc ? launch4<KS, false, false, true, false, true, true>(p, grid, stream) : launch4<KS, false, false, false, false, true, true>(p, grid, stream);
  }
  if (jfast) {
    if (!acc) return hipErrorInvalidValue;
    if (win) {
      if (full) return hipErrorInvalidValue;
      return sq ? launch4<KS, true, false, true, true, true>(p, grid, stream) : launch4<KS, true, false, true, true>(p, grid, stream);
    }
    if (full) return launch4<KS, true, true, true, false>(p, grid, stream);
    return sq ? launch4<KS, true, false, true, false, true>(p, grid, stream) : launch4<KS, true, false, true, false>(p, grid, stream);
  }
  if (win) return hipErrorInvalidValue;
  if (full) return acc ? launch4<KS, false, true, true, false>(p, grid, stream) : launch4<KS, false, true, false, false>(p, grid, stream);
  if (sq) return acc ? launch4<KS, false, false, true, false, true>(p, grid, stream) : launch4<KS, false, false, false, false, true>(p, grid, stream);
  return acc ? launch4<KS, false, false, true, false>(p, grid, stream) : launch4<KS, false, false, false, false>(p, grid, stream);
}

hipError_t fused4_launch(const DiffMat &m, Fused4Params p, bool jfast, bool full, bool acc, bool win, hipStream_t stream) {
  if (!fused4_eligible(m)) return hipErrorInvalidValue;
  p.P = m.P; p.H = m.H; p.fragE2 = m.fragE2; p.fragO2 = m.fragO2;
  // every offset the kernel forms must stay 16-B aligned where it moves 16 bytes, and below the T_INVALID marks
  const unsigned lim = 0x38000000u;
  if (p.in_bytes >= lim || p.coef_bytes >= lim || p.gout_bytes >= lim || p.acc_bytes >= lim || p.out_bytes >= lim || p.sub_bytes >= lim || p.w0_bytes >= lim)
    return hipErrorInvalidValue;
  auto al = [](const void *q) { return ((size_t)q & 15) == 0; };
  if (!al(p.in) || (!p.eta_square && !al(p.coef)) || !al(p.out) || (acc && !al(p.acc))) return hipErrorInvalidValue;
  if (jfast) { if ((p.gi.os | p.gi.ls) & 1) return hipErrorInvalidValue; }
  else if ((p.gi.os | p.gi.rs | p.go.os | p.go.rs | p.qmax) & 1 || (acc && ((p.ga.os | p.ga.rs) & 1))) return hipErrorInvalidValue;
  return m.KS == 16 ? launch4_ks<16>(p, jfast, full, acc, win, stream) : launch4_ks<32>(p, jfast, full, acc, win, stream);
}

}  // namespace chebhip
