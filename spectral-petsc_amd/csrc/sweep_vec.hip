// sweep_vec.hip -- cheb_sweep_kernel specialised for 16-byte global accesses.
//
// Same algorithm, tiling and LDS images as sweep.hip (matrix halves in registers, double-buffered
// parity-split tile, v_mfma_f64_16x16x4_f64 chains), restricted to what the hot launches need --
// plain input, STORE or ACC output -- and with every HBM access widened from 8 to 16 bytes per lane:
//   * a CU's load/store path moves 8-B accesses at ~0.6x the rate of 16-B ones, and at P = 256 the
//     launch is as much per-CU-bandwidth-bound as MFMA-bound;
//   * half as many memory instructions and address computations per point.
// Loads: COLFAST lanes take two neighbouring lines of one row, JFAST lanes two neighbouring points
// (and the mirrored pair) of one line; both go to LDS with one ds_write_b128 per image.
// Stores: the accumulator layout gives a lane ONE column (COLFAST) / ONE point (JFAST) of four rows;
// neighbouring lanes swap half of their values (DPP quad_perm) so that each ends up with TWO adjacent
// columns/points of two rows and stores them as one 16-B piece.
// Preconditions (checked on the host, otherwise sweep.hip runs): COLFAST: even line stride;
// JFAST: stride 1 and even line length; all arrays 16-B aligned.
#include "sweep.h"
#include <cstdlib>
#include <type_traits>

const double *chebhip_stamp_buf();   // chebhip.hip: diagnostic builds (CHEB_STAMPS) write in-kernel cycle stamps there
int chebhip_stamp_next();

namespace chebhip {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// Pitch of a line of the JFAST tile image in LDS, in doubles beyond the half length HP.  ODD (1): the MFMA operand reads -- lane
// (l16, kq) reads points kq + 4k, kq + 4k + 4 of line l16 as one ds_read2_b64 -- are free of bank conflicts; with the pitch = 2 mod 32
// of rounds 1-5 every such read is a 2-way conflict (tools/lds_probe.hip under --pmc, profiles/r06_lds_probe.txt: SQ_LDS_BANK_CONFLICT
// = half of SQ_LDS_IDX_ACTIVE at pitch 130, 0 at 129, three quarters at 132 -- the 8.26 M conflict cycles of the JFAST launch in every
// counter record since round 3).  The price: lines start on 8-byte boundaries only, so the parity split parks its 16-byte pieces as
// ds_write2_b64 instead of ds_write_b128.
#ifndef V_LDJ_PAD
#define V_LDJ_PAD 1
#endif
// one 16-byte piece into an LDS image whose lines may start on 8-byte boundaries
template <bool ALIGNED16>
__device__ __forceinline__ void lds_put2(double *dst, double __attribute__((ext_vector_type(2))) v) {
  if (ALIGNED16) *(double __attribute__((ext_vector_type(2))) *)dst = v;
  else { dst[0] = v.x; dst[1] = v.y; }
}

__device__ __forceinline__ void lds_barrier_v() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exchange with the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ double swap1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// LDS doubles of the 16-byte kernels for a (KS, tiling) pair: two (E, O) tile images, double-buffered, plus the
// LDS-resident matrix fragments at KS = 32
template <int KS, bool JFAST>
constexpr int vec_lds_doubles() {
  constexpr int MTP = KS / 4, NG = 8 / MTP, HP = 4 * KS, NSUB = (KS >= 16) ? 2 : 1, NT = 16 * NG * NSUB, LDJ = HP + V_LDJ_PAD;
  constexpr int LDS_ELEMS = JFAST ? NT * LDJ : HP * NT;
  constexpr int NFL = (KS == 32) ? (JFAST ? 7 : 8) : 0;
  return 4 * LDS_ELEMS + 8 * NFL * 64;
}

// Body of cheb_sweep_vec_kernel (lines of at most 64 points, KS <= 8): workgroup BID of NBLK (the launch's, or those of one job
// of a multi-job launch).
//
// At these sizes a launch is all latency (64^3: two tiles per workgroup, the data sits in L2 / the Infinity Cache).  In-kernel stamps of
// the nine-job launch of a 64^3 StokesMatMult (tools/stamp_probe_multi.py, profiles/r06_multi64_stamps.txt) with the round 1-5 schedule:
// matrix fragments landed after 2.6-5.3 k cycles, THEN the first tile was requested and the loop entered after 5.6-10 k; each of the two
// tiles took ~6 k, 2.4-3.4 k of it the wait for the next tile's lines, requested only one short chain earlier -- 17-25 k cycles per
// workgroup.  Round 6:
//   * the first TWO tiles are requested before the matrix fragments: one memory round trip in front of the loop instead of two;
//   * two register sets (A, B) of prefetched lines: tile t + 2 is requested at the top of tile t and split into LDS at the end of tile
//     t + 1's chain, i.e. it has a whole tile to land instead of a chain of 16-32 MFMAs;
//   * the loop is unrolled by two (static register sets) and has no branch around a memory instruction (invalid tiles read the zero
//     line), so hipcc's wait counts are exact: the wait for tile t + 1 does not cover the request for t + 2.
// Launches with a VecAXPY operand (ACC) and IN_SUM3 (the three-term input of grad div v, whose loads are all in flight before the first
// sum) have ONE register set -- a second one does not fit 128 VGPRs, i.e. two workgroups per CU -- and request tile t + 2 right after the
// split of tile t + 1.
// ACC = false: plain STORE launches only (every job of a multi-job launch) -- no operand registers.
template <int KS, bool JFAST, bool SUM3 = false, bool ACC = true>
__device__ __forceinline__ void vec1_body(const SweepParams &p, double *smem, const u32 BID, const u32 NBLK) {
  static_assert(KS <= 8, "lines of more than 64 points run vec4_body");
  static_assert(!SUM3 || !ACC, "IN_SUM3 exists for plain stores only");
  constexpr int MTP = KS / 4;
  constexpr int NG = 8 / MTP;
  constexpr int HP = 4 * KS;
  constexpr int NT = 16 * NG;
  constexpr int LDJ = HP + V_LDJ_PAD;
  constexpr int LDS_ELEMS = JFAST ? NT * LDJ : HP * NT;
  constexpr int CH = HP * NT / 2 / 512;               // 16-B slots per thread per tile
  constexpr int QSTEP = JFAST ? 512 / (HP / 2) : 512 / (NT / 2);   // line step (JFAST) / j-pair step (COLFAST)
  constexpr int LDS_QSTEP = JFAST ? QSTEP * LDJ : QSTEP * NT;
  constexpr int KSTR = JFAST ? 4 : 4 * NT;
  static_assert(CH >= 1 && (QSTEP % 2 == 0 || JFAST), "tile geometry");

#ifdef CHEB_STAMPS
  const unsigned long long st_entry = __builtin_amdgcn_s_memtime(), rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mt = w % MTP, ng = w / MTP;
  const int kq = lane >> 4, l16 = lane & 15;
  const int odd = l16 & 1, l16e = l16 & ~1;
  const int nn = p.P - 1, H = p.H;
  const u32 inner = p.inner, ncols = p.ncols;
  const u32 lineLen = (u32)p.P * inner;

  const u32 tpo = JFAST ? 1u : (inner + NT - 1) / NT;
  const u32 nxcd = (NBLK % 8 == 0) ? 8u : 1u;
  const u32 t_per = (p.ntiles + nxcd - 1) / nxcd;
  const u32 t_lo = (BID % nxcd) * t_per;
  const u32 t_hi = (t_lo + t_per < p.ntiles) ? t_lo + t_per : p.ntiles;
  const u32 t_step = NBLK / nxcd;
  u32 tile = t_lo + BID / nxcd;
  if (tile >= t_hi) return;                             // the whole workgroup: no barrier is skipped by part of it

  // loader slots: COLFAST (line pair 2*ld_a, j-pair ld_b + s*QSTEP); JFAST (points 2*ld_a, 2*ld_a+1 of line ld_b + s*QSTEP)
  const int ld_a = JFAST ? tid % (HP / 2) : tid % (NT / 2);
  const int ld_b = JFAST ? tid / (HP / 2) : tid / (NT / 2);
  const int ld_lds0 = JFAST ? ld_b * LDJ + 2 * ld_a : ld_b * NT + ((2 * ld_a) ^ ((ld_b & 1) << 4));

  // two register sets of prefetched lines where the register file allows it at two workgroups per CU (128 VGPRs): plain stores
  constexpr bool TWO = !SUM3 && !ACC;
  struct Lines { d2 j[CH], m[CH]; };
  Lines LA, LB;
  const d2 *zero2 = (const d2 *)p.zero;

  const bool raw_in = p.raw == 2, raw_out = p.raw == 1;   // the line transforms of precond.hip (sweep.h)
  // the lines of tile `tl` -> registers; !valid (past the workgroup's last tile): every lane reads the zero line
  auto issue_loads = [&](u32 tl, bool valid, Lines &L) {
    d2 j1[SUM3 ? CH : 1], m1[SUM3 ? CH : 1], j2[SUM3 ? CH : 1], m2[SUM3 ? CH : 1];
    if (!JFAST) {
      const u32 o = tl / tpo, q0 = (tl - o * tpo) * NT;
      const u32 q = q0 + 2 * ld_a;
      const bool cv = valid && q < inner;                   // inner is even: the pair is in or out together
      const u32 base = o * lineLen + q;
      int jp = ld_b;
      u32 rel = (u32)jp * inner;
      const u32 top = base + (u32)nn * inner;
      asm volatile("" : "+v"(rel), "+v"(jp));
#pragma unroll
      for (int s = 0; s < CH; s++, jp += QSTEP, rel += QSTEP * inner) {
        const bool ok = cv && jp < H, okm = ok && nn - jp != jp;
        L.j[s] = *(ok ? (const d2 *)(p.in0 + (base + rel)) : zero2);
        L.m[s] = *(okm ? (const d2 *)(p.in0 + (top - rel)) : zero2);
        if (SUM3) {                                           // IN_SUM3: (in0 + in1) + in2, the order of the pointwise sum it replaces
          j1[s] = *(ok ? (const d2 *)(p.in1 + (base + rel)) : zero2); m1[s] = *(okm ? (const d2 *)(p.in1 + (top - rel)) : zero2);
          j2[s] = *(ok ? (const d2 *)(p.in2 + (base + rel)) : zero2); m2[s] = *(okm ? (const d2 *)(p.in2 + (top - rel)) : zero2);
        }
      }
    } else {
      const int j = 2 * ld_a;                               // points j, j+1 and their mirrors n-j-1, n-j
#pragma unroll
      for (int s = 0; s < CH; s++) {
        const u32 c = tl * NT + ld_b + s * QSTEP;
        const bool ok = valid && c < ncols && j < H;
        const u32 base = (ok ? c : 0u) * lineLen;
        L.j[s] = *(ok ? (const d2 *)(p.in0 + (base + (u32)j)) : zero2);
        L.m[s] = *(ok ? (const d2 *)(p.in0 + (base + (u32)(nn - j - 1))) : zero2);
        if (SUM3) {
          j1[s] = *(ok ? (const d2 *)(p.in1 + (base + (u32)j)) : zero2); m1[s] = *(ok ? (const d2 *)(p.in1 + (base + (u32)(nn - j - 1))) : zero2);
          j2[s] = *(ok ? (const d2 *)(p.in2 + (base + (u32)j)) : zero2); m2[s] = *(ok ? (const d2 *)(p.in2 + (base + (u32)(nn - j - 1))) : zero2);
        }
      }
    }
    if (SUM3) {                                               // all the loads of the tile are in flight together; the sums wait for them here
#pragma unroll
      for (int s = 0; s < CH; s++) { L.j[s] = (L.j[s] + j1[s]) + j2[s]; L.m[s] = (L.m[s] + m1[s]) + m2[s]; }
    }
  };

  // parity split of a register set into tile image `buf`
  auto park_chunk = [&](int buf, const Lines &L) {
    double *dE = smem + buf * (2 * LDS_ELEMS), *dO = dE + LDS_ELEMS;
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const int idx = ld_lds0 + s * LDS_QSTEP;
      const d2 rj = L.j[s], rm = L.m[s];
      d2 e, o;
      if (!JFAST) {
        const bool mid = 2 * (ld_b + s * QSTEP) == nn;                  // rm was left 0 there
        if (raw_in) { e = rj; o = rm; }                                 // already split: e_j = x_j, o_j = x_{n-j}
        else { e = rj + rm; o = rj - rm; }
        if (mid) o = d2{0.0, 0.0};
      } else {
        // rj = (x_j, x_{j+1}), rm = (x_{n-j-1}, x_{n-j}); point j+1 may be past the half (H odd) -> 0
        const bool v1 = 2 * ld_a + 1 < H;
        if (raw_in) { e = d2{rj.x, v1 ? rj.y : 0.0}; o = d2{rm.y, v1 ? rm.x : 0.0}; }
        else {
          e = d2{rj.x + rm.y, v1 ? rj.y + rm.x : 0.0};
          o = d2{rj.x - rm.y, v1 ? rj.y - rm.x : 0.0};
        }
      }
      lds_put2<!JFAST || (LDJ % 2 == 0)>(dE + idx, e);
      lds_put2<!JFAST || (LDJ % 2 == 0)>(dO + idx, o);
    }
  };

  // the first two tiles and the matrix fragments are requested together: one round trip for all three (IN_SUM3, whose loads are
  // summed as they land: fragments first, then the tile)
  double ae[KS], ao[KS];
  if (!SUM3) issue_loads(tile, true, LA);
  if (TWO) issue_loads(tile + t_step, tile + t_step < t_hi, LB);
#pragma unroll
  for (int s = 0; s < KS; s++) {
    ae[s] = p.fragE[((long)(mt * KS + s)) * 64 + lane];
    ao[s] = p.fragO[((long)(mt * KS + s)) * 64 + lane];
  }
  if (SUM3) issue_loads(tile, true, LA);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): retires the fragment loads in front of the loop (see sweep.hip)

  const int i0 = mt * 16 + (JFAST ? l16 : kq);
  const bool mul_on = ACC && (p.out_mode == OUT_MUL);     // out = operand * (alpha r): the operand rides the VecAXPY path
  const bool acc2_on = ACC && (p.out_mode == OUT_ACC2);   // out = (acc + acc2) + alpha r
  const bool acc_on = ACC && ((p.out_mode == OUT_ACC) || mul_on || acc2_on);
  const double alpha = p.alpha;

#ifdef CHEB_STAMPS
  unsigned long long st_pre = 0, st_chain = 0, st_post = 0, st_bar = 0, st_t0, st_t1, st_begin, st_loop, st_end, st_tiles = 0;
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(v) do { } while (0)
#endif
#ifdef CHEB_STAMPS
  STAMP(st_begin);
#endif
  park_chunk(0, LA);
  if (!TWO && tile + t_step < t_hi) issue_loads(tile + t_step, true, LA);   // one register set: the next request follows the split
  lds_barrier_v();
#ifdef CHEB_STAMPS
  STAMP(st_loop);
#endif

  // one tile: chain on image `cur`, results out; `issue_fn` asks for the tile after next at the top, `park_fn` splits the next
  // tile's lines into image cur ^ 1 after the chain (before the stores: its wait covers loads only)
  auto tile_body = [&](int cur, auto &&issue_fn, auto &&park_fn) {
    const double *sE = smem + cur * (2 * LDS_ELEMS), *sO = sE + LDS_ELEMS;
    const u32 t_o = tile / tpo, t_q0 = (tile - t_o * tpo) * NT;
#ifdef CHEB_STAMPS
    STAMP(st_t0); st_tiles++;
#endif
    issue_fn();
    const int nb = ng * 16;

    // After the lane exchange this lane owns, for rp = 0,1: accumulator row r = 2*rp + odd, and of it
    // the two adjacent columns (COLFAST) / points (JFAST) starting at the even lane of its pair.
    u32 a_hi[2], a_lo[2];      // element offsets of the two 16-B pieces (row i / mirror row n-i)
    bool ok_hi[2], ok_lo[2], fold[2];
    d2 acc_hi[2], acc_lo[2];
#pragma unroll
    for (int rp = 0; rp < 2; rp++) {
      const int r = 2 * rp + odd;
      if (!JFAST) {
        const u32 q = t_q0 + nb + l16e;
        const int i = i0 + 4 * r;
        const u32 b = t_o * lineLen + q;
        ok_hi[rp] = q < inner && i < H;
        ok_lo[rp] = ok_hi[rp] && (nn - i != i);
        fold[rp] = false;
        a_hi[rp] = b + (u32)i * inner;
        a_lo[rp] = b + (u32)(nn - i) * inner;
      } else {
        const u32 c = tile * NT + nb + 4 * r + kq;
        const int ie = mt * 16 + l16e;                   // points ie, ie+1 of line c; mirrors n-ie-1, n-ie
        const u32 b = (c < ncols ? c : 0u) * lineLen;
        ok_hi[rp] = c < ncols && ie < H;
        fold[rp] = ok_hi[rp] && (ie + 1 >= H);            // H odd: point ie+1 IS the mirror of ie
        ok_lo[rp] = ok_hi[rp] && !fold[rp];
        a_hi[rp] = b + (u32)ie;
        a_lo[rp] = b + (u32)(nn - ie - 1);
      }
      acc_hi[rp] = d2{0.0, 0.0}; acc_lo[rp] = d2{0.0, 0.0};
    }
    // the VecAXPY operand of this tile
    if (acc_on) {
#pragma unroll
      for (int rp = 0; rp < 2; rp++) {
        acc_hi[rp] = *(ok_hi[rp] ? (const d2 *)(p.acc + a_hi[rp]) : zero2);
        acc_lo[rp] = *(ok_lo[rp] ? (const d2 *)(p.acc + a_lo[rp]) : zero2);
        if (acc2_on) {
          acc_hi[rp] = acc_hi[rp] + *(ok_hi[rp] ? (const d2 *)(p.acc2 + a_hi[rp]) : zero2);
          acc_lo[rp] = acc_lo[rp] + *(ok_lo[rp] ? (const d2 *)(p.acc2 + a_lo[rp]) : zero2);
        }
      }
    }

#ifdef CHEB_STAMPS
    STAMP(st_t1); st_pre += st_t1 - st_t0;
#endif
    v4d ce = {0.0, 0.0, 0.0, 0.0}, co = {0.0, 0.0, 0.0, 0.0};
    {
      const int frag = JFAST ? (nb + l16) * LDJ + kq : kq * NT + ((nb + l16) ^ ((kq & 1) << 4));
      const double *fE = sE + frag, *fO = sO + frag;
      double fb[2][4];
      fb[0][0] = fE[0]; fb[0][1] = fE[KSTR]; fb[0][2] = fO[0]; fb[0][3] = fO[KSTR];
#pragma unroll
      for (int g = 0; g < KS / 2; g++) {
        const int cb = g & 1, nbuf = cb ^ 1;
        if (g + 1 < KS / 2) {
          fb[nbuf][0] = fE[(2 * g + 2) * KSTR]; fb[nbuf][1] = fE[(2 * g + 3) * KSTR];
          fb[nbuf][2] = fO[(2 * g + 2) * KSTR]; fb[nbuf][3] = fO[(2 * g + 3) * KSTR];
        }
        // fence: keep the fragment reads of group g+1 ABOVE the MFMAs of group g (hipcc otherwise sinks
        // them to just before their use and every group starts with an exposed LDS round trip)
        __builtin_amdgcn_sched_barrier(0);
        if (!JFAST) {
          ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[2 * g], fb[cb][0], ce, 0, 0, 0);
          co = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[2 * g], fb[cb][2], co, 0, 0, 0);
          ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[2 * g + 1], fb[cb][1], ce, 0, 0, 0);
          co = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[2 * g + 1], fb[cb][3], co, 0, 0, 0);
        } else {
          ce = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][0], ae[2 * g], ce, 0, 0, 0);
          co = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][2], ao[2 * g], co, 0, 0, 0);
          ce = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][1], ae[2 * g + 1], ce, 0, 0, 0);
          co = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][3], ao[2 * g + 1], co, 0, 0, 0);
        }
      }
    }
#ifdef CHEB_STAMPS
    STAMP(st_t0); st_chain += st_t0 - st_t1;
#endif
    park_fn();                                            // before the stores: the wait covers loads only

    // hi = value of row i, lo = value of the mirror row n-i  (D: b - a;  D D: a - b)
    double hi[4], lo[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (raw_out) { hi[r] = ce[r]; lo[r] = co[r]; }
      else { hi[r] = ce[r] + co[r]; lo[r] = p.sym ? ce[r] - co[r] : co[r] - ce[r]; }
    }
#pragma unroll
    for (int rp = 0; rp < 2; rp++) {
      // even lane keeps row 2rp and gets the neighbour's row 2rp; odd lane keeps row 2rp+1
      const double own_hi = odd ? hi[2 * rp + 1] : hi[2 * rp], snd_hi = odd ? hi[2 * rp] : hi[2 * rp + 1];
      const double own_lo = odd ? lo[2 * rp + 1] : lo[2 * rp], snd_lo = odd ? lo[2 * rp] : lo[2 * rp + 1];
      const double rcv_hi = swap1(snd_hi), rcv_lo = swap1(snd_lo);
      d2 vh = odd ? d2{rcv_hi, own_hi} : d2{own_hi, rcv_hi};   // ascending columns / points
      d2 vl;
      if (!JFAST) vl = odd ? d2{rcv_lo, own_lo} : d2{own_lo, rcv_lo};
      else vl = odd ? d2{own_lo, rcv_lo} : d2{rcv_lo, own_lo};  // mirrors of (ie, ie+1) are (n-ie, n-ie-1): descending
      if (JFAST && fold[rp]) vh = d2{vh.x, odd ? rcv_lo : own_lo};   // (y_ie, y_{n-ie}) : adjacent when H is odd
      if (!ACC) { vh = alpha * vh; vl = alpha * vl; }
      else if (mul_on) { vh = acc_hi[rp] * (alpha * vh); vl = acc_lo[rp] * (alpha * vl); }
      else { vh = acc_hi[rp] + alpha * vh; vl = acc_lo[rp] + alpha * vl; }
      if (ok_hi[rp]) *(d2 *)(p.out + a_hi[rp]) = vh;
      if (ok_lo[rp]) *(d2 *)(p.out + a_lo[rp]) = vl;
    }
#ifdef CHEB_STAMPS
    STAMP(st_t1); st_post += st_t1 - st_t0;
#endif
    lds_barrier_v();
#ifdef CHEB_STAMPS
    STAMP(st_t0); st_bar += st_t0 - st_t1;
#endif
  };

  const u32 t2 = 2 * t_step;
  if (!TWO) {
    // one register set: tile t + 2 is requested right after the split of tile t + 1 (it lands under the stores, the barrier and the next chain)
    for (int cur = 0; tile < t_hi; tile += t_step, cur ^= 1)
      tile_body(cur, [] {}, [&] {                          // (uniform branches: a workgroup's last tile asks for nothing -- IN_SUM3 would wait for it)
        if (tile + t_step < t_hi) park_chunk(cur ^ 1, LA);
        if (tile + t2 < t_hi) issue_loads(tile + t2, true, LA);
      });
  } else {
    for (;;) {
      tile_body(0, [&] { issue_loads(tile + t2, tile + t2 < t_hi, LA); }, [&] { park_chunk(1, LB); });
      tile += t_step; if (tile >= t_hi) break;
      tile_body(1, [&] { issue_loads(tile + t2, tile + t2 < t_hi, LB); }, [&] { park_chunk(0, LA); });
      tile += t_step; if (tile >= t_hi) break;
    }
  }
#ifdef CHEB_STAMPS
  STAMP(st_end);
  if (lane == 0 && p.in4) {
    unsigned long long *dbg = (unsigned long long *)p.in4 + ((size_t)BID * 8 + w) * 16;
    dbg[0] = st_pre; dbg[1] = st_chain; dbg[2] = st_post; dbg[3] = st_bar;
    dbg[4] = st_loop - st_begin; dbg[5] = st_end - st_loop; dbg[6] = st_begin; dbg[7] = st_end;
    // [8] first tiles and fragments landed, [9] tiles walked, [10] whole kernel in 100 MHz ticks, [11] in shader cycles
    dbg[8] = st_begin - st_entry; dbg[10] = __builtin_amdgcn_s_memrealtime() - rt_entry; dbg[11] = st_end - st_entry;
    dbg[9] = st_tiles;
  }
#endif
}

// (second launch bound: four waves per SIMD = two workgroups per CU, wgs_per_cu: at most 128 VGPRs)
template <int KS, bool JFAST, bool ACC>
__global__ __launch_bounds__(512, 4) void cheb_sweep_vec_kernel(const SweepParams p) {
  __shared__ double smem[vec_lds_doubles<KS, JFAST>()];
  vec1_body<KS, JFAST, false, ACC>(p, smem, blockIdx.x, gridDim.x);
}

// odd-half fragment s of a wave: a register, or (KS = 32: the last NFL of them) its slot in LDS
#define AO(s_) (((s_) < KR) ? ao[((s_) < KR) ? (s_) : 0] : aoL[((s_) - KR) * 64])

// Diagnostic builds only (make diag, tools/stamp_probe3.py): in-kernel cycle stamps kept in SGPRs
#ifdef CHEB_STAMPS
#define STAMP3(k_) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    st_seg[k_] += t_ - st_prev; st_prev = t_; } while (0)
#define STAMP3_MARK(v_) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); v_ = t_; st_prev = t_; } while (0)
#else
#define STAMP3(k_) do { } while (0)
#define STAMP3_MARK(v_) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// cheb_sweep_vec4_kernel: the kernel for lines of more than 64 points (KS >= 16) -- a STRAIGHT-LINE tile loop with as few
// VALU instructions as the algorithm allows.
//
// gfx950 retires loads and stores through one in-order counter (vmcnt).  hipcc places the waits, and it can only count
// exactly through straight-line code: a branch around a load or store (a "has next tile" test, an exec-masked store,
// the two in-chain positions of the wave groups) makes it fall back to vmcnt(0), which waits for EVERYTHING in flight --
// including the prefetch issued a few hundred cycles earlier.  Here the loop body has no branch around any memory
// instruction: the loop is instantiated once per wave group (GRPB), so the in-chain placement is static; STORE and ACC
// are separate instantiations; masking is done by the buffer range check.  With exact counts the operand pipeline can be
// deep: line chunks ride under two MFMA chains, the VecAXPY operand is requested one sub-tile ahead into its own register
// set (X / Y), and the wait for it does not cover the stores issued in between.  Per-array geometry (sweep.h): input,
// VecAXPY operand and output may have different row pitches.
//
// On gfx950 the FP64 MFMA and the VALU do not co-execute (SQ_VALU_MFMA_COEXEC_CYCLES reads 0 for every launch
// of this kernel, profiles/r02_*; the FP64 matrix rate equals the FP64 vector rate): every vector instruction
// of a wave is paid in matrix-pipe time, ~6 cycles each, and the flat-address predecessor of this kernel (round 1-2,
// deleted) issued 300-450 of them per tile and wave next to 128 MFMAs (in-kernel stamps: an epilogue beside the partner
// wave's chain took 4,400 cycles).  Here
//   * every global access is a raw BUFFER access: 32-bit byte offset = per-lane constant + one scalar per
//     tile (one v_add), the hardware range check replaces every `ok ? address : dummy` select (out-of-range
//     loads return 0, out-of-range stores are dropped; a masked lane gets offset 0x80000000);
//   * loads are not masked at all: lanes outside the tile read other points of the SAME line (or zeros past
//     the end of the array), which only meet zero entries of the matrix or feed columns that are never stored;
//   * the lane exchange that makes 16-byte pieces is one DPP broadcast + one select per dword;
//   * the centro-symmetry sign rides in the scalar factor of the mirror row.
// Diagnostic builds only (-DV4_ABLATE=bits; tools/v4_ablate.sh): 1 = no input loads, 2 = no accumulator loads, 8 = no global
// stores, 16 = no MFMA chains.  Results are wrong; the timing shows what each stream costs.
#ifndef V4_ABLATE
#define V4_ABLATE 0
#endif
// cache policy of the result stores (aux operand of the raw buffer store: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef V4_STORE_AUX
#define V4_STORE_AUX 0
#endif
// 1 (shipped): the matrix fragments are requested in the order the chains consume them and the first tile's chains
// start on the k-steps that have landed (no wait for the whole set); 0: the round 2-5 prologue (rotated fetch order,
// vmcnt(0) before the first tile) -- kept for the A/B builds of tools/v4_overlap_ab.sh
#ifndef V4_OVERLAP
#define V4_OVERLAP 1
#endif
// fragment pairs requested ahead of the first tile's first chain; the rest is requested INSIDE that chain, one pair
// per MFMA group, V4_FRAG_AHEAD groups ahead of its use (KS / 2 and more: all of them up front)
#ifndef V4_FRAG_AHEAD
#define V4_FRAG_AHEAD 2
#endif

// MODE: 0 = STORE, 1 = ACC (out = acc + alpha r), 3 = ACC2 (out = (acc + acc2) + alpha r), 2 = MUL (out = acc * (alpha r): OUT_MUL, the modal scaling of the
// preconditioner's fast diagonalisation folded into its last forward transform -- RAW = 1 only)
// INM: 1 = IN_MUL, every input element is multiplied by the element of in1 at the same place as it is split into LDS (the 1 / eta
// of the preconditioner's P_1^-1 (r / eta) folded into its first forward transform -- RAW = 1, STORE only)
// GATH: the rows of every line come from up to GATHER_MAX different arrays (sweep.h GatherSrc; the pencil of a slab partition read from the
// ranks' slabs in place, over xGMI for the remote ones): 64-bit global loads from per-thread row bases instead of the buffer loads --
// COLFAST, plain input, STORE only.  A thread's loader slots are the same rows for every tile, so their bases are found once.
template <int KS, bool JFAST, int MODE, int RAW = 0, int INM = 0, int GATH = 0>
__device__ __forceinline__ void vec4_body(const SweepParams &p, double *smem, const u32 BID, const u32 NBLK, const GatherSrc *gs = nullptr) {
  static_assert(!GATH || (!JFAST && MODE == 0 && RAW == 0 && INM == 0), "the gather loader exists for strided lines, plain input, STORE");
  constexpr bool PUSH = GATH == 2;                     // GATH = 2: the results are stored into the row owners' arrays as well (sweep.h GatherSrc::push)
  constexpr bool ACC = MODE != 0, MUL = MODE == 2, ACC2 = MODE == 3;   // ACC: the operand stream exists; ACC2: two of them, (acc + acc2) + alpha r
  static_assert(!ACC2 || RAW == 0, "OUT_ACC2 has no raw mode");
  // KS = 32 has no registers for a second operand set on top of the X / Y pair (238 VGPRs of 256 in ACC mode).  Its OUT_ACC2 keeps ONE
  // set of both operands instead: requested right after the stores of the previous sub-tile, i.e. one whole chain (~8 k cycles)
  // before the epilogue that consumes them -- the same 32 VGPRs the X / Y pair takes.
  constexpr bool ONEBUF = ACC2 && KS == 32;
  static_assert(INM == 0 || (RAW == 1 && MODE == 0), "IN_MUL exists for the raw forward transform with a plain store only");
  static_assert(RAW == 0 || MODE != 1, "the raw modes (sweep.h) are STORE / MUL only");
  static_assert(!MUL || RAW == 1, "OUT_MUL exists for the raw forward transform only");
  constexpr int MTP = KS / 4;
  constexpr int NG = 8 / MTP;
  constexpr int HP = 4 * KS;
  constexpr int NSUB = 2;
  constexpr int NT = 16 * NG * NSUB;
  constexpr int LDJ = HP + V_LDJ_PAD;
  constexpr int LDS_ELEMS = JFAST ? NT * LDJ : HP * NT;
  constexpr int ITEMS = HP * NT / 2 / 512;
  constexpr int CH = ITEMS / NSUB;
  constexpr int QSTEP = JFAST ? 512 / (HP / 2) : 512 / (NT / 2);
  constexpr int LDS_QSTEP = JFAST ? QSTEP * LDJ : QSTEP * NT;
  constexpr int KSTR = JFAST ? 4 : 4 * NT;
  constexpr int NFL = (KS == 32) ? (JFAST ? 7 : 8) : 0;
  constexpr int KR = KS - NFL;
  // offsets beyond every buffer (the launcher keeps them < 1 GiB): a per-lane constant may carry INVALID, the scalar
  // part of an offset T_INVALID, and their sum must not wrap back into range
  constexpr u32 INVALID = 0x80000000u, T_INVALID = 0x40000000u;
  static_assert(KS >= 16 && CH >= 1, "v4 needs two sub-tiles per tile");

#ifdef CHEB_STAMPS
  unsigned long long st_seg[5] = {0, 0, 0, 0, 0}, st_prev = 0, st_begin = 0, st_loop = 0, st_first = 0;   // diagnostic build only (tools/stamp_probe3.py)
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();                     // 100 MHz: the in-kernel clock is d(memtime) / d(memrealtime) x 100 MHz
#endif
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  STAMP3_MARK(st_begin);
  const int mt = w % MTP, ng = w / MTP;
  const int kq = lane >> 4, l16 = lane & 15;
  const bool odd = l16 & 1; const int l16e = l16 & ~1;
  const int nn = p.P - 1, H = p.H;
  const u32 qmax = p.qmax;

  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in0, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_in1 = __builtin_amdgcn_make_buffer_rsrc((void *)(INM ? p.in1 : p.in0), 0, INM ? p.in_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_acc = __builtin_amdgcn_make_buffer_rsrc((void *)(ACC ? p.acc : p.in0), 0, ACC ? p.acc_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_acc2 = __builtin_amdgcn_make_buffer_rsrc((void *)(ACC2 ? p.acc2 : p.in0), 0, ACC2 ? p.acc_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);
  auto ld16 = [](__amdgpu_buffer_rsrc_t r, u32 off) { return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); };
  auto st16 = [](__amdgpu_buffer_rsrc_t r, u32 off, d2 v) { if (!(V4_ABLATE & 8) || v.x == 1.2345e300) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, (int)off, 0, V4_STORE_AUX); };

  double ae[KS], ao[KR > 0 ? KR : 1];
  double *aoL = smem + 4 * LDS_ELEMS + (w * NFL) * 64 + lane;

  const u32 tpo = JFAST ? 1u : (qmax + NT - 1) / NT;
  const u32 nxcd = (NBLK % 8 == 0) ? 8u : 1u;
  const u32 t_per = (p.ntiles + nxcd - 1) / nxcd;
  const u32 t_lo = (BID % nxcd) * t_per;
  const u32 t_hi = (t_lo + t_per < p.ntiles) ? t_lo + t_per : p.ntiles;
  const u32 t_step = NBLK / nxcd;

  // loader slots as in v1: COLFAST (line pair 2*ld_a, point pair ld_b + s*QSTEP); JFAST (points 2*ld_a, 2*ld_a+1 of
  // line ld_b + s*QSTEP).  Per-lane byte offsets of slot 0 (point / mirror); a slot adds a scalar
  const int ld_a = JFAST ? tid % (HP / 2) : tid % (NT / 2);
  const int ld_b = JFAST ? tid / (HP / 2) : tid / (NT / 2);
  const int ld_lds0 = JFAST ? ld_b * LDJ + 2 * ld_a : ld_b * NT + ((2 * ld_a) ^ ((ld_b & 1) << 4));
  const u32 in_os8 = p.in_os * 8u, in_rs8 = p.in_rs * 8u;
  const u32 lj = JFAST ? (u32)ld_b * in_os8 + (u32)(2 * ld_a) * 8u : (u32)(2 * ld_a) * 8u + (u32)ld_b * in_rs8;
  const u32 lm = JFAST ? (u32)ld_b * in_os8 + (u32)(nn - 2 * ld_a - 1) * 8u : (u32)(2 * ld_a) * 8u + (u32)(nn - ld_b) * in_rs8;
  const u32 slot8 = JFAST ? (u32)QSTEP * in_os8 : (u32)QSTEP * in_rs8;      // byte step between two slots of a lane
  // byte offset of a tile's origin in an array of geometry (os8, rs8), or INVALID past the workgroup's last tile
  auto tile_off = [&](u32 tl, bool valid, u32 os8) -> u32 {
    if (JFAST) return valid ? tl * NT * os8 : T_INVALID;
    const u32 o = tl / tpo, q0 = (tl - o * tpo) * NT;
    return valid ? o * os8 + q0 * 8u : T_INVALID;
  };

  // ... of the INPUT, whose fields may be spaced out (sweep.h: in_fblocks / in_fskip)
  const u32 fb = p.in_fblocks, fskip8 = p.in_fskip * 8u;
  auto in_tile_off = [&](u32 tl) -> u32 {
    u32 off = tile_off(tl, true, in_os8);
    if (fb) off += __umulhi(JFAST ? tl * NT : tl / tpo, p.in_fblocks_inv) * fskip8;
    return off;
  };

  // GATH: byte address of (row, first column) in vector 0 of the array that holds the row, and that array's doubles per vector,
  // for the thread's ITEMS loader slots (row ld_b + sg QSTEP) and their mirrors
  typedef const d2 __attribute__((address_space(1))) *gd2p;
  unsigned long long gb_j[GATH ? ITEMS : 1], gb_m[GATH ? ITEMS : 1];
  u32 glq_j[GATH ? ITEMS : 1], glq_m[GATH ? ITEMS : 1];
  // PUSH: the same for the rows the lane STORES (accumulator rows 2 rp + odd of its 16-row block, and their mirrors), in the owners'
  // result arrays; psink: where a lane with nothing to store writes (the sweep's own output array: straight-line stores keep the
  // loop's wait counts exact)
  unsigned long long pb_hi[2] = {0, 0}, pb_lo[2] = {0, 0}, psink = 0;
  u32 pl_hi[2] = {0, 0}, pl_lo[2] = {0, 0};
  if constexpr (GATH) {
    // A table BY ROW in LDS (the second operand image, first written a whole tile later): thread i < P finds the owner of row i once,
    // without a search loop and without per-lane loads of the kernel argument -- the plane ranges are compared as scalars (uniform
    // indices: s_load) and the owner's entries selected on the way (ranges ascend: the last hit wins) -- then every thread looks its
    // eight rows up.  History (tools/stamp_probe_rank.py, profiles/r06_dist/rank_stamps.txt; the local jobs of the same launch reach
    // their first tile after 8.7 k cycles): a while loop over the ranges per row 15.0 k, eight compare chains with per-lane loads of
    // the owner's entries 12.1 k, with an LDS copy of the kernel argument's table 10.6 k.
    unsigned long long *gRB = (unsigned long long *)(smem + 2 * LDS_ELEMS);
    u32 *gRL = (u32 *)(gRB + 2 * HP);
    unsigned long long *gDB = (unsigned long long *)(gRL + 2 * HP);    // PUSH: the row's place in its owner's result array
    if (tid <= nn) {
      const int gG = gs->G;
      unsigned long long pb = (unsigned long long)gs->p[0], db = PUSH ? (unsigned long long)gs->dp[0] : 0ull; u32 lq = gs->lq[0]; int s0v = gs->s0[0], pmv = gs->pmax[0];
#pragma unroll
      for (int k = 1; k < GATHER_MAX; k++) {
        const bool c = (k < gG) && (tid >= gs->s0[k]);
        pb = c ? (unsigned long long)gs->p[k] : pb; lq = c ? gs->lq[k] : lq; s0v = c ? gs->s0[k] : s0v; pmv = c ? gs->pmax[k] : pmv;
        if constexpr (PUSH) db = c ? (unsigned long long)gs->dp[k] : db;
      }
      int pl = tid - s0v; if (pl > pmv - 1) pl = pmv - 1; if (pl < 0) pl = 0;
      const unsigned long long ro = ((unsigned long long)pl * gs->rowlen + gs->col0) * 8ull;
      gRB[tid] = pb + ro;
      gRL[tid] = lq;
      if constexpr (PUSH) gDB[tid] = db + ro;
    }
    __syncthreads();
    if constexpr (PUSH) {
#pragma unroll
      for (int rp = 0; rp < 2; rp++) {
        int i = mt * 16 + kq + 4 * (2 * rp + (odd ? 1 : 0)); if (i > nn) i = nn;       // (rows >= H are never stored: see o_hi / o_lo below)
        pb_hi[rp] = gDB[i]; pl_hi[rp] = gRL[i];
        pb_lo[rp] = gDB[nn - i]; pl_lo[rp] = gRL[nn - i];
      }
      psink = (unsigned long long)p.out + (unsigned long long)tid * 16ull;
    }
#pragma unroll
    for (int sg = 0; sg < ITEMS; sg++) {
      int row = ld_b + sg * QSTEP; if (row > nn) row = nn;       // (rows past the half meet zero columns of the matrix; they stay inside the line)
      gb_j[sg] = gRB[row]; glq_j[sg] = gRL[row];
      gb_m[sg] = gRB[nn - row]; glq_m[sg] = gRL[nn - row];
    }
  }
  d2 rjA[CH], rmA[CH], rjB[CH], rmB[CH];
  constexpr int ECH = INM ? CH : 1;
  d2 ejA[ECH], emA[ECH], ejB[ECH], emB[ECH];           // IN_MUL: the multipliers of chunk A / B, in flight beside them
  d2 accX_hi[2], accX_lo[2], accY_hi[2], accY_lo[2];
  d2 acc2X_hi[2], acc2X_lo[2], acc2Y_hi[2], acc2Y_lo[2];   // ACC2 only

  auto issue_loads = [&](u32 tl, bool valid, int chunk, d2 (&rj)[CH], d2 (&rm)[CH]) {
    if constexpr (GATH) {
      // a tile past the workgroup's last re-reads tile 0 (never used); lanes past the last column of the block re-read its last pair
      // (they feed columns that are never stored) -- every address stays inside the arrays
      const u32 tle = valid ? tl : 0u;
      const u32 gf0 = p.gfield0;
      const u32 o = tle / tpo, q0 = (tle - o * tpo) * NT;
      u32 qq = q0 + 2u * (u32)ld_a; if (qq > qmax - 2u) qq = qmax - 2u;
#pragma unroll
      for (int s = 0; s < CH; s++) {
        const int sg = chunk * CH + s;
        rj[s] = *(gd2p)(gb_j[sg] + ((unsigned long long)(o + gf0) * glq_j[sg] + qq) * 8ull);
        rm[s] = *(gd2p)(gb_m[sg] + ((unsigned long long)(o + gf0) * glq_m[sg] + qq) * 8ull);
      }
      return;
    }
    const u32 t0 = in_tile_off(tl);
    if constexpr (INM != 0) {
      d2 (&ej)[ECH] = (&rj == &rjA) ? ejA : ejB; d2 (&em)[ECH] = (&rj == &rjA) ? emA : emB;
#pragma unroll
      for (int s = 0; s < CH; s++) {
        const u32 so = (u32)(chunk * CH + s) * slot8;
        ej[s] = ld16(r_in1, lj + (valid ? t0 + so : T_INVALID));
        em[s] = ld16(r_in1, lm + (valid ? (JFAST ? t0 + so : t0 - so) : T_INVALID));
      }
    }
    if constexpr (V4_ABLATE & 1) { if (tl != p.ntiles + 12345u) {
#pragma unroll
      for (int s = 0; s < CH; s++) { rj[s] = d2{1.0 + s, 2.0}; rm[s] = d2{0.5, 0.25 * chunk}; }
      return; } }
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const u32 so = (u32)(chunk * CH + s) * slot8;                          // scalar
      rj[s] = ld16(r_in, lj + (valid ? t0 + so : T_INVALID));
      rm[s] = ld16(r_in, lm + (valid ? (JFAST ? t0 + so : t0 - so) : T_INVALID));
    }
  };
  const bool oddP = (p.P & 1) != 0;                    // a self-paired middle point exists (COLFAST only; JFAST needs even P)
  auto park_chunk = [&](int buf, int chunk, const d2 (&rj)[CH], const d2 (&rm)[CH]) {
    double *dE = smem + buf * (2 * LDS_ELEMS), *dO = dE + LDS_ELEMS;
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const int idx = ld_lds0 + (chunk * CH + s) * LDS_QSTEP;
      d2 e, o;
      d2 xj = rj[s], xm = rm[s];
      if constexpr (INM != 0) {
        const d2 (&ej)[ECH] = (&rj == &rjA) ? ejA : ejB; const d2 (&em)[ECH] = (&rj == &rjA) ? emA : emB;
        xj = xj * ej[s]; xm = xm * em[s];
      }
      if (RAW == 2) {                                  // already split: e_j = x_j, o_j = x_{n-j} (a middle o meets a zero column)
        e = xj;
        o = JFAST ? d2{xm.y, xm.x} : xm;
      } else if (!JFAST) {
        e = xj + xm;
        o = xj - xm;                                   // the middle point of an odd line is its own mirror: o = 0
        if (oddP) { const bool mid = 2 * (ld_b + (chunk * CH + s) * QSTEP) == nn; if (mid) e = xj; }
      } else {
        e = d2{xj.x + xm.y, xj.y + xm.x};
        o = d2{xj.x - xm.y, xj.y - xm.x};
      }
      lds_put2<!JFAST || (LDJ % 2 == 0)>(dE + idx, e);
      lds_put2<!JFAST || (LDJ % 2 == 0)>(dO + idx, o);
    }
  };

  // Output side.  After the lane exchange a lane owns, for rp = 0, 1: accumulator row r = 2 rp + odd, and of it the
  // two adjacent columns (COLFAST) / points (JFAST) starting at the even lane of its pair.  Per-lane byte offsets of
  // the two 16-B pieces (row i / mirror row n-i) relative to the sub-tile's origin; rows that do not exist are INVALID
  const int i0 = mt * 16 + (JFAST ? l16 : kq);
  u32 o_hi[2], o_lo[2], c_hi[2], c_lo[2];              // out / acc geometry
  bool fold[2] = {false, false};
#pragma unroll
  for (int rp = 0; rp < 2; rp++) {
    const int r = 2 * rp + (odd ? 1 : 0);
    if (!JFAST) {
      const int i = i0 + 4 * r;
      const bool ok = i < H, okl = ok & (nn - i != i);
      o_hi[rp] = ok ? (u32)l16e * 8u + (u32)i * (p.out_rs * 8u) : INVALID;
      o_lo[rp] = okl ? (u32)l16e * 8u + (u32)(nn - i) * (p.out_rs * 8u) : INVALID;
      c_hi[rp] = (u32)l16e * 8u + (u32)(ok ? i : 0) * (p.acc_rs * 8u);
      c_lo[rp] = (u32)l16e * 8u + (u32)(ok ? nn - i : 0) * (p.acc_rs * 8u);
    } else {
      const int ie = mt * 16 + l16e;                   // points ie, ie+1 of line 4 r + kq; mirrors n-ie-1, n-ie
      const bool ok = ie < H;
      fold[rp] = ok & (ie + 1 >= H);                   // H odd: point ie+1 IS the mirror of ie
      const bool okl = ok & !fold[rp];
      o_hi[rp] = ok ? (u32)(4 * r + kq) * (p.out_os * 8u) + (u32)ie * 8u : INVALID;
      o_lo[rp] = okl ? (u32)(4 * r + kq) * (p.out_os * 8u) + (u32)(nn - ie - 1) * 8u : INVALID;
      c_hi[rp] = (u32)(4 * r + kq) * (p.acc_os * 8u) + (u32)(ok ? ie : 0) * 8u;
      c_lo[rp] = (u32)(4 * r + kq) * (p.acc_os * 8u) + (u32)(ok ? nn - ie - 1 : 0) * 8u;
    }
  }
  const u32 out_os8 = p.out_os * 8u, acc_os8 = p.acc_os * 8u;
  const u32 sub8_out = JFAST ? 16u * out_os8 : 16u * 8u, sub8_acc = JFAST ? 16u * acc_os8 : 16u * 8u;   // sub-tile 1 vs 0
  const u32 ng8_out = (u32)ng * NSUB * sub8_out, ng8_acc = (u32)ng * NSUB * sub8_acc;
  const double alpha = p.alpha, alpha_lo = (p.sym || RAW == 1) ? p.alpha : -p.alpha;      // mirror row: D: b - a; D D: a - b

  auto acc_issue = [&](u32 tl, bool valid, int sub, d2 (&ah)[2], d2 (&al)[2]) {
    if (!ACC) return;
    if constexpr (V4_ABLATE & 2) { ah[0] = d2{1.0, 2.0}; ah[1] = ah[0]; al[0] = d2{3.0, 4.0}; al[1] = al[0]; if (tl != p.ntiles + 12345u) return; }
    const u32 t0 = tile_off(tl, valid, acc_os8) + ng8_acc + (u32)sub * sub8_acc;
#pragma unroll
    for (int rp = 0; rp < 2; rp++) { ah[rp] = ld16(r_acc, c_hi[rp] + t0); al[rp] = ld16(r_acc, c_lo[rp] + t0); }
    if constexpr (ACC2) {
      d2 (&bh)[2] = (&ah == &accX_hi) ? acc2X_hi : acc2Y_hi; d2 (&bl)[2] = (&ah == &accX_hi) ? acc2X_lo : acc2Y_lo;
#pragma unroll
      for (int rp = 0; rp < 2; rp++) { bh[rp] = ld16(r_acc2, c_hi[rp] + t0); bl[rp] = ld16(r_acc2, c_lo[rp] + t0); }
    }
  };

  u32 tile = t_lo + BID / nxcd;
  if (tile >= t_hi) return;                            // whole workgroup: no barrier is skipped by part of it
  // the first tile's lines are requested BEFORE the matrix fragments: one memory round trip instead of two
  issue_loads(tile, true, 0, rjA, rmA); issue_loads(tile, true, 1, rjB, rmB);
  // The matrix halves: 256 KiB per CU at P = 256.  A CU reads from its L2 at ~25-30 B/clk (MI355X_MICROARCH.md "Indexed
  // rows": 66-73 GB/s per CU), so the set takes ~10-12 k cycles however it is asked for, and a wave that asks for all of
  // it at once sits in the ISSUE of those loads for that long (in-kernel stamps, profiles/r06_prologue_stamps.txt: the
  // 36 requests of a wave were issued after 11.6 k cycles, landed after 12.1 k) -- longer than an MFMA chain.  So the
  // fetch is PACED: up front only what the first chain's first groups need (FA pairs) and the odd-half fragments that
  // live in LDS (their ds_write needs them early; temporaries that are dead before the tile loop), behind the first
  // tile's lines; the other pairs are requested inside the first tile's first chain, one pair per MFMA group, FA groups
  // ahead of their use.  The first tile is a peeled copy of the loop body (`run`): in its straight-line code hipcc counts
  // vmcnt exactly, so MFMA group g waits for pair g only.  (Rounds 2-5, V4_OVERLAP = 0: everything up front in a rotated
  // order per CU, vmcnt(0).)
  constexpr int FA = (V4_FRAG_AHEAD < KS / 2) ? V4_FRAG_AHEAD : KS / 2;
  // pair g of both halves -> registers (the odd half's LDS-resident members have been dealt with up front)
  auto frag_pair = [&](int g) {
    const d2 ve = ((const d2 *)p.fragE2)[((long)(mt * (KS / 2) + g)) * 64 + lane];
    ae[2 * g] = ve.x; ae[2 * g + 1] = ve.y;
    if (2 * g + 1 < KR) {
      const d2 vo = ((const d2 *)p.fragO2)[((long)(mt * (KS / 2) + g)) * 64 + lane];
      ao[2 * g] = vo.x; ao[2 * g + 1] = vo.y;
    }
  };
  if constexpr (V4_OVERLAP != 0) {
    constexpr int GL = KR / 2;                         // first pair with an LDS-resident member
    constexpr int NTL = (NFL > 0) ? KS / 2 - GL : 1;
    d2 tl[NTL];
    if constexpr (NFL > 0) {
#pragma unroll
      for (int g = GL; g < KS / 2; g++) tl[g - GL] = ((const d2 *)p.fragO2)[((long)(mt * (KS / 2) + g)) * 64 + lane];
    }
#pragma unroll
    for (int g = 0; g < FA; g++) frag_pair(g);
    if constexpr (NFL > 0) {
#pragma unroll
      for (int g = GL; g < KS / 2; g++) {
        if (2 * g >= KR) aoL[(2 * g - KR) * 64] = tl[g - GL].x; else ao[2 * g] = tl[g - GL].x;
        if (2 * g + 1 >= KR) aoL[(2 * g + 1 - KR) * 64] = tl[g - GL].y; else ao[2 * g + 1] = tl[g - GL].y;
      }
    }
  } else {
  // Every workgroup of the chip fetches the same 256 KiB at the same moment.  The CUs of an XCD start at four
  // different places of their fragment sets (a static rotation per code path: the registers are fixed), which
  // spreads the requests over the L2 channels instead of queueing them on one line at a time.
  auto load_frags = [&](auto ROT_) {
    constexpr int ROT = decltype(ROT_)::value;
#pragma unroll
    for (int g0 = 0; g0 < KS / 2; g0++) {              // two fragments per 16-byte load
      const int g = (g0 + ROT) % (KS / 2);
      const d2 ve = ((const d2 *)p.fragE2)[((long)(mt * (KS / 2) + g)) * 64 + lane];
      const d2 vo = ((const d2 *)p.fragO2)[((long)(mt * (KS / 2) + g)) * 64 + lane];
      ae[2 * g] = ve.x; ae[2 * g + 1] = ve.y;
      if (2 * g < KR) ao[2 * g] = vo.x; else aoL[(2 * g - KR) * 64] = vo.x;
      if (2 * g + 1 < KR) ao[2 * g + 1] = vo.y; else aoL[(2 * g + 1 - KR) * 64] = vo.y;
    }
  };
  switch ((BID / nxcd) & 3u) {
    case 0: load_frags(std::integral_constant<int, 0>{}); break;
    case 1: load_frags(std::integral_constant<int, KS / 8>{}); break;
    case 2: load_frags(std::integral_constant<int, KS / 4>{}); break;
    default: load_frags(std::integral_constant<int, 3 * KS / 8>{}); break;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): see sweep.hip
  }
#ifdef CHEB_STAMPS
  unsigned long long st_wait = 0, st_park = 0;
  STAMP3_MARK(st_wait);
#endif
  park_chunk(0, 0, rjA, rmA); park_chunk(0, 1, rjB, rmB);
#ifdef CHEB_STAMPS
  STAMP3_MARK(st_park);
#endif

  auto chain = [&](const double *sE, const double *sO, int sub, int g_issue, int g_park, v4d &ce, v4d &co, auto &&issue_fn, auto &&park_fn, auto &&grp_fn) {
    const int nb = (ng * NSUB + sub) * 16;
    ce = v4d{0.0, 0.0, 0.0, 0.0}; co = v4d{0.0, 0.0, 0.0, 0.0};
    const int frag = JFAST ? (nb + l16) * LDJ + kq : kq * NT + ((nb + l16) ^ ((kq & 1) << 4));
    const double *fE = sE + frag, *fO = sO + frag;
    if constexpr (V4_ABLATE & 16) {                    // no matrix work: what the memory streams cost alone
      issue_fn(); park_fn();
#pragma unroll
      for (int g = 0; g < KS / 2; g++) grp_fn(g);
      ce[0] = fE[0] + ae[0]; co[0] = fO[KSTR] + AO(KS - 1); ce[1] = fE[2 * KSTR]; co[2] = fO[3 * KSTR];
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
      grp_fn(g);
      if (g == g_issue) issue_fn();
      if (g == g_park) park_fn();
      if (!JFAST) {
        ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[2 * g], fb[cb][0], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f64_16x16x4f64(AO(2 * g), fb[cb][2], co, 0, 0, 0);
        ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[2 * g + 1], fb[cb][1], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f64_16x16x4f64(AO(2 * g + 1), fb[cb][3], co, 0, 0, 0);
      } else {
        ce = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][0], ae[2 * g], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][2], AO(2 * g), co, 0, 0, 0);
        ce = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][1], ae[2 * g + 1], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][3], AO(2 * g + 1), co, 0, 0, 0);
      }
    }
  };
  // value of the even lane of each pair / of the odd lane, in both lanes (DPP quad_perm [0,0,2,2] / [1,1,3,3])
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
  // out = (acc +) alpha * (sums) for sub-tile `sub` of tile tl
  auto epilogue = [&](u32 tl, int sub, const v4d &ce, const v4d &co, const d2 (&acc_hi)[2], const d2 (&acc_lo)[2]) {
    u32 t0 = tile_off(tl, true, out_os8) + ng8_out + (u32)sub * sub8_out;
    u32 t0v = t0;                                      // per-lane: INVALID where the tile's last column block ends early
    if (!JFAST) {
      const u32 o = tl / tpo, q = (tl - o * tpo) * NT + (ng * NSUB + sub) * 16 + l16e;
      t0v = (q < qmax) ? t0 : T_INVALID;
    }
    double hi[4], lo[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (RAW == 1) { hi[r] = ce[r]; lo[r] = co[r]; }  // the halves as they are: row i <- (ME e)_i, row n-i <- (MO o)_i
      else { hi[r] = ce[r] + co[r]; lo[r] = ce[r] - co[r]; }
    }
#pragma unroll
    for (int rp = 0; rp < 2; rp++) {
      // even lane: row 2rp of its own column and of the odd neighbour's; odd lane: row 2rp+1 of the even neighbour's and its own
      const double ha = hi[2 * rp], hb = hi[2 * rp + 1], la = lo[2 * rp], lb = lo[2 * rp + 1];
      // the broadcasts run with EVERY lane active (a DPP read of a lane that EXEC has switched off returns 0), the
      // selects come after
      const double hbe = bc_even(hb), hao = bc_odd(ha), lbe = bc_even(lb), lao = bc_odd(la);
      d2 vh = d2{odd ? hbe : ha, odd ? hb : hao};                         // ascending columns / points
      d2 vl;
      if (!JFAST) vl = d2{odd ? lbe : la, odd ? lb : lao};
      else vl = d2{odd ? lb : lao, odd ? lbe : la};                       // mirrors of (ie, ie+1) are (n-ie, n-ie-1): descending
      if (MUL) { vh = acc_hi[rp] * (alpha * vh); vl = acc_lo[rp] * (alpha_lo * vl); }
      else if (ACC2) {
        const d2 (&bh)[2] = (&acc_hi == &accX_hi) ? acc2X_hi : acc2Y_hi; const d2 (&bl)[2] = (&acc_hi == &accX_hi) ? acc2X_lo : acc2Y_lo;
        vh = (acc_hi[rp] + bh[rp]) + alpha * vh; vl = (acc_lo[rp] + bl[rp]) + alpha_lo * vl;
      }
      else if (ACC) { vh = acc_hi[rp] + alpha * vh; vl = acc_lo[rp] + alpha_lo * vl; }
      else { vh = alpha * vh; vl = alpha_lo * vl; }
      if (JFAST && (H & 1)) { if (fold[rp]) vh = d2{vh.x, vl.y}; }       // (y_ie, y_{n-ie}): adjacent when H is odd
      if constexpr (PUSH) {
        // row i of this output line belongs to the rank that owns plane i: the 16 bytes go into ITS result array, where the row
        // sits as in the array the line was read from -- remote stores inside the same launch as the remote loads (the two
        // directions of a link at once); the final sum then reads local memory only (dist.hip)
        typedef d2 __attribute__((address_space(1))) *gd2w;
        const u32 o = tl / tpo, q = (tl - o * tpo) * NT + (ng * NSUB + sub) * 16 + l16e;
        const bool cv = q < qmax;
        const unsigned long long of = (unsigned long long)(o + p.gfield0);   // (a field keeps its index on the way back: slabx.hip)
        const unsigned long long ah = (cv && o_hi[rp] != INVALID) ? pb_hi[rp] + (of * pl_hi[rp] + q) * 8ull : psink;
        const unsigned long long al = (cv && o_lo[rp] != INVALID) ? pb_lo[rp] + (of * pl_lo[rp] + q) * 8ull : psink;
        *(gd2w)ah = vh;
        *(gd2w)al = vl;
      } else {
        st16(r_out, o_hi[rp] + t0v, vh);
        st16(r_out, o_lo[rp] + t0v, vl);
      }
    }
  };

  auto run = [&](auto GRPB_) {
    constexpr bool GRPB = decltype(GRPB_)::value;
    constexpr int G_ISSUE = GRPB ? 0 : KS / 8, G_PARK = GRPB ? KS / 4 : 3 * KS / 8;
    {
      // reproduce the in-flight state of the loop's back edge (operand X, chunk A, four stores) so
      // that the wait counts of the loop hold from iteration 0
      const u32 nx = tile + t_step;
      if constexpr (!ONEBUF) acc_issue(tile, true, 0, accX_hi, accX_lo);
      issue_loads(nx, nx < t_hi, 0, rjA, rmA);
#pragma unroll
      for (int q = 0; q < 4; q++) st16(r_out, INVALID, d2{0.0, 0.0});
      if constexpr (ONEBUF) acc_issue(tile, true, 0, accX_hi, accX_lo);
    }
    lds_barrier_v();
    STAMP3_MARK(st_loop);
    int cur = 0;
    auto tile_body = [&](auto FIRST_) {
      constexpr bool FIRST = decltype(FIRST_)::value;
      const u32 nxt = tile + t_step, nxt2 = nxt + t_step;
      const bool v1 = nxt < t_hi, v2 = nxt2 < t_hi;
      const double *sE = smem + cur * (2 * LDS_ELEMS), *sO = sE + LDS_ELEMS;
      v4d ce, co;
      chain(sE, sO, 0, G_ISSUE, G_PARK, ce, co,
            [&] { if constexpr (!ONEBUF) acc_issue(tile, true, 1, accY_hi, accY_lo); issue_loads(nxt, v1, 1, rjB, rmB); },
            [&] { park_chunk(cur ^ 1, 0, rjA, rmA); },
            [&](int g) { if constexpr (FIRST) { if (g + FA < KS / 2) frag_pair(g + FA); } });
      STAMP3(0);
      epilogue(tile, 0, ce, co, accX_hi, accX_lo);
      if constexpr (ONEBUF) acc_issue(tile, true, 1, accX_hi, accX_lo);      // the set is free again: sub-tile 1's operands ride under chain 1
      STAMP3(1);
      chain(sE, sO, 1, G_ISSUE, G_PARK, ce, co,
            [&] { if constexpr (!ONEBUF) acc_issue(nxt, v1, 0, accX_hi, accX_lo); issue_loads(nxt2, v2, 0, rjA, rmA); },
            [&] { park_chunk(cur ^ 1, 1, rjB, rmB); }, [](int) {});
      STAMP3(2);
      if constexpr (ONEBUF) { epilogue(tile, 1, ce, co, accX_hi, accX_lo); acc_issue(nxt, v1, 0, accX_hi, accX_lo); }
      else epilogue(tile, 1, ce, co, accY_hi, accY_lo);
      STAMP3(3);
      lds_barrier_v();
      STAMP3(4);
      cur ^= 1;
    };
    if constexpr (V4_OVERLAP != 0) {
      // the first tile, peeled: the same body in straight-line code behind the fragment requests, so that every MFMA
      // group waits for its own fragments only; the loop below then starts from the state of its own back edge
      tile_body(std::true_type{});
      tile += t_step;
      STAMP3_MARK(st_first);
    }
#pragma unroll 1
    for (; tile < t_hi; tile += t_step) tile_body(std::false_type{});
  };
  if (w >= 4) run(std::true_type{}); else run(std::false_type{});
#ifdef CHEB_STAMPS
  {
    unsigned long long st_end; STAMP3_MARK(st_end);
    if (lane == 0 && p.in4) {
      unsigned long long *dbg = (unsigned long long *)p.in4 + ((size_t)BID * 8 + w) * 16;
      dbg[0] = st_seg[0]; dbg[1] = st_seg[1]; dbg[2] = st_seg[2]; dbg[3] = st_seg[3]; dbg[4] = st_seg[4];
      // [5] prologue, [6] whole kernel in shader cycles, [7] whole kernel in 100 MHz ticks (MI355X_MICROARCH.md, DVFS note 6)
      dbg[5] = st_loop - st_begin; dbg[6] = st_end - st_begin; dbg[7] = __builtin_amdgcn_s_memrealtime() - rt_begin;
      // [8] fragments requested (V4_OVERLAP) / landed (0), [9] first tile's lines parked, [10] first tile done (V4_OVERLAP), [11] / [12] begin / end (absolute, s_memtime)
      dbg[8] = st_wait - st_begin; dbg[9] = st_park - st_begin; dbg[10] = st_first ? st_first - st_begin : 0; dbg[11] = st_begin; dbg[12] = st_end;
    }
  }
#endif
}

template <int KS, bool JFAST, int MODE, int RAW = 0, int INM = 0>
__global__ __launch_bounds__(512) void cheb_sweep_vec4_kernel(const SweepParams p) {
  __shared__ double smem[vec_lds_doubles<KS, JFAST>()];
  vec4_body<KS, JFAST, MODE, RAW, INM>(p, smem, blockIdx.x, gridDim.x);
}

template <int KS, bool PUSH>
__global__ __launch_bounds__(512) void cheb_sweep_vec4_gather_kernel(const SweepParams p, const GatherSrc g) {
  __shared__ double smem[vec_lds_doubles<KS, false>()];
  vec4_body<KS, false, 0, 0, 0, PUSH ? 2 : 1>(p, smem, blockIdx.x, gridDim.x, &g);
}

// ---------------------------------------------------------------------------------------------
// Multi-job launch: up to MULTI_MAX independent plain STORE sweeps with the same KS in ONE launch.  The workgroups
// [bstart[j], bstart[j+1]) run job j with the single-launch code; a job's share is a multiple of 8 workgroups so
// that its XCD-aware tile walk holds.  Small grids are bound by the fixed cost of a launch (~5 us at 64^3), not
// by its work: the d gradient sweeps of a Stokes callback, its d divergence sweeps and its d pressure-gradient
// sweeps are one launch each this way (stokes.hip); on one stream the gradient and pressure-gradient sweeps share a launch.
constexpr int MULTI_MAX = 9;
struct MultiParams { int njobs; unsigned bstart[MULTI_MAX + 1]; SweepParams job[MULTI_MAX]; };

template <int KS, bool SUM3 = false>
__global__ __launch_bounds__(512, (KS <= 8 ? 4 : 2)) void cheb_sweep_multi_kernel(const MultiParams mp) {
  constexpr int LDS = vec_lds_doubles<KS, true>() > vec_lds_doubles<KS, false>() ? vec_lds_doubles<KS, true>() : vec_lds_doubles<KS, false>();
  __shared__ double smem[LDS];
  int j = 0;
  while (j + 1 < mp.njobs && blockIdx.x >= mp.bstart[j + 1]) j++;
  const SweepParams &p = mp.job[j];
  const u32 bid = blockIdx.x - mp.bstart[j], nblk = mp.bstart[j + 1] - mp.bstart[j];
  if (p.inner < 16) {
    if constexpr (KS >= 16) vec4_body<KS, true, 0>(p, smem, bid, nblk); else vec1_body<KS, true, SUM3, false>(p, smem, bid, nblk);
  } else {
    if constexpr (KS >= 16) vec4_body<KS, false, 0>(p, smem, bid, nblk); else vec1_body<KS, false, SUM3, false>(p, smem, bid, nblk);
  }
}

// ... with ONE of the jobs reading its lines from the arrays of a GatherSrc (lines of more than 64 points): the three directions of a
// slab rank's matvec -- the local ones and the pencil direction over the peers' slabs -- as one launch (dist.hip)
template <int KS, bool PUSH>
__global__ __launch_bounds__(512) void cheb_sweep_multi_gather_kernel(const MultiParams mp, const GatherSrc g, const unsigned gmask) {
  static_assert(KS >= 16, "the gather loader is part of the long-line kernel");
  constexpr int LDS = vec_lds_doubles<KS, true>() > vec_lds_doubles<KS, false>() ? vec_lds_doubles<KS, true>() : vec_lds_doubles<KS, false>();
  __shared__ double smem[LDS];
  int j = 0;
  while (j + 1 < mp.njobs && blockIdx.x >= mp.bstart[j + 1]) j++;
  const SweepParams &p = mp.job[j];
  const u32 bid = blockIdx.x - mp.bstart[j], nblk = mp.bstart[j + 1] - mp.bstart[j];
  if ((gmask >> j) & 1u) vec4_body<KS, false, 0, 0, 0, PUSH ? 2 : 1>(p, smem, bid, nblk, &g);
  else if (p.inner < 16) vec4_body<KS, true, 0>(p, smem, bid, nblk);
  else vec4_body<KS, false, 0>(p, smem, bid, nblk);
}

template <int KS, bool JFAST>
static hipError_t launch_v4(const SweepParams &p, unsigned grid, hipStream_t stream) {
  if (p.in_mode == IN_MUL) {
    if (p.raw != 1 || p.out_mode != OUT_STORE) return hipErrorInvalidValue;
    hipLaunchKernelGGL((cheb_sweep_vec4_kernel<KS, JFAST, 0, 1, 1>), dim3(grid), dim3(512), 0, stream, p);
  }
  else if (p.out_mode == OUT_MUL) {
    if (p.raw != 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((cheb_sweep_vec4_kernel<KS, JFAST, 2, 1>), dim3(grid), dim3(512), 0, stream, p);
  }
  else if (p.out_mode == OUT_ACC2) {
    if (p.raw) return hipErrorInvalidValue;
    hipLaunchKernelGGL((cheb_sweep_vec4_kernel<KS, JFAST, 3>), dim3(grid), dim3(512), 0, stream, p);
  }
  else if (p.out_mode == OUT_ACC) hipLaunchKernelGGL((cheb_sweep_vec4_kernel<KS, JFAST, 1>), dim3(grid), dim3(512), 0, stream, p);
  else if (p.raw == 1) hipLaunchKernelGGL((cheb_sweep_vec4_kernel<KS, JFAST, 0, 1>), dim3(grid), dim3(512), 0, stream, p);
  else if (p.raw == 2) hipLaunchKernelGGL((cheb_sweep_vec4_kernel<KS, JFAST, 0, 2>), dim3(grid), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL((cheb_sweep_vec4_kernel<KS, JFAST, 0>), dim3(grid), dim3(512), 0, stream, p);
  sweep_note_launch();
  return hipGetLastError();
}

// Geometry defaults, tile count and (KS >= 16) the byte sizes of the buffer descriptors.  Returns 4: cheb_sweep_vec4_kernel
// runs the launch; 1: cheb_sweep_vec_kernel (lines of at most 64 points, dense geometry); 0: neither (per-array geometry
// on short lines, or an array of 0.94 GB and more -- the 32-bit buffer offsets of the long-line kernel do not reach;
// the general kernel of sweep.hip runs those).
template <int KS, bool JFAST>
static int prepare_v(SweepParams &p) {
  constexpr int MTP = KS / 4, NG = 8 / MTP, NSUB = (KS >= 16) ? 2 : 1, NT = 16 * NG * NSUB;
  const bool custom = p.qmax != 0 || p.in_os != 0;         // per-array geometry given by the caller (KS >= 16 only)
  if (JFAST) {
    if (!p.in_os) p.in_os = (unsigned)p.P;
    if (!p.acc_os) p.acc_os = (unsigned)p.P;
    if (!p.out_os) p.out_os = (unsigned)p.P;
    p.in_rs = p.acc_rs = p.out_rs = 1; p.nouter = p.ncols; p.qmax = 1;
    p.ntiles = (p.ncols + NT - 1) / NT;
  } else {
    if (!p.qmax) { p.qmax = p.inner; p.nouter = p.ncols / p.inner; }
    if (!p.in_os) { p.in_os = (unsigned)p.P * p.inner; p.in_rs = p.inner; }
    if (!p.acc_os) { p.acc_os = (unsigned)p.P * p.inner; p.acc_rs = p.inner; }
    if (!p.out_os) { p.out_os = (unsigned)p.P * p.inner; p.out_rs = p.inner; }
    p.ntiles = p.nouter * ((p.qmax + NT - 1) / NT);
  }
  if constexpr (KS >= 16) {
    if (!p.sink) return 0;
    // byte sizes of the three arrays for the buffer descriptors (exact: the range check is the mask)
    const unsigned long long P_ = (unsigned long long)p.P;
    auto span = [&](unsigned os, unsigned rs) -> unsigned long long {
      if (JFAST) return ((unsigned long long)(p.ncols - 1) * os + P_) * 8ull;
      return ((unsigned long long)(p.nouter - 1) * os + (p.qmax - 1) + (P_ - 1) * rs + 1) * 8ull;
    };
    unsigned long long bi = span(p.in_os, p.in_rs);
    const unsigned long long ba = span(p.acc_os, p.acc_rs), bo = span(p.out_os, p.out_rs);
    if (p.in_fblocks) {                                    // spaced-out input fields (sweep.h)
      const unsigned units = JFAST ? p.ncols : p.nouter;
      if (units % p.in_fblocks || (JFAST && p.in_fblocks % NT) || p.out_mode != OUT_STORE || p.raw) return 0;
      bi += (unsigned long long)(units / p.in_fblocks - 1) * p.in_fskip * 8ull;
      p.in_fblocks_inv = (unsigned)((0x100000000ull + p.in_fblocks - 1) / p.in_fblocks);
      if ((unsigned long long)(JFAST ? p.ncols + NT : p.nouter) * p.in_fblocks >= 0x100000000ull) return 0;   // exactness of the reciprocal
    }
    if (!(bi < 0x38000000ull && ba < 0x38000000ull && bo < 0x38000000ull)) return 0;   // < 1 GiB minus slack: see T_INVALID
    p.in_bytes = (unsigned)bi; p.acc_bytes = (unsigned)ba; p.out_bytes = (unsigned)bo;
    return 4;
  }
  return custom ? 0 : 1;
}

// Workgroups of a persistent launch per CU.  Lines of at most 64 points (KS <= 8) need 64-70 KiB of LDS and at most 124
// VGPRs per workgroup: two fit on a CU, and at such sizes (64^3: a few tiles per workgroup) one workgroup's memory round
// trips then overlap the other's MFMA chains, and the tiles divide more evenly over the walkers.
template <int KS> constexpr int wgs_per_cu() { return KS <= 8 ? 2 : 1; }

template <int KS, bool JFAST>
static hipError_t launch_v(const SweepParams &p0, hipStream_t stream) {
  SweepParams p = p0;
  const int gen = prepare_v<KS, JFAST>(p);
  if (gen == 0) return hipErrorInvalidValue;             // sweep_vec_eligible said otherwise
  hipError_t cu_err; int ncu = sweep_num_cus(&cu_err);
  if (cu_err != hipSuccess) return cu_err;
  ncu *= wgs_per_cu<KS>();
  const unsigned grid = p.ntiles < (unsigned)ncu ? p.ntiles : (unsigned)ncu;
  if (grid == 0) return hipSuccess;
  if constexpr (KS >= 16) return launch_v4<KS, JFAST>(p, grid, stream);
  else {
    if (p.raw && p.out_mode != OUT_STORE && !(p.out_mode == OUT_MUL && p.raw == 1)) return hipErrorInvalidValue;   // (OUT_ACC2 with raw: refused by sweep_vec_eligible)
    if (p.out_mode == OUT_MUL && p.raw != 1) return hipErrorInvalidValue;
    if (p.in_mode == IN_MUL) return hipErrorInvalidValue;
    if (p.out_mode == OUT_STORE) hipLaunchKernelGGL((cheb_sweep_vec_kernel<KS, JFAST, false>), dim3(grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((cheb_sweep_vec_kernel<KS, JFAST, true>), dim3(grid), dim3(512), 0, stream, p);
    sweep_note_launch();
    return hipGetLastError();
  }
}

template <int KS>
static bool prepare_ok_t(SweepParams p, bool jfast) { return (jfast ? prepare_v<KS, true>(p) : prepare_v<KS, false>(p)) != 0; }

bool sweep_vec_eligible(const DiffMat &m, const SweepParams &p0) {
  const SweepParams &p = p0;
  if (p.in_mode == IN_SUM3) {                            // multi-job launches of short lines only (launch_multi_t)
    auto al16 = [](const void *q) { return q && ((size_t)q & 15) == 0; };
    if (m.KS > 8 || p.out_mode != OUT_STORE || p.raw || p.in_fblocks || p.qmax || p.in_os || !al16(p.in1) || !al16(p.in2)) return false;
  } else
  if (p.in_mode == IN_MUL) {                             // lines of more than 64 points, raw forward transform, plain store, dense geometry
    if (m.KS < 16 || p.raw != 1 || p.out_mode != OUT_STORE || p.in_fblocks || p.qmax || p.in_os || !p.in1 || ((size_t)p.in1 & 15)) return false;
  } else
  if (p.out_mode == OUT_ACC2) {                          // plain input, no raw mode
    if (p.in_mode != IN_PLAIN || p.raw || p.in_fblocks || !p.acc || !p.acc2 || ((size_t)p.acc2 & 15)) return false;
  } else
  if (p.in_mode != IN_PLAIN || (p.out_mode != OUT_STORE && p.out_mode != OUT_ACC && !(p.out_mode == OUT_MUL && p.raw == 1))) return false;
  const bool jfast = p.inner < 16;
  if (p.in_fblocks && (m.KS < 16 || (p.in_fskip & 1))) return false;
  if (p.qmax != 0 || p.in_os != 0) {                     // per-array geometry: the long-line kernel only, every offset must stay 16-B aligned
    if (m.KS < 16) return false;
    const unsigned all = p.qmax | p.in_os | p.in_rs | p.acc_os | p.acc_rs | p.out_os | p.out_rs;
    if (jfast ? ((p.in_os | p.acc_os | p.out_os) & 1) : (all & 1)) return false;
  }
  if (jfast && (p.inner != 1 || (m.P & 1))) return false;
  if (!jfast && (p.inner & 1)) return false;
  auto al = [](const void *q) { return ((size_t)q & 15) == 0; };
  if (!al(p.in0) || !al(p.out) || ((p.out_mode == OUT_ACC || p.out_mode == OUT_MUL || p.out_mode == OUT_ACC2) && (!p.acc || !al(p.acc)))) return false;
  SweepParams q = p0;                                    // sizes: the buffer offsets must reach (prepare_v)
  q.P = m.P; q.H = m.H; q.sink = m.sink;
  switch (m.KS) {
    case 4: return prepare_ok_t<4>(q, jfast);
    case 8: return prepare_ok_t<8>(q, jfast);
    case 16: return prepare_ok_t<16>(q, jfast);
    case 32: return prepare_ok_t<32>(q, jfast);
    default: return false;
  }
}

// ... and may carry p.raw != 0: the raw modes (sweep.h) are STORE-only, but for raw = 1 with OUT_MUL.  Option
// "general_kernels" (every sweep on the general kernel, which has no raw modes) switches them off as well.
bool sweep_vec_raw_eligible(const DiffMat &m, const SweepParams &p0) {
  if ((p0.out_mode != OUT_STORE && !(p0.out_mode == OUT_MUL && p0.raw == 1)) || opt(OPT_NO_RAW_TRANSFORMS) || opt(OPT_GENERAL_KERNELS)) return false;
  return sweep_vec_eligible(m, p0);
}

hipError_t sweep_vec_launch(const DiffMat &m, SweepParams p, hipStream_t stream) {
  if (p.in_mode == IN_SUM3) return hipErrorInvalidValue;   // exists in the multi-job launch only
  const bool jfast = p.inner < 16;
  switch (m.KS) {
    case 4: return jfast ? launch_v<4, true>(p, stream) : launch_v<4, false>(p, stream);
    case 8: return jfast ? launch_v<8, true>(p, stream) : launch_v<8, false>(p, stream);
    case 16: return jfast ? launch_v<16, true>(p, stream) : launch_v<16, false>(p, stream);
    case 32: return jfast ? launch_v<32, true>(p, stream) : launch_v<32, false>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

template <int KS>
static hipError_t launch_gather_t(SweepParams p, const GatherSrc &g, hipStream_t stream, bool *done) {
  const int gen = prepare_v<KS, false>(p);
  if (gen != 4) return hipSuccess;
  hipError_t cu_err; const int ncu = sweep_num_cus(&cu_err);
  if (cu_err != hipSuccess) return cu_err;
  const unsigned grid = p.ntiles < (unsigned)ncu ? p.ntiles : (unsigned)ncu;
  *done = true;
  if (grid == 0) return hipSuccess;
  if (g.push) hipLaunchKernelGGL((cheb_sweep_vec4_gather_kernel<KS, true>), dim3(grid), dim3(512), 0, stream, p, g);
  else hipLaunchKernelGGL((cheb_sweep_vec4_gather_kernel<KS, false>), dim3(grid), dim3(512), 0, stream, p, g);
  sweep_note_launch();
  return hipGetLastError();
}

// The gather launch: the dense geometry of the tensor (qmax = inner = the columns of a block, nouter = vectors of a batch); every
// array 16-byte aligned, every row start an even number of doubles into it.
static bool gather_ok(const DiffMat &m, SweepParams &p, const GatherSrc &g) {
  if (m.KS < 16 || p.inner < 16 || (p.inner & 1) || p.in_mode != IN_PLAIN || p.out_mode != OUT_STORE || p.raw || p.in_fblocks || p.qmax || p.in_os) return false;
  if (g.G < 1 || g.G > GATHER_MAX || (g.rowlen & 1) || (g.col0 & 1) || (size_t)g.col0 + p.inner > g.rowlen) return false;
  for (int s = 0; s < g.G; s++) if (!g.p[s] || ((size_t)g.p[s] & 15) || (g.lq[s] & 1) || g.pmax[s] < 1 || g.s0[s + 1] < g.s0[s]) return false;
  if (g.push) { for (int s = 0; s < g.G; s++) if (!g.dp[s] || ((size_t)g.dp[s] & 15)) return false; if (!p.out) return false; }
  if (g.s0[0] != 0 || g.s0[g.G] != m.P) return false;                    // the planes of the arrays are exactly the rows of a line
  p.in0 = g.p[0];                                                          // (alignment checks and descriptor sizes of the unused buffer path)
  return sweep_vec_eligible(m, p);
}
hipError_t sweep_vec_launch_gather(const DiffMat &m, SweepParams p, const GatherSrc &g, hipStream_t stream, bool *done) {
  *done = false;
  if (!gather_ok(m, p, g)) return hipSuccess;
  switch (m.KS) {
    case 16: return launch_gather_t<16>(p, g, stream, done);
    case 32: return launch_gather_t<32>(p, g, stream, done);
    default: return hipSuccess;
  }
}

// n <= MULTI_MAX independent sweeps as ONE launch when they qualify (plain in, STORE out, 16-byte kernel, same KS);
// *done = false: the caller launches them one by one
template <int KS>
static hipError_t launch_multi_t(int n, SweepParams *jobs, hipStream_t stream, bool *done, const GatherSrc *g = nullptr, unsigned gmask = 0) {
  if (g && KS < 16) { *done = false; return hipSuccess; }
  for (int j = 0; j < n; j++) if (jobs[j].raw || jobs[j].in_mode == IN_MUL) { *done = false; return hipSuccess; }
  bool sum3 = jobs[0].in_mode == IN_SUM3;                  // all jobs or none (stokes.hip: the three sweeps of grad div v)
  for (int j = 1; j < n; j++) if ((jobs[j].in_mode == IN_SUM3) != sum3) { *done = false; return hipSuccess; }
  if (sum3 && KS > 8) { *done = false; return hipSuccess; }
  MultiParams mp = {};
  mp.njobs = n;
  hipError_t cu_err; int ncu = sweep_num_cus(&cu_err);
  if (cu_err != hipSuccess) return cu_err;
  ncu *= wgs_per_cu<KS>();
  unsigned long long total = 0;
  for (int j = 0; j < n; j++) {
    SweepParams &p = jobs[j];
    const bool jfast = p.inner < 16;
    const int gen = jfast ? prepare_v<KS, true>(p) : prepare_v<KS, false>(p);
    if (gen == 0) return hipSuccess;                       // not for this launch: done stays false
    total += p.ntiles;
  }
  // One workgroup per CU in all, shared out in proportion to the jobs' tiles: every workgroup walks several tiles with
  // its prefetch pipeline running (one fill, one drain per launch) instead of the jobs taking the chip one after another.
  const int split = opt(OPT_EQUAL_SHARES) ? 0 : 1;
  unsigned gs[MULTI_MAX];
  for (int j = 0; j < n; j++) {
    const unsigned nt = jobs[j].ntiles;
    gs[j] = ((nt < (unsigned)ncu ? nt : (unsigned)ncu) + 7u) & ~7u;   // whole XCD rounds: the surplus workgroups exit at once
  }
  if (split && total > (unsigned long long)ncu && ncu % 8 == 0 && ncu >= 8 * n) {
    // shares in units of 8 workgroups (one per XCD) that add up to the CU count exactly: one workgroup too many would
    // wait for a whole job to finish
    const unsigned units = (unsigned)ncu / 8u;
    unsigned u[MULTI_MAX], used = 0;
    unsigned long long rem[MULTI_MAX];
    for (int j = 0; j < n; j++) {
      const unsigned long long x = (unsigned long long)units * jobs[j].ntiles;
      u[j] = (unsigned)(x / total); rem[j] = x % total;
      if (u[j] == 0 && jobs[j].ntiles) { u[j] = 1; rem[j] = 0; }
      used += u[j];
    }
    while (used < units) {                                  // largest remainders first
      int best = 0;
      for (int j = 1; j < n; j++) if (rem[j] > rem[best]) best = j;
      u[best]++; rem[best] = 0; used++;
    }
    if (used == units) {
      for (int j = 0; j < n; j++) if (8u * u[j] < gs[j]) gs[j] = 8u * u[j];
      // Shares in whole XCD rounds can leave one job a tile behind (three jobs of 256 / 254 / 254 tiles on 256 CUs: 88 / 88 / 80
      // workgroups, and a tenth of the last job's walkers take 4 tiles while everybody else takes 3).  When shares counted in single
      // workgroups shorten the longest walk they are used instead; such a job walks its tiles without the XCD grouping (NBLK % 8 != 0).
      auto walk = [&](unsigned nt, unsigned g_) -> unsigned {      // tiles of the busiest workgroup of a job
        if (g_ == 0) return nt ? 0xffffffffu : 0u;
        if (g_ % 8 == 0) { const unsigned per = (nt + 7) / 8, w8 = g_ / 8; return (per + w8 - 1) / w8; }
        return (nt + g_ - 1) / g_;
      };
      unsigned span_al = 0;
      for (int j = 0; j < n; j++) { const unsigned w_ = walk(jobs[j].ntiles, gs[j]); span_al = w_ > span_al ? w_ : span_al; }
      unsigned g1[MULTI_MAX], used1 = 0; unsigned long long rem1[MULTI_MAX];
      for (int j = 0; j < n; j++) {
        const unsigned long long x = (unsigned long long)ncu * jobs[j].ntiles;
        g1[j] = (unsigned)(x / total); rem1[j] = x % total;
        if (g1[j] == 0 && jobs[j].ntiles) { g1[j] = 1; rem1[j] = 0; }
        used1 += g1[j];
      }
      while (used1 < (unsigned)ncu) { int best = 0; for (int j = 1; j < n; j++) if (rem1[j] > rem1[best]) best = j; g1[best]++; rem1[best] = 0; used1++; }
      if (used1 == (unsigned)ncu) {
        unsigned span_1 = 0;
        for (int j = 0; j < n; j++) { const unsigned w_ = walk(jobs[j].ntiles, g1[j]); span_1 = w_ > span_1 ? w_ : span_1; }
        if (span_1 < span_al) for (int j = 0; j < n; j++) gs[j] = g1[j] < jobs[j].ntiles ? g1[j] : jobs[j].ntiles;
      }
    }
  }
  unsigned b = 0;
  for (int j = 0; j < n; j++) { mp.bstart[j] = b; b += gs[j]; mp.job[j] = jobs[j]; }
  mp.bstart[n] = b;
#ifdef CHEB_STAMPS
  // diagnostic builds (tools/stamp_probe_multi.py): one stamp area of 512 x 8 x 16 words per job behind the three areas of the
  // single launches; two sets of 9, taken in turn by successive multi-job launches (the two launches of a 64^3 StokesMatMult)
  if (::chebhip_stamp_buf()) {
    const int set = ::chebhip_stamp_next() & 1;
    for (int j = 0; j < n; j++) mp.job[j].in4 = ::chebhip_stamp_buf() + (size_t)3 * (256 * 8 * 16) + (size_t)(set * 9 + j) * (512 * 8 * 16);
  }
#endif
  if (b == 0) { *done = true; return hipSuccess; }
  if constexpr (KS <= 8) {
    if (sum3) hipLaunchKernelGGL((cheb_sweep_multi_kernel<KS, true>), dim3(b), dim3(512), 0, stream, mp);
    else hipLaunchKernelGGL((cheb_sweep_multi_kernel<KS>), dim3(b), dim3(512), 0, stream, mp);
  } else {
    if (g && g->push) hipLaunchKernelGGL((cheb_sweep_multi_gather_kernel<KS, true>), dim3(b), dim3(512), 0, stream, mp, *g, gmask);
    else if (g) hipLaunchKernelGGL((cheb_sweep_multi_gather_kernel<KS, false>), dim3(b), dim3(512), 0, stream, mp, *g, gmask);
    else hipLaunchKernelGGL((cheb_sweep_multi_kernel<KS>), dim3(b), dim3(512), 0, stream, mp);
  }
  sweep_note_launch();
  *done = true;
  return hipGetLastError();
}

hipError_t sweep_vec_launch_multi(int n, const DiffMat *const *m, SweepParams *jobs, hipStream_t stream, bool *done) {
  *done = false;
  if (n < 1 || n > MULTI_MAX) return hipSuccess;
  for (int j = 0; j < n; j++) {
    if (m[j]->KS != m[0]->KS || jobs[j].out_mode != OUT_STORE || !sweep_vec_eligible(*m[j], jobs[j])) return hipSuccess;
  }
  switch (m[0]->KS) {
    case 4: return launch_multi_t<4>(n, jobs, stream, done);
    case 8: return launch_multi_t<8>(n, jobs, stream, done);
    case 16: return launch_multi_t<16>(n, jobs, stream, done);
    case 32: return launch_multi_t<32>(n, jobs, stream, done);
    default: return hipSuccess;
  }
}

// ... the jobs of gmask reading their lines from the arrays of g (all jobs lines of more than 64 points, the same KS)
hipError_t sweep_vec_launch_multi_gather(int n, const DiffMat *const *m, SweepParams *jobs, unsigned gmask, const GatherSrc &g, hipStream_t stream, bool *done) {
  *done = false;
  if (n < 2 || n > MULTI_MAX || gmask == 0 || (gmask >> n) != 0) return hipSuccess;
  for (int j = 0; j < n; j++) {
    if (m[j]->KS != m[0]->KS || jobs[j].out_mode != OUT_STORE) return hipSuccess;
    if (((gmask >> j) & 1u) ? !gather_ok(*m[j], jobs[j], g) : !sweep_vec_eligible(*m[j], jobs[j])) return hipSuccess;
  }
  switch (m[0]->KS) {
    case 16: return launch_multi_t<16>(n, jobs, stream, done, &g, gmask);
    case 32: return launch_multi_t<32>(n, jobs, stream, done, &g, gmask);
    default: return hipSuccess;
  }
}

}  // namespace chebhip
