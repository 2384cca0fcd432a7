// fused.hip -- one launch = out (+)= alpha * D_k( coef( D_k u ) ) along one dimension.
//
// MatMult_Elliptic (elliptic.C:297-339) applies, per direction k, a gradient sweep, a pointwise
// flux and a divergence sweep.  Both sweeps run along the SAME lines, so a workgroup that holds a
// full set of lines can do all three without the gradient ever leaving the chip:
//
//   stage 1:  g = D u        MFMA chains over the (e,o)-split input tile in LDS region IN
//             f = coef(g)    in the accumulator registers (eta and c = deta du0 fetched per output before
//                            the chain, u read back from the LDS tile)
//             parity-split f straight from the accumulators into LDS region F
//   stage 2:  t = D f        MFMA chains over F;  out = acc + alpha*t  from the accumulators
//
// HBM traffic per point and direction drops from 16+16(+8) B to 8 (u) + 8/16 (out) (+ coefficients),
// which takes the memory system off the critical path: the launch is bound by the f64 MFMA rate
// (2 * P flop/point).  The matrix halves stay in registers as in sweep.hip; tile t+1 is loaded into
// registers during stage 1 of tile t and parked into IN during stage 2 (IN is dead by then).
#include "sweep.h"
#include <type_traits>
#include <cstdlib>

namespace chebhip {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
typedef unsigned u32;

template <int M> using mode_c = std::integral_constant<int, M>;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0): every
// wave would wait at each tile boundary for its own prefetch loads and result stores, which
// serialises the HBM stream with the MFMA phases.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Branch-free fetch with NO use of the loaded value: an invalid slot reads from a zero word
// (p.zero) instead of being masked afterwards.  Any arithmetic on the value here would make the
// compiler wait for the load right after issuing it and expose the whole HBM latency.
template <int MODE>
__device__ __forceinline__ double fetch_u(const SweepParams &p, u32 a, int j, int gb, bool ok) {
  const double *src;
  if (MODE == IN_PLAIN) src = p.in0 + a;
  else { ok = ok && gb >= 0 && j >= 1 && j <= p.P - 2; src = p.in0 + ((long)gb + (long)(j - 1) * p.gstride); }
  return *(ok ? src : p.zero);
}

template <int KS, bool JFAST, int COEF>
__global__ __launch_bounds__(512) void cheb_fused_kernel(const SweepParams p) {
  constexpr int MTP = KS / 4;
  constexpr int NG = 8 / MTP;
  constexpr int HP = 4 * KS;
  constexpr int NSUB = (KS >= 16) ? 2 : 1;
  constexpr int NT = 16 * NG * NSUB;
  constexpr int LDJ = HP + 1;   // odd pitch: conflict-free operand reads (sweep_vec.hip V_LDJ_PAD; this kernel's LDS accesses are all 8-byte)
  constexpr int LDS_ELEMS = JFAST ? NT * LDJ : HP * NT;
  constexpr int ITEMS = HP * NT / 512;
  constexpr int CH = ITEMS / NSUB;
  constexpr int QSTEP = JFAST ? 512 / HP : 512 / NT;
  constexpr int LDS_QSTEP = JFAST ? QSTEP * LDJ : QSTEP * NT;
  constexpr int KSTR = JFAST ? 4 : 4 * NT;
  __shared__ double smem[4 * LDS_ELEMS];           // IN (E,O) and F (E,O)
  double *inE = smem, *inO = smem + LDS_ELEMS, *fE_ = smem + 2 * LDS_ELEMS, *fO_ = smem + 3 * LDS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mt = w % MTP, ng = w / MTP;
  const int kq = lane >> 4, l16 = lane & 15;
  const int nn = p.P - 1, H = p.H;
  const u32 inner = p.inner, ncols = p.ncols;
  // trim: the arrays hold only the interior points 1..n-1 of every line (and only interior lines):
  // point j sits at base + (j-1)*inner, points 0 and n are implicit zeros on load and dropped on store.
  const bool trim = p.trim != 0;
  const u32 lineLen = (u32)(trim ? p.P - 2 : p.P) * inner;
  const u32 joff = trim ? inner : 0u;
  const int jlo = trim ? 1 : 0;
  const bool need_g = (p.in_mode == IN_GATHER) || (p.out_mode == OUT_ACC_SCATTER);

  double ae[KS], ao[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) {
    ae[s] = p.fragE[((long)(mt * KS + s)) * 64 + lane];
    ao[s] = p.fragO[((long)(mt * KS + s)) * 64 + lane];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see sweep.hip

  const u32 tpo = JFAST ? 1u : (inner + NT - 1) / NT;
  // XCD-aware tile walk: workgroups b and b+8 share an XCD (and its L2).  Give each XCD one
  // contiguous range of tiles and let its CUs take neighbouring tiles at the same time, so that a
  // 128-B line straddled by two neighbouring row pieces is fetched from HBM once, not once per XCD.
  const u32 nxcd = (gridDim.x % 8 == 0) ? 8u : 1u;
  const u32 t_per = (p.ntiles + nxcd - 1) / nxcd;
  const u32 t_lo = (blockIdx.x % nxcd) * t_per;
  const u32 t_hi = (t_lo + t_per < p.ntiles) ? t_lo + t_per : p.ntiles;
  const u32 t_step = gridDim.x / nxcd;
  const int ld_n = JFAST ? tid / HP : tid % NT;
  const int ld_j = JFAST ? tid % HP : tid / NT;
  const int ld_lds0 = JFAST ? ld_n * LDJ + ld_j : ld_j * NT + (ld_n ^ ((ld_j & 1) << 4));
  const int i0 = mt * 16 + (JFAST ? l16 : kq);

  double rj[CH], rm[CH];   // one chunk of the next tile in flight at a time

  auto issue_loads = [&](auto MODE, u32 tile, int chunk, double (&xj_)[CH], double (&xm_)[CH]) {
    constexpr int IM = decltype(MODE)::value;
    if (!JFAST) {
      const u32 o = tile / tpo, q0 = (tile - o * tpo) * NT;
      const u32 q = q0 + ld_n;
      const bool cv = q < inner;
      const u32 base = o * lineLen + q - joff;
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
        const bool ok = cv && jp < H && jp >= jlo;
        xj_[s] = fetch_u<IM>(p, base + rel, jp, gb, ok);
        xm_[s] = fetch_u<IM>(p, top - rel, jm, gb, ok && jm != jp);
      }
    } else {
      const int jp = ld_j, jm = nn - jp;
#pragma unroll
      for (int s = 0; s < CH; s++) {
        const u32 c = tile * NT + ld_n + (chunk * CH + s) * QSTEP;
        const bool ok = c < ncols && jp < H && jp >= jlo;
        const u32 cc = ok ? c : 0u;
        const u32 base = ((inner == 1) ? cc * lineLen : (cc / inner) * lineLen + (cc % inner)) - joff;
        const int gb = (IM == IN_GATHER) ? p.gcol[cc] : -1;
        xj_[s] = fetch_u<IM>(p, base + (u32)jp * inner, jp, gb, ok);
        xm_[s] = fetch_u<IM>(p, base + (u32)jm * inner, jm, gb, ok && jm != jp);
      }
    }
  };
  auto issue_loads_any = [&](u32 tile, int chunk, double (&xj_)[CH], double (&xm_)[CH]) {
    if (p.in_mode == IN_GATHER) issue_loads(mode_c<IN_GATHER>{}, tile, chunk, xj_, xm_);
    else issue_loads(mode_c<IN_PLAIN>{}, tile, chunk, xj_, xm_);
  };
  auto park_chunk = [&](int chunk, const double (&xj_)[CH], const double (&xm_)[CH]) {
    const bool mid = JFAST && (2 * ld_j == nn);
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const int idx = ld_lds0 + (chunk * CH + s) * LDS_QSTEP;
      const bool m2 = JFAST ? mid : (2 * (ld_j + (chunk * CH + s) * QSTEP) == nn);
      inE[idx] = xj_[s] + xm_[s];
      inO[idx] = m2 ? 0.0 : xj_[s] - xm_[s];
    }
  };

  // two accumulator chains over one 16-line sub-tile of an (E,O) LDS image
  auto chains = [&](const double *sE, const double *sO, int nb, v4d &ce, v4d &co) {
    const int frag = JFAST ? (nb + l16) * LDJ + kq : kq * NT + ((nb + l16) ^ ((kq & 1) << 4));
    const double *fE = sE + frag, *fO = sO + frag;
    ce = v4d{0.0, 0.0, 0.0, 0.0}; co = v4d{0.0, 0.0, 0.0, 0.0};
    double fb[2][4];
    fb[0][0] = fE[0]; fb[0][1] = fE[KSTR]; fb[0][2] = fO[0]; fb[0][3] = fO[KSTR];
#pragma unroll
    for (int g = 0; g < KS / 2; g++) {
      const int cb = g & 1, nbuf = cb ^ 1;
      if (g + 1 < KS / 2) {
        fb[nbuf][0] = fE[(2 * g + 2) * KSTR]; fb[nbuf][1] = fE[(2 * g + 3) * KSTR];
        fb[nbuf][2] = fO[(2 * g + 2) * KSTR]; fb[nbuf][3] = fO[(2 * g + 3) * KSTR];
      }
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
      // pin the order "LDS reads of group g+1, then the 4 MFMAs of group g": the reads then
      // complete under the MFMAs instead of being issued (and waited for) right before their use
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  };

  // addressing of this lane's accumulator rows in sub-tile `sub` of tile `tile`
  auto out_geom = [&](u32 tile, int nb, u32 (&ob)[4], bool (&ov)[4], int (&og)[4]) {
    const u32 t_o = tile / tpo, t_q0 = (tile - t_o * tpo) * NT;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      u32 b, cidx; bool lv;
      if (!JFAST) {
        const u32 q = t_q0 + nb + l16;
        lv = q < inner; b = t_o * lineLen + q - joff; cidx = t_o * inner + q;
        ov[r] = lv && (i0 + 4 * r < H);
      } else {
        const u32 c = tile * NT + nb + 4 * r + kq;
        lv = c < ncols; cidx = c;
        b = ((inner == 1) ? c * lineLen : (c / inner) * lineLen + (c % inner)) - joff;
        ov[r] = lv && (i0 < H);
      }
      ob[r] = b;
      og[r] = need_g ? p.gcol[lv ? cidx : 0u] : -1;   // clamped index, no select on the loaded value
    }
  };

  u32 tile = t_lo + blockIdx.x / nxcd;
  if (tile < t_hi) {
#pragma unroll 1
    for (int ch = 0; ch < NSUB; ch++) { issue_loads_any(tile, ch, rj, rm); park_chunk(ch, rj, rm); }
  }
  lds_barrier();
  for (; tile < t_hi; tile += t_step) {
    const u32 nxt = tile + t_step;
    const bool has_next = nxt < t_hi;

    // ======================= stage 1: g = D u, f = coef(g) -> F =======================
#pragma unroll 1
    for (int sub = 0; sub < NSUB; sub++) {
      // chunk 0 of the next tile is issued under the last chain of stage 1 and parked after the
      // first chain of stage 2; later chunks are issued when the previous one has been parked.
      if (has_next && sub == NSUB - 1) issue_loads_any(nxt, 0, rj, rm);
      const int nb = (ng * NSUB + sub) * 16;
      u32 ob[4]; bool ov[4]; int og[4];
      out_geom(tile, nb, ob, ov, og);
      // eta (and c = deta * du0) at (line, i) and (line, n-i): requested BEFORE the chain, so that the
      // round trip to HBM is covered by it
      double cv[8], cc[(COEF == COEF_FULL) ? 8 : 1];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        cv[2 * r] = 1.0; cv[2 * r + 1] = 1.0;
        if (COEF == COEF_FULL) { cc[2 * r] = 0.0; cc[2 * r + 1] = 0.0; }
        if (COEF == COEF_ETA && ov[r]) {
          const int i = i0 + (JFAST ? 0 : 4 * r);
          cv[2 * r] = p.in1[ob[r] + (u32)i * inner];
          cv[2 * r + 1] = p.in1[ob[r] + (u32)(nn - i) * inner];
        }
        if (COEF == COEF_FULL && ov[r]) {                  // in2 holds the pairs {eta, c}: one 16-B load per point
          const int i = i0 + (JFAST ? 0 : 4 * r);
          const d2v ei = *(const d2v *)(p.in2 + 2 * (size_t)(ob[r] + (u32)i * inner));
          const d2v em = *(const d2v *)(p.in2 + 2 * (size_t)(ob[r] + (u32)(nn - i) * inner));
          cv[2 * r] = ei[0]; cc[2 * r] = ei[1];
          cv[2 * r + 1] = em[0]; cc[2 * r + 1] = em[1];
        }
      }
      v4d ce, co;
      chains(inE, inO, nb, ce, co);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = i0 + (JFAST ? 0 : 4 * r), im = nn - i;
        double e2 = 0.0, o2 = 0.0;
        if (ov[r]) {
          const double gi = ce[r] + co[r], gm = co[r] - ce[r];
          const u32 ai = ob[r] + (u32)i * inner, am = ob[r] + (u32)im * inner;
          if (p.gout) { p.gout[ai] = gi; if (im != i) p.gout[am] = gm; }     // c->gradu[k], elliptic.C:498
          double fi = cv[2 * r] * gi, fm = cv[2 * r + 1] * gm;               // eta * g
          if (COEF == COEF_FULL) {                                     // + deta * u * du0 (elliptic.C:321)
            // u is on chip: the parity-split tile holds e = u_i + u_{n-i} and o = u_i - u_{n-i}
            const int uidx = JFAST ? (nb + 4 * r + kq) * LDJ + i : i * NT + ((nb + l16) ^ ((i & 1) << 4));
            const double ue = inE[uidx], uo = inO[uidx];
            const double ui = (im != i) ? ue + uo : 2.0 * ue, um = ue - uo;   // 2 u: in2 carries c / 2
            fi = fi + cc[2 * r] * ui;
            fm = fm + cc[2 * r + 1] * um;
          }
          if (im != i) { e2 = fi + fm; o2 = fi - fm; } else { e2 = fi; o2 = 0.0; }
        }
        // every (row < HP, line < NT) slot of F is written by exactly one lane (zeros in the padding)
        const int fidx = JFAST ? (nb + 4 * r + kq) * LDJ + i : i * NT + ((nb + l16) ^ ((i & 1) << 4));
        fE_[fidx] = e2; fO_[fidx] = o2;
      }
    }
    lds_barrier();   // F complete, IN dead

    // ======================= stage 2: t = D f, out = acc + alpha t =======================
#pragma unroll 1
    for (int sub = 0; sub < NSUB; sub++) {
      const int nb = (ng * NSUB + sub) * 16;
      u32 ob[4]; bool ov[4]; int og[4];
      out_geom(tile, nb, ob, ov, og);
      double accv[8];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        accv[2 * r] = 0.0; accv[2 * r + 1] = 0.0;
        const int i = i0 + (JFAST ? 0 : 4 * r);
        ov[r] = ov[r] && i >= jlo;                       // trimmed arrays have no slot for points 0 and n
        if (ov[r] && p.out_mode != OUT_STORE && p.acc) {
          accv[2 * r] = p.acc[ob[r] + (u32)i * inner];
          accv[2 * r + 1] = p.acc[ob[r] + (u32)(nn - i) * inner];
        }
      }
      v4d ce, co;
      chains(fE_, fO_, nb, ce, co);
      // Refill IN with the next tile BEFORE issuing this sub-tile's stores: the wait in front of
      // the parity split then covers loads only (vmcnt retires in order, so a wait placed after
      // the stores would also wait for them to reach memory).
      if (has_next) {
        park_chunk(sub, rj, rm);
        if (sub + 1 < NSUB) issue_loads_any(nxt, sub + 1, rj, rm);
      }
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
    lds_barrier();   // IN complete, F dead
  }
}

template <int KS, bool JFAST, int COEF>
static hipError_t launch_c(const SweepParams &p, unsigned grid, hipStream_t stream) {
  hipLaunchKernelGGL((cheb_fused_kernel<KS, JFAST, COEF>), dim3(grid), dim3(512), 0, stream, p);
  sweep_note_launch();
  return hipGetLastError();
}

template <int KS, bool JFAST>
static hipError_t launch_f(const SweepParams &p0, hipStream_t stream) {
  constexpr int MTP = KS / 4, NG = 8 / MTP, NSUB = (KS >= 16) ? 2 : 1, NT = 16 * NG * NSUB;
  SweepParams p = p0;
  if (JFAST) p.ntiles = (p.ncols + NT - 1) / NT;
  else p.ntiles = (p.ncols / p.inner) * ((p.inner + NT - 1) / NT);
  hipError_t cu_err; const int ncu = sweep_num_cus(&cu_err);
  if (cu_err != hipSuccess) return cu_err;
  const unsigned grid = p.ntiles < (unsigned)ncu ? p.ntiles : (unsigned)ncu;
  if (grid == 0) return hipSuccess;
  switch (p.coef_mode) {
    case COEF_UNIT: return launch_c<KS, JFAST, COEF_UNIT>(p, grid, stream);
    case COEF_ETA: return launch_c<KS, JFAST, COEF_ETA>(p, grid, stream);
    default: return launch_c<KS, JFAST, COEF_FULL>(p, grid, stream);
  }
}

hipError_t fused_launch(const DiffMat &m, SweepParams p, hipStream_t stream) {
  if (p.in_mode != IN_PLAIN && p.in_mode != IN_GATHER) return hipErrorInvalidValue;
  p.P = m.P; p.H = m.H; p.fragE = m.fragE; p.fragO = m.fragO; p.zero = m.zero;
  const bool jfast = p.inner < 16;
  switch (m.KS) {
    case 4: return jfast ? launch_f<4, true>(p, stream) : launch_f<4, false>(p, stream);
    case 8: return jfast ? launch_f<8, true>(p, stream) : launch_f<8, false>(p, stream);
    case 16: return jfast ? launch_f<16, true>(p, stream) : launch_f<16, false>(p, stream);
    case 32: return jfast ? launch_f<32, true>(p, stream) : launch_f<32, false>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace chebhip
