// tdk_rcd_quad.h -- RCD column strips, register-blocked: a lane owns FOUR adjacent columns of a row (round 4).
// Included by rcd.hip inside its anonymous namespace, after tdk_rcd_stream.h (shares its plane geometry, lags, border-rule
// helpers and slide; div_pos / div_signed, Range, stale_diff, ring_piece come from rcd.hip).
//
// Same nine steps, the same expressions and the same schedule as rs::rcd_stream (reference csrc/debayer/rcd.cu:63-282): a
// workgroup owns a strip of 108 output columns (128 computed), walks DOWN the frame 8 rows per step, every step runs its
// vertical tap reach behind the newest CFA row on sliding linear LDS planes.  What changes is who reads what:
//  * rs: wave = row, lane = column PAIR, planes de-interleaved by column parity; every tap is its own 4-byte LDS read (148 per
//    lane and step, 74 per pixel), 24 waves per CU.
//  * here: half a wave = one row, lane = columns 4 q .. 4 q + 3, planes in natural column order.  A lane reads ONLY ITS OWN
//    columns: one ds_read_b128 per plane row it touches (one ds_read_b64 of the compacted half-density planes), whatever the
//    number of taps in that row; the taps of the neighbour columns (reach <= 4 = one quad) come from the neighbour lanes'
//    registers through DPP wavefront shifts (v_*_dpp wave_shr:1 / wave_shl:1, folded into the consuming instruction by
//    hipcc: +2 cycles, no extra instruction).  47 16-byte + 27 8-byte reads per lane and step = 19 reads per pixel.
//  * the two halves of a wave take rows r and r + 4 of the 8-row block (same CFA phase, so the R/B column parity stays a
//    compile-time constant of the wave's code variant); a shift that crosses lane 31 | 32 brings a value of the other row --
//    into the outermost halo columns only, which hold garbage by construction (they read beyond the window in rs as well).
//  * 256-thread workgroups (4 waves x 2 rows), the same 47 KB of planes: three workgroups per CU = 12 waves, 104-112 VGPRs.
// Measured (profiles/r04/experiments/rcd_quad.txt, coresidency.txt): 45 % of rs's LDS and 94 % of its VALU instructions, the same
// bits on all RCD tests; alone on the GPU 183 us against 164 (three waves per SIMD do not cover the dependent chains of the exact
// divisions), but with other frames in flight on other streams the chain gains 3 %: the wave slots and registers this kernel
// leaves (5 slots, 176 VGPRs per SIMD) take the streaming kernels of the other frames.  Selected by TDK_RCD_CONCURRENT.
// CPL = 2 (a lane owns two columns, a wave a row, 24 waves per CU: experiments only) needs two DPP hops for the far taps and lost
// on both counts (profiles/r04/experiments/rcd_pair_dpp.txt).
#pragma once

namespace rq {

#ifdef TDK_RCD_TIMING
// experiments: clock deltas of wave 0 of one mid-frame workgroup per phase of a step, summed over its steps (profiles/rcd_phase_exp.py):
// 0 slide + new rows (two barriers), 1 steps 2.1 / 1.1 / 4.1, 2 steps 1.2 / 4.2, 3 step 3.1, 4 step 5.1, 5 step 5.2 + stores;
// 8 + k: the same phase up to the ARRIVAL at its closing barrier (the rest is waiting for the other waves)
#define RQ_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x == 300u) { const unsigned long long t_ = clock64(); atomicAdd(&g_rcd_phase_cycles[k], t_ - rq_t0); if ((k) < 8) rq_t0 = t_; } } while (0)
#else
#define RQ_MARK(k)
#endif

using namespace rs;  // plane geometry (CFA_B ... COL_B, *_L, *_W), LAG_*, RB, TWS, HALO, PAD, fdiv2, lmask, keep, SLOW / FASTM / INNER, slot16, Pair

constexpr int HALO = rs::HALO;  // (declared here: rcd.hip's tile kernel has a constant of the same name)

// CPL columns per lane: 4 (half a wave = one row, 256 threads, 12 waves per CU) or 2 (a wave = one row as in rs, 512 threads, 24
// waves per CU; full planes read 8 bytes at a time, the half-density ones 4, taps up to two lanes away = two DPP shifts)
template <int CPL> struct Geo {
  static_assert(CPL == 2 || CPL == 4, "columns per lane");
  static constexpr int LPR = 128 / CPL;  // lanes per row
  static constexpr int NT = RB * LPR;    // threads
  static constexpr int HPL = CPL / 2;    // entries of a half-density plane row per lane = sites of one kind per lane
  static constexpr int WPE = WG_PER_CU * NT / 256;  // waves per SIMD with three workgroups per CU
  static constexpr int SLIDE_PT = (SLIDE_SLOTS + NT - 1) / NT;  // float4 slide slots per thread (the last round for 128 threads)
  static_assert(SLIDE_SLOTS - (SLIDE_PT - 1) * NT == 128, "the last slide round is two whole waves");
};

// row offsets (floats, relative to the lane bases bF / bH) of plane row dr as seen by a step at `lag`
template <int BASE, int LIVE, int LAGW> constexpr int rowF(int lag, int dr) { return BASE + (LIVE + LAGW - lag + dr) * 128; }
template <int BASE, int LIVE, int LAGW> constexpr int rowH(int lag, int dr) { return BASE + (LIVE + LAGW - lag + dr) * 64; }

template <int N> struct Row { float v[N]; };
// 16-byte / 8-byte LDS accesses as native vector types (a struct of two floats is split into scalar loads before the back end
// sees its alignment, and comes back as a bank-conflicting ds_read2_b32)
typedef float vec4f __attribute__((ext_vector_type(4)));
typedef float vec2f __attribute__((ext_vector_type(2)));
template <int N> __device__ __forceinline__ Row<N> ldv(const float* p) {
  if constexpr (N == 4) {
    const vec4f t = *reinterpret_cast<const vec4f*>(__builtin_assume_aligned(p, 16));
    return Row<4>{{t.x, t.y, t.z, t.w}};
  } else if constexpr (N == 2) {
    const vec2f t = *reinterpret_cast<const vec2f*>(__builtin_assume_aligned(p, 8));
    return Row<2>{{t.x, t.y}};
  } else {
    return Row<1>{{*p}};
  }
}
template <int N> __device__ __forceinline__ void stv(float* p, const float (&v)[N]) {
  if constexpr (N == 4) *reinterpret_cast<vec4f*>(__builtin_assume_aligned(p, 16)) = vec4f{v[0], v[1], v[2], v[3]};
  else if constexpr (N == 2) *reinterpret_cast<vec2f*>(__builtin_assume_aligned(p, 8)) = vec2f{v[0], v[1]};
  else *p = v[0];
}

// the value the lane below / above holds in the same register (0 at the wave's ends)
__device__ __forceinline__ float from_lo(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xF, 0xF, true)); }  // wave_shr:1
__device__ __forceinline__ float from_hi(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xF, 0xF, true)); }  // wave_shl:1

// element idx of a plane row relative to the lane's first element, at most two lanes away (idx is a constant after unrolling)
template <int N> __device__ __forceinline__ float tap(const Row<N>& r, int idx) {
  const int s = idx >= 0 ? idx / N : -((N - 1 - idx) / N);  // lane distance: floor(idx / N)
  float x = r.v[idx - s * N];
  if (s <= -1) x = from_lo(x);
  if (s <= -2) x = from_lo(x);
  if (s >= 1) x = from_hi(x);
  if (s >= 2) x = from_hi(x);
  return x;
}

// per-thread view of a step: lane bases, frame position of the lane's first column, the frame rows of its site in each step
struct Lane {
  float* bF;   // lds + row_in_block * 128 + 4 q
  float* bH;   // lds + row_in_block * 64 + 2 q
  int gxq;     // frame x of column 0 of the quad
  int w, h;
};
// border rule of site column ci for this lane, as a lane mask (non-INNER variants only: border blocks)
__device__ __forceinline__ lmask colrow(bool row_ok, int gx, int lo, int hi) { return __builtin_amdgcn_ballot_w64(row_ok && gx >= lo && gx <= hi); }

// ---- steps 2.1 (lag 1), 1.1 and 4.1 (lag 3)
template <int CPL, int MODE, int PE, typename TI>
__device__ __forceinline__ void q_step_2_1_1_1_4_1(const Lane& t, int gyb, const TI* __restrict__ in) {
  // ---- step 2.1: lpf at the two R/B sites of the quad
  {
    constexpr int L = LAG_21, p = PE ^ (L & 1);
    Row<CPL> c[3];
#pragma unroll
    for (int dr = -1; dr <= 1; dr++) c[dr + 1] = ldv<CPL>(t.bF + rowF<CFA_B, CFA_L, CFA_W>(L, dr));
    float o[CPL / 2];
#pragma unroll
    for (int k = 0; k < CPL / 2; k++) {
      const int ci = p + 2 * k;
      auto a = [&](int dr, int dc) { return tap(c[dr + 1], ci + dc); };
      const float v = lpf9<approx(MODE)>(a(0, 0), a(-1, 0), a(1, 0), a(0, -1), a(0, 1), a(-1, -1), a(-1, 1), a(1, -1), a(1, 1));
      if constexpr (!has_rules(MODE)) o[k] = v;
      else { const int gy = gyb - L; o[k] = keep(colrow(gy >= 2 && gy <= t.h - 2, t.gxq + ci, 2, t.w - 2), v); }
    }
    stv(t.bH + rowH<LPF_B, LPF_L, LPF_W>(L, 0), o);
  }
  // ---- steps 1.1 (v_diff / h_diff at all four columns) and 4.1 (p/q_diff at the odd columns): the same seven CFA rows
  {
    constexpr int L = LAG_11;
    static_assert(LAG_41 == LAG_11, "steps 1.1 and 4.1 share their CFA rows");
    Row<CPL> c[7];
#pragma unroll
    for (int dr = -3; dr <= 3; dr++) c[dr + 3] = ldv<CPL>(t.bF + rowF<CFA_B, CFA_L, CFA_W>(L, dr));
    const int gy = gyb - L;
    float vd[CPL], hd[CPL];
#pragma unroll
    for (int ci = 0; ci < CPL; ci++) {
      auto a = [&](int dr, int dc) { return tap(c[dr + 3], ci + dc); };
      vd[ci] = sqf(hp7<approx(MODE)>(a(-3, 0), a(-2, 0), a(-1, 0), a(0, 0), a(1, 0), a(2, 0), a(3, 0)));
      hd[ci] = sqf(hp7<approx(MODE)>(a(0, -3), a(0, -2), a(0, -1), a(0, 0), a(0, 1), a(0, 2), a(0, 3)));
      if constexpr (has_rules(MODE)) {
        const lmask ok = colrow(gy >= 3 && gy <= t.h - 4, t.gxq + ci, 3, t.w - 4);
        vd[ci] = keep(ok, vd[ci]);
        hd[ci] = keep(ok, hd[ci]);
      }
    }
    stv(t.bF + rowF<VD_B, VD_L, VD_W>(L, 0), vd);
    stv(t.bF + rowF<HD_B, HD_L, HD_W>(L, 0), hd);
    float pv[CPL / 2], qv[CPL / 2];
#pragma unroll
    for (int k = 0; k < CPL / 2; k++) {
      const int ci = 1 + 2 * k;
      auto a = [&](int dr, int dc) { return tap(c[dr + 3], ci + dc); };
      pv[k] = sqf(dg7<approx(MODE)>(a(-3, -3), a(-2, -2), a(-1, -1), a(0, 0), a(1, 1), a(2, 2), a(3, 3)));
      qv[k] = sqf(dg7<approx(MODE)>(a(-3, 3), a(-2, 2), a(-1, 1), a(0, 0), a(1, -1), a(2, -2), a(3, -3)));
      if constexpr (has_rules(MODE)) {
        // slots step 4.1 does not write keep the same call's v_diff / h_diff of the shared buffer (rcd.cu:637-652; stale_diff in
        // rcd.hip); outside the frame: 0
        const int gx = t.gxq + ci;
        const bool inside = gy >= 0 && gy < t.h && gx >= 0 && gx < t.w, ranged = gy >= 3 && gy <= t.h - 4 && gx >= 3 && gx <= t.w - 4;
        if (__builtin_amdgcn_ballot_w64(inside && !ranged) != 0) {
          if (inside && !ranged) stale_pair(in, gy, gx, t.w, t.h, pv[k], qv[k]);
        }
        const lmask m = __builtin_amdgcn_ballot_w64(inside);
        pv[k] = keep(m, pv[k]);
        qv[k] = keep(m, qv[k]);
      }
    }
    stv(t.bH + rowH<P_B, P_L, P_W>(L, 0), pv);
    stv(t.bH + rowH<Q_B, Q_L, Q_W>(L, 0), qv);
  }
}

// ---- steps 1.2 (VH_dir at all four columns) and 4.2 (PQ_dir at the R/B sites), lag 4
template <int CPL, int MODE, int PE>
__device__ __forceinline__ void q_step_1_2_4_2(const Lane& t, int gyb) {
  constexpr int L = LAG_12;
  static_assert(LAG_42 == LAG_12, "steps 1.2 and 4.2 run at the same lag");
  const int gy = gyb - L;
  {
    Row<CPL> vdr[3];
#pragma unroll
    for (int dr = -1; dr <= 1; dr++) vdr[dr + 1] = ldv<CPL>(t.bF + rowF<VD_B, VD_L, VD_W>(L, dr));
    const Row<CPL> hdr = ldv<CPL>(t.bF + rowF<HD_B, HD_L, HD_W>(L, 0));
    float vh[CPL];
#pragma unroll
    for (int ci = 0; ci < CPL; ci++) {
      const float eps = 1e-10f;
      const float V_Stat = fmaxf(eps, vdr[0].v[ci] + vdr[1].v[ci] + vdr[2].v[ci]);
      const float H_Stat = fmaxf(eps, tap(hdr, ci - 1) + tap(hdr, ci) + tap(hdr, ci + 1));
      vh[ci] = qdiv<MODE>(V_Stat, V_Stat + H_Stat);
      if constexpr (has_rules(MODE)) vh[ci] = keep(colrow(gy >= 2 && gy <= t.h - 3, t.gxq + ci, 2, t.w - 3), vh[ci]);
    }
    stv(t.bF + rowF<VH_B, VH_L, VH_W>(L, 0), vh);
  }
  {
    // p/q slot of odd column 2 j + 1 = entry j; the slots of (col - 1) | 1 on the neighbour rows: j - 1 + p (rcd.cu:166-182)
    constexpr int p = PE ^ (L & 1), jm = p - 1;
    Row<CPL / 2> pr[3], qr[3];
#pragma unroll
    for (int dr = -1; dr <= 1; dr++) {
      pr[dr + 1] = ldv<CPL / 2>(t.bH + rowH<P_B, P_L, P_W>(L, dr));
      qr[dr + 1] = ldv<CPL / 2>(t.bH + rowH<Q_B, Q_L, Q_W>(L, dr));
    }
    float pq[CPL / 2];
#pragma unroll
    for (int k = 0; k < CPL / 2; k++) {
      const float eps = 1e-10f;
      const float P_Stat = fmaxf(eps, tap(pr[0], k + jm) + pr[1].v[k] + tap(pr[2], k + jm + 1));
      const float Q_Stat = fmaxf(eps, tap(qr[0], k + jm + 1) + qr[1].v[k] + tap(qr[2], k + jm));
      // plain division wherever a stale slot (see step 4.1) can be near: it holds values of samples no range check has seen
      pq[k] = qdiv42<MODE>(P_Stat, P_Stat + Q_Stat);
      if constexpr (has_rules(MODE)) pq[k] = keep(colrow(gy >= 2 && gy <= t.h - 3, t.gxq + p + 2 * k, 2, t.w - 3), pq[k]);
    }
    stv(t.bH + rowH<PQ_B, PQ_L, PQ_W>(L, 0), pq);
  }
}

// ---- step 3.1 (lag 5): green at the two R/B sites
template <int CPL, int MODE, int PE>
__device__ __forceinline__ void q_step_3_1(const Lane& t, int gyb) {
  constexpr int L = LAG_31, p = PE ^ (L & 1);
  Row<CPL> c[9], vhr[3];
  Row<CPL / 2> lpr[5];
#pragma unroll
  for (int dr = -4; dr <= 4; dr++) c[dr + 4] = ldv<CPL>(t.bF + rowF<CFA_B, CFA_L, CFA_W>(L, dr));
#pragma unroll
  for (int dr = -1; dr <= 1; dr++) vhr[dr + 1] = ldv<CPL>(t.bF + rowF<VH_B, VH_L, VH_W>(L, dr));
#pragma unroll
  for (int dr = -2; dr <= 2; dr++) lpr[dr + 2] = ldv<CPL / 2>(t.bH + rowH<LPF_B, LPF_L, LPF_W>(L, dr));
  float g[CPL / 2];
#pragma unroll
  for (int k = 0; k < CPL / 2; k++) {
    const int ci = p + 2 * k;
    auto a = [&](int dr, int dc) { return tap(c[dr + 4], ci + dc); };
    auto vh = [&](int dr, int dc) { return tap(vhr[dr + 1], ci + dc); };
    auto lp = [&](int dr, int sh) { return tap(lpr[dr + 2], k + sh); };
    const float eps = 1e-5f;
    const float VH_c = vh(0, 0);
    const float VH_n = 0.25f * (vh(-1, -1) + vh(-1, 1) + vh(1, -1) + vh(1, 1));
    const float VH_Disc = (fabsf(0.5f - VH_c) < fabsf(0.5f - VH_n)) ? VH_n : VH_c;
    const float cfai = a(0, 0);
    const float N_Grad = eps + fabsf(a(-1, 0) - a(1, 0)) + fabsf(cfai - a(-2, 0)) + fabsf(a(-1, 0) - a(-3, 0)) + fabsf(a(-2, 0) - a(-4, 0));
    const float S_Grad = eps + fabsf(a(1, 0) - a(-1, 0)) + fabsf(cfai - a(2, 0)) + fabsf(a(1, 0) - a(3, 0)) + fabsf(a(2, 0) - a(4, 0));
    const float W_Grad = eps + fabsf(a(0, -1) - a(0, 1)) + fabsf(cfai - a(0, -2)) + fabsf(a(0, -1) - a(0, -3)) + fabsf(a(0, -2) - a(0, -4));
    const float E_Grad = eps + fabsf(a(0, 1) - a(0, -1)) + fabsf(cfai - a(0, 2)) + fabsf(a(0, 1) - a(0, 3)) + fabsf(a(0, 2) - a(0, 4));
    const float lpfi = lp(0, 0);
    const float N_Est = qdiv<MODE>(a(-1, 0) * (lpfi + lpfi), eps + lpfi + lp(-2, 0));
    const float S_Est = qdiv<MODE>(a(1, 0) * (lpfi + lpfi), eps + lpfi + lp(2, 0));
    const float W_Est = qdiv<MODE>(a(0, -1) * (lpfi + lpfi), eps + lpfi + lp(0, -1));
    const float E_Est = qdiv<MODE>(a(0, 1) * (lpfi + lpfi), eps + lpfi + lp(0, 1));
    const float V_Est = qdiv<MODE>(dot2<approx(MODE)>(S_Grad, N_Est, N_Grad, S_Est), N_Grad + S_Grad);
    const float H_Est = qdiv<MODE>(dot2<approx(MODE)>(W_Grad, E_Est, E_Grad, W_Est), E_Grad + W_Grad);
    g[k] = mixq<approx(MODE)>(V_Est, H_Est, VH_Disc);
    if constexpr (has_rules(MODE)) { const int gy = gyb - L; g[k] = keep(colrow(gy >= 4 && gy <= t.h - 5, t.gxq + ci, 4, t.w - 5), g[k]); }
  }
  stv(t.bH + rowH<GRN_B, GRN_L, GRN_W>(L, 0), g);
}

// ---- step 5.1 (lag 7): the opposite colour at the two R/B sites.  `rel`: the lanes whose numerators count in the wave's
// fast-division test (the outermost quads compute on garbage, which nothing that is stored ever reads).
template <int CPL, int MODE, int PE>
__device__ __forceinline__ void q_step_5_1(const Lane& t, int gyb, lmask rel) {
  constexpr int L = LAG_51, p = PE ^ (L & 1);
  Row<CPL> c[7];  // rows -3, -1, 1, 3 are read
  Row<CPL / 2> pqr[3], gr[5];
#pragma unroll
  for (int dr = -3; dr <= 3; dr += 2) c[dr + 3] = ldv<CPL>(t.bF + rowF<CFA_B, CFA_L, CFA_W>(L, dr));
#pragma unroll
  for (int dr = -1; dr <= 1; dr++) pqr[dr + 1] = ldv<CPL / 2>(t.bH + rowH<PQ_B, PQ_L, PQ_W>(L, dr));
#pragma unroll
  for (int dr = -2; dr <= 2; dr++) gr[dr + 2] = ldv<CPL / 2>(t.bH + rowH<GRN_B, GRN_L, GRN_W>(L, dr));
  float num[CPL], den[CPL], g0v[CPL / 2], disc[CPL / 2];
  lmask ok[CPL / 2];
  float mn = 1.0f;  // smallest numerator magnitude over the sites that count
#pragma unroll
  for (int k = 0; k < CPL / 2; k++) {
    const int ci = p + 2 * k;
    auto a = [&](int dr, int dc) { return tap(c[dr + 3], ci + dc); };
    auto pq = [&](int dr, int sh) { return tap(pqr[dr + 1], k + sh); };
    // green of R/B site (row + dr, col + dc): rows of the same parity keep the site's entry shifted by dc / 2, the other rows
    // hold their R/B sites on the other column parity: entry k + (p + dc - (1 - p)) / 2
    auto G = [&](int dr, int dc) { return tap(gr[dr + 2], k + ((dr & 1) ? fdiv2(2 * p + dc - 1) : dc / 2)); };
    constexpr int s = p - 1;  // entry of slot (col - 1) / 2 on the neighbour rows (rcd.cu:199-207)
    const float eps = 1e-5f;
    const float PQ_c = pq(0, 0);
    const float PQ_n = 0.25f * (pq(-1, s) + pq(-1, s + 1) + pq(1, s) + pq(1, s + 1));
    disc[k] = (fabsf(0.5f - PQ_c) < fabsf(0.5f - PQ_n)) ? PQ_n : PQ_c;
    const float g0 = G(0, 0);
    g0v[k] = g0;
    const float NW_Grad = eps + fabsf(a(-1, -1) - a(1, 1)) + fabsf(a(-1, -1) - a(-3, -3)) + fabsf(g0 - G(-2, -2));
    const float NE_Grad = eps + fabsf(a(-1, 1) - a(1, -1)) + fabsf(a(-1, 1) - a(-3, 3)) + fabsf(g0 - G(-2, 2));
    const float SW_Grad = eps + fabsf(a(-1, 1) - a(1, -1)) + fabsf(a(1, -1) - a(3, -3)) + fabsf(g0 - G(2, -2));
    const float SE_Grad = eps + fabsf(a(-1, -1) - a(1, 1)) + fabsf(a(1, 1) - a(3, 3)) + fabsf(g0 - G(2, 2));
    const float NW_Est = a(-1, -1) - G(-1, -1);
    const float NE_Est = a(-1, 1) - G(-1, 1);
    const float SW_Est = a(1, -1) - G(1, -1);
    const float SE_Est = a(1, 1) - G(1, 1);
    num[2 * k] = dot2<approx(MODE)>(NW_Grad, SE_Est, SE_Grad, NW_Est);
    num[2 * k + 1] = dot2<approx(MODE)>(NE_Grad, SW_Est, SW_Grad, NE_Est);
    den[2 * k] = NW_Grad + SE_Grad;
    den[2 * k + 1] = NE_Grad + SW_Grad;
    if constexpr (!has_rules(MODE)) ok[k] = rel;
    else { const int gy = gyb - L; ok[k] = colrow(gy >= 4 && gy <= t.h - 4, t.gxq + ci, 4, t.w - 4); }
    if constexpr (MODE == FASTM || MODE == INNER) {
      float r;  // min(|num|) of the site where it counts, 1 elsewhere
      const float m2 = fminf(fabsf(num[2 * k]), fabsf(num[2 * k + 1]));
      const lmask cnt = MODE == INNER ? rel : (ok[k] & rel);
      asm("v_cndmask_b32_e64 %0, 1.0, %1, %2" : "=v"(r) : "v"(m2), "s"(cnt));
      mn = fminf(mn, r);
    }
  }
  float est[CPL];  // P_Est, Q_Est of the sites
  if constexpr (approx(MODE)) {
#pragma unroll
    for (int i = 0; i < CPL; i++) est[i] = div_rcp(num[i], den[i]);
  } else {
    div_signed_min<MODE != SLOW>(num, den, est, mn);
  }
  float o[CPL / 2];
#pragma unroll
  for (int k = 0; k < CPL / 2; k++) {
    o[k] = g0v[k] + mixq<approx(MODE)>(est[2 * k], est[2 * k + 1], disc[k]);
    if constexpr (has_rules(MODE)) o[k] = keep(ok[k], o[k]);
  }
  stv(t.bH + rowH<COL_B, COL_L, COL_W>(L, 0), o);
}

// ---- step 5.2 (lag 10) at the two green sites of the quad + its four finished pixels.  sto[ci]: column ci of this lane is
// stored; stm: the lanes that store anything (their numerators count in the fast-division test).
template <int CPL, int MODE, int PE, typename T>
__device__ __forceinline__ void q_step_5_2_out(const Lane& t, bool red_row, T* __restrict__ dst, const bool (&sto)[CPL], lmask stm) {
  constexpr int L = LAG_52, p = PE ^ (L & 1), pg = 1 - p;  // R/B sites on parity p, the green sites on pg
  Row<CPL> c[7], vhr[3];
  Row<CPL / 2> gr[3], cr[7];  // colour rows -3, -1, 0, 1, 3 are read
#pragma unroll
  for (int dr = -3; dr <= 3; dr++) c[dr + 3] = ldv<CPL>(t.bF + rowF<CFA_B, CFA_L, CFA_W>(L, dr));
#pragma unroll
  for (int dr = -1; dr <= 1; dr++) {
    vhr[dr + 1] = ldv<CPL>(t.bF + rowF<VH_B, VH_L, VH_W>(L, dr));
    gr[dr + 1] = ldv<CPL / 2>(t.bH + rowH<GRN_B, GRN_L, GRN_W>(L, dr));
  }
#pragma unroll
  for (int dr = -3; dr <= 3; dr++)
    if (dr != -2 && dr != 2) cr[dr + 3] = ldv<CPL / 2>(t.bH + rowH<COL_B, COL_L, COL_W>(L, dr));
  float num[2 * CPL], den[2 * CPL], gsite[CPL / 2], disc[CPL / 2];
  float mn = 1.0f;
#pragma unroll
  for (int k = 0; k < CPL / 2; k++) {
    const int ci = pg + 2 * k;
    auto a = [&](int dr, int dc) { return tap(c[dr + 3], ci + dc); };
    auto vh = [&](int dr, int dc) { return tap(vhr[dr + 1], ci + dc); };
    auto grn = [&](int dr, int sh) { return tap(gr[dr + 1], k + sh); };
    auto col = [&](int dr, int sh) { return tap(cr[dr + 3], k + sh); };
    const float eps = 1e-5f;
    const float VH_c = vh(0, 0);
    const float VH_n = 0.25f * (vh(-1, -1) + vh(-1, 1) + vh(1, -1) + vh(1, 1));
    disc[k] = (fabsf(0.5f - VH_c) < fabsf(0.5f - VH_n)) ? VH_n : VH_c;
    const float g = a(0, 0);
    gsite[k] = g;
    const float N1 = eps + fabsf(g - a(-2, 0));
    const float S1 = eps + fabsf(g - a(2, 0));
    const float W1 = eps + fabsf(g - a(0, -2));
    const float E1 = eps + fabsf(g - a(0, 2));
    // green at the four R/B neighbours: above / below the site's own entry (those rows hold their R/B sites on this column's
    // parity), left / right the entries of columns col -+ 1 of this row
    const float gN = grn(-1, 0), gS = grn(1, 0), gW = grn(0, -p), gE = grn(0, 1 - p);
    // colour of this row's R/B sites (`own`): native left / right, from step 5.1 above / below; the other colour the other way round
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
      float cN, cS, cW, cE, cN3, cS3, cW3, cE3;
      if (cc == 0) {
        cW = a(0, -1); cE = a(0, 1); cW3 = a(0, -3); cE3 = a(0, 3);
        cN = col(-1, 0); cS = col(1, 0); cN3 = col(-3, 0); cS3 = col(3, 0);
      } else {
        cW = col(0, -p); cE = col(0, 1 - p); cW3 = col(0, -p - 1); cE3 = col(0, 2 - p);
        cN = a(-1, 0); cS = a(1, 0); cN3 = a(-3, 0); cS3 = a(3, 0);
      }
      const float SNabs = fabsf(cN - cS);
      const float EWabs = fabsf(cW - cE);
      const float N_Grad = N1 + SNabs + fabsf(cN - cN3);
      const float S_Grad = S1 + SNabs + fabsf(cS - cS3);
      const float W_Grad = W1 + EWabs + fabsf(cW - cW3);
      const float E_Grad = E1 + EWabs + fabsf(cE - cE3);
      const float N_Est = cN - gN;
      const float S_Est = cS - gS;
      const float W_Est = cW - gW;
      const float E_Est = cE - gE;
      num[4 * k + 2 * cc] = dot2<approx(MODE)>(N_Grad, S_Est, S_Grad, N_Est);
      den[4 * k + 2 * cc] = N_Grad + S_Grad;
      num[4 * k + 2 * cc + 1] = dot2<approx(MODE)>(E_Grad, W_Est, W_Grad, E_Est);
      den[4 * k + 2 * cc + 1] = E_Grad + W_Grad;
    }
    if constexpr (MODE == FASTM || MODE == INNER) {
      float r;
      const float m4 = fminf(fminf(fabsf(num[4 * k]), fabsf(num[4 * k + 1])), fminf(fabsf(num[4 * k + 2]), fabsf(num[4 * k + 3])));
      asm("v_cndmask_b32_e64 %0, 1.0, %1, %2" : "=v"(r) : "v"(m4), "s"(stm));
      mn = fminf(mn, r);
    }
  }
  float est[2 * CPL];
  if constexpr (approx(MODE)) {
#pragma unroll
    for (int i = 0; i < 2 * CPL; i++) est[i] = div_rcp(num[i], den[i]);
  } else {
    div_signed_min<MODE != SLOW>(num, den, est, mn);
  }
  // pixels of the two column pairs (2 k, 2 k + 1): the R/B pixel (native, green from step 3.1, other colour from step 5.1) and
  // the green pixel
#pragma unroll
  for (int k = 0; k < CPL / 2; k++) {
    const float g = gsite[k];
    const float own = fmaxf(g + mixq<approx(MODE)>(est[4 * k], est[4 * k + 1], disc[k]), 0.0f), oth = fmaxf(g + mixq<approx(MODE)>(est[4 * k + 2], est[4 * k + 3], disc[k]), 0.0f);
    const float native = fmaxf(c[3].v[p + 2 * k], 0.0f), green = fmaxf(gr[1].v[k], 0.0f), other = fmaxf(cr[3].v[k], 0.0f);
    const float gg = fmaxf(g, 0.0f);
    T* d = dst + 6 * k;
    const bool st0 = sto[2 * k], st1 = sto[2 * k + 1];
    // {R, B} of the green pixel and of the R/B pixel; red_row is wave-uniform: a branch, not six selects
    auto store = [&](float gr_, float gb_, float rr_, float rb_) {
      // column 2 k first: the R/B pixel when the R/B sites of this row sit on even columns
      const float f0 = p == 0 ? rr_ : gr_, f1 = p == 0 ? green : gg, f2 = p == 0 ? rb_ : gb_;
      const float s0 = p == 0 ? gr_ : rr_, s1 = p == 0 ? gg : green, s2 = p == 0 ? gb_ : rb_;
      if (st0 && st1) {
#if defined(TDK_EXPERIMENTS) && defined(TDK_RQ_FAKE_LAB)
        // timing experiment only (profiles/rcd_lab_fusion_exp.py; the caller's buffer holds 12 bytes per pixel): what it would cost this
        // kernel to emit log-lightness + Lab chroma instead of RGB, i.e. to absorb lum_lab_extract
        if constexpr (sizeof(T) == 2) {
          const f3 l0 = cA::rgb_to_lab(mk3(f0, f1, f2)), l1 = cA::rgb_to_lab(mk3(s0, s1, s2));
          struct alignas(8) px6 { float a, b, c, d, e, f; };
          *reinterpret_cast<px6*>(reinterpret_cast<float*>(dst) + 6 * k) =
              px6{tdk_log(fmaxf(1e-4f, fmaxf(l0.x, 0.0f))), l0.y, l0.z, tdk_log(fmaxf(1e-4f, fmaxf(l1.x, 0.0f))), l1.y, l1.z};
        } else
#endif
        if constexpr (sizeof(T) == 4) {
          struct alignas(8) px6 { float a, b, c, d, e, f; };
          *reinterpret_cast<px6*>(d) = px6{f0, f1, f2, s0, s1, s2};
        } else {
          struct alignas(4) pair6 { __half2 a, b, c; };
          *reinterpret_cast<pair6*>(d) = pair6{__floats2half2_rn(f0, f1), __floats2half2_rn(f2, s0), __floats2half2_rn(s1, s2)};
        }
      } else if (st0) {  // the frame's last stored column is even (w - 8) ...
        ::st(d, 0, f0); ::st(d, 1, f1); ::st(d, 2, f2);
      } else if (st1) {  // ... its first one odd (7)
        ::st(d, 3, s0); ::st(d, 4, s1); ::st(d, 5, s2);
      }
    };
    if (red_row) store(own, oth, native, other);
    else store(oth, own, other, native);
  }
}

// Workgroup = one segment of one strip (grid: nstrips * nsegs) as in rs::rcd_stream.  CPL = 4: 256 threads, wave wv, half hf of
// the wave -> row wv + 4 hf of the step's 8-row block, lane q of the half -> window columns 4 q .. 4 q + 3.  CPL = 2: 512 threads,
// wave wv -> row wv, lane q -> columns 2 q, 2 q + 1.
template <int CPL, typename TI, typename T, bool AP = false>
__global__ __launch_bounds__(Geo<CPL>::NT) __attribute__((amdgpu_waves_per_eu(Geo<CPL>::WPE, Geo<CPL>::WPE))) void rcd_quad(
    const TI* __restrict__ in, T* __restrict__ out, int w, int h, uint32_t pattern, int nstrips, int seg_rows, int nbx, int nby) {
  using G = Geo<CPL>;
  constexpr int NT = G::NT, LPR = G::LPR, HPL = G::HPL, SLIDE_PT = G::SLIDE_PT;
  extern __shared__ float lds[];
#ifdef TDK_RCD_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 1024u) g_rcd_wg_times[2 * blockIdx.x] = wall_clock64();
#endif
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hf = lane / LPR, q = lane % LPR;
  const int rr = wv + (RB / 2) * hf;  // row of this thread in a step's block
  const StripMap sm_ = strip_map((int)blockIdx.x, nstrips, (int)gridDim.x / nstrips, 2 * (nbx + nby));
  if ((int)blockIdx.x >= sm_.ring_first && (int)blockIdx.x < sm_.ring_first + sm_.ring_count) {  // workgroup-uniform
    for (int b = (int)blockIdx.x - sm_.ring_first; b < 2 * (nbx + nby); b += sm_.ring_count) {
      ring_piece(in, out, w, h, pattern, nbx, nby, b, lds, NT);
      __syncthreads();
    }
  }
  const int strip = sm_.strip, seg = sm_.seg;
  const int xs = min(strip * TWS, w - TWS), ys = min(seg * seg_rows, h - seg_rows);
  const int gx0 = xs - HALO, gy0 = ys - HALO;  // frame position of window column 0 / row 0; both even
  const int nsteps = (seg_rows + 2 * HALO + RB - 1) / RB;
  Lane t;
  t.bF = lds + rr * 128 + CPL * q;
  t.bH = lds + rr * 64 + HPL * q;
  t.gxq = gx0 + CPL * q;
  t.w = w; t.h = h;
  uint32_t* verdict = reinterpret_cast<uint32_t*>(lds + LDS_FLOATS);

  // R/B column parity of this wave's rows in the even-lag steps (rows wv and wv + 4: the same CFA phase)
  const int rowpar0 = cfa_color(0, 0, pattern) & 1, rowpar1 = cfa_color(1, 0, pattern) & 1;
  const int pe = ((gy0 + wv) & 1) ? rowpar1 : rowpar0;
  const bool red_row = cfa_color(gy0 + wv, pe, pattern) == 0;  // colour of the R/B sites in this wave's output rows (lag 10: even)

  // per column of the lane: inside the strip's own columns and the frame's stored range [7, w - 7) (a bit per column, in a
  // VGPR: as lane masks they would cost an SGPR pair each for the whole kernel); the lanes that store anything; and the lanes
  // with a site whose step-5.1 colour a stored pixel can read (3 columns beyond the own range)
  int own7 = 0;
#pragma unroll
  for (int ci = 0; ci < CPL; ci++) {
    const int wc = CPL * q + ci, gx = t.gxq + ci;
    own7 |= (wc >= HALO && wc < HALO + TWS && gx >= 7 && gx < w - 7) ? (1 << ci) : 0;
  }
  asm volatile("" : "+v"(own7));
  const lmask own_lanes = __builtin_amdgcn_ballot_w64(own7 != 0);
  const lmask rel51 = __builtin_amdgcn_ballot_w64(CPL * q + CPL - 1 >= HALO - 4 && CPL * q < HALO + TWS + 4);
  // the frame cuts the window at even columns: a column pair is in or out as a whole
  bool pair_in[HPL];
#pragma unroll
  for (int k = 0; k < HPL; k++) pair_in[k] = t.gxq + 2 * k >= 0 && t.gxq + 2 * k < w;
  const bool inner_cols = gx0 >= 4 && gx0 + 127 <= w - 5;

  int sd[SLIDE_PT], sdist[SLIDE_PT];
#pragma unroll
  for (int k = 0; k < SLIDE_PT; k++) slide_slot(tid + k * NT, sd[k], sdist[k]);
  const bool last_round = tid + (SLIDE_PT - 1) * NT < SLIDE_SLOTS;  // whole waves

  // samples of window row 8 b + rr, the lane's column pairs; outside the frame: 0
  Pair<TI> smp[HPL];
  bool smp_in[HPL];
#pragma unroll
  for (int k = 0; k < HPL; k++) smp_in[k] = false;
  auto prefetch = [&](int b) {
    const int gy = gy0 + RB * b + rr;
    const bool row_in = gy >= 0 && gy < h;
#pragma unroll
    for (int k = 0; k < HPL; k++) {
      smp_in[k] = row_in && pair_in[k];
      if (smp_in[k]) smp[k].fetch(in + (size_t)gy * w + t.gxq + 2 * k);
    }
  };
  prefetch(0);
  bool ok1 = false, ok2 = false;  // range verdicts of the two previous blocks

#ifdef TDK_RCD_TIMING
  unsigned long long rq_t0 = clock64();
#endif
  const int wslot = TDK_FAIR_PRIO ? tdk_wave_slot() : 0;
  for (int b = 0; b < nsteps; b++) {
    tdk_rotate_prio<3>(b, wslot);
    // ---- slide: every plane moves up by 8 rows (its live rows; the rest is rewritten in this step)
    if (b > 0) {
      // (named registers, not an array: across the fence of the barrier an array would be kept in scratch memory)
      static_assert(SLIDE_PT == 5 || SLIDE_PT == 3, "slide slots per thread");
      const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      constexpr int LR = SLIDE_PT - 1;  // the round only two waves take
      const float4 s0 = *slot16(lds + sd[0] + sdist[0]), s1 = *slot16(lds + sd[1] + sdist[1]);
      float4 s2 = zero4, s3 = zero4, sl = zero4;
      if constexpr (SLIDE_PT == 5) { s2 = *slot16(lds + sd[2] + sdist[2]); s3 = *slot16(lds + sd[3] + sdist[3]); }
      if (last_round) sl = *slot16(lds + sd[LR] + sdist[LR]);
      wg_barrier();
      *slot16(lds + sd[0]) = s0;
      *slot16(lds + sd[1]) = s1;
      if constexpr (SLIDE_PT == 5) { *slot16(lds + sd[2]) = s2; *slot16(lds + sd[3]) = s3; }
      if (last_round) *slot16(lds + sd[LR]) = sl;
    }
    // ---- the new CFA rows (max(0, in)), their range verdict, and the next block's samples on their way
    {
      float a[CPL];
      Range rg;
#pragma unroll
      for (int k = 0; k < HPL; k++) {
        a[2 * k] = 0.0f; a[2 * k + 1] = 0.0f;
        if (smp_in[k]) { const float2 v = smp[k].get(); a[2 * k] = fmaxf(0.0f, v.x); a[2 * k + 1] = fmaxf(0.0f, v.y); }
        rg.add(a[2 * k]); rg.add(a[2 * k + 1]);
      }
      stv(t.bF + rowF<CFA_B, CFA_L, CFA_W>(0, 0), a);
      if constexpr (!AP) {
        const bool wave_ok = __builtin_amdgcn_ballot_w64(!rg.ok()) == 0;
        if (lane == 0) verdict[(b & 3) * 8 + wv] = wave_ok ? 1u : 0u;
      }
      if (b + 1 < nsteps) prefetch(b + 1);
    }
    wg_barrier();
    bool fast = true;
    if constexpr (!AP) {
      uint32_t all = 1u;
#pragma unroll
      for (int k = 0; k < NT / 64; k++) all &= verdict[(b & 3) * 8 + k];
      const bool ok0 = __builtin_amdgcn_readfirstlane(all) != 0;
      fast = ok0 && ok1 && ok2;  // the 24 newest CFA rows >= the 21 rows any step of this block reads
      ok2 = ok1; ok1 = ok0;
    }
    // the rows of every step of this block (lags 1 .. 10 behind rows gy0 + 8 b .. + 7) inside every step's row range
    const bool inner = inner_cols && gy0 + RB * b - LAG_52 >= 4 && gy0 + RB * b + RB - 1 - LAG_21 <= h - 5;
    RQ_MARK(0);

    const int gyb = gy0 + RB * b + rr;  // frame row of this thread's lag-0 site
    const int orow = RB * b - LAG_52 + rr, gyo = gy0 + orow;
    const bool rout = orow >= HALO && orow < HALO + seg_rows && gyo >= 7 && gyo < h - 7;
    bool st[CPL];
#pragma unroll
    for (int ci = 0; ci < CPL; ci++) st[ci] = rout && ((own7 >> ci) & 1) != 0;
    const lmask stm = __builtin_amdgcn_ballot_w64(rout) & own_lanes;
#if defined(TDK_EXPERIMENTS) && defined(TDK_RQ_FAKE_LAB)
    T* dst = sizeof(T) == 2 ? reinterpret_cast<T*>(reinterpret_cast<float*>(out) + ((size_t)gyo * w + t.gxq) * 3) : out + ((size_t)gyo * w + t.gxq) * 3;
#else
    T* dst = out + ((size_t)gyo * w + t.gxq) * 3;
#endif

#define RQ_STEP(MODEV, PEV)                                              \
  do {                                                                   \
    q_step_2_1_1_1_4_1<CPL, MODEV, PEV, TI>(t, gyb, in);                 \
    RQ_MARK(9);                                                          \
    wg_barrier();                                                        \
    RQ_MARK(1);                                                          \
    q_step_1_2_4_2<CPL, MODEV, PEV>(t, gyb);                             \
    RQ_MARK(10);                                                         \
    wg_barrier();                                                        \
    RQ_MARK(2);                                                          \
    q_step_3_1<CPL, MODEV, PEV>(t, gyb);                                 \
    RQ_MARK(11);                                                         \
    wg_barrier();                                                        \
    RQ_MARK(3);                                                          \
    q_step_5_1<CPL, MODEV, PEV>(t, gyb, rel51);                          \
    RQ_MARK(12);                                                         \
    wg_barrier();                                                        \
    RQ_MARK(4);                                                          \
    q_step_5_2_out<CPL, MODEV, PEV, T>(t, red_row, dst, st, stm);        \
    RQ_MARK(5);                                                          \
  } while (0)
    if constexpr (AP) {
      if (inner) { if (pe) RQ_STEP(AINNER, 1); else RQ_STEP(AINNER, 0); }
      else { if (pe) RQ_STEP(ABORD, 1); else RQ_STEP(ABORD, 0); }
    } else {
      if (fast && inner) { if (pe) RQ_STEP(INNER, 1); else RQ_STEP(INNER, 0); }
      else if (fast) { if (pe) RQ_STEP(FASTM, 1); else RQ_STEP(FASTM, 0); }
      else { if (pe) RQ_STEP(SLOW, 1); else RQ_STEP(SLOW, 0); }
    }
#undef RQ_STEP
  }
#ifdef TDK_RCD_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 1024u) g_rcd_wg_times[2 * blockIdx.x + 1] = wall_clock64();
#endif
}

}  // namespace rq
