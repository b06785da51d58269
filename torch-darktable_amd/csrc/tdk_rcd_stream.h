// tdk_rcd_stream.h -- RCD as column strips walked down the frame (every frame of at least 128 x 64 whose rows load as sample pairs).
// Included by rcd.hip inside its anonymous namespace (shares div_pos / div_signed, Range, CFA_MIN / CFA_MAX).
//
// Same nine steps and the same expressions as rcd_phases (reference csrc/debayer/rcd.cu:63-282); what changes is the order in
// which sites are visited.  rcd_phases owns a 64 x 64 tile per 1024-thread workgroup: the 10-px dependency halo makes it compute
// 84 x 84 sites per plane (1.72 x the output), its five planes fill the LDS of a CU, and between its seven barriers the 16 waves
// of the one resident workgroup first all wait for LDS and then all compute.  Here a 512-thread workgroup owns a strip of
// 108 output columns (128 computed: halo only left and right, 1.19 x) and walks DOWN the image 8 rows per step.  Each of the
// nine steps runs `lag` rows behind the newest CFA row -- as far as its vertical taps reach:
//     load 0 | 2.1 lpf 1 | 1.1 v/h_diff 3 | 4.1 p/q_diff 3 | 1.2 VH_dir 4 | 4.2 PQ_dir 4 | 3.1 green 5 | 5.1 colour 7 | 5.2 + output 10
// so a plane only has to hold the rows its readers still need (`live`) plus the 8 new ones: 47 KB for all ten planes, three
// workgroups per CU, whose barriers and LDS / VALU phases interleave.  Planes are linear, not rings: after a step every plane
// slides up by 8 rows (its live rows are copied, 6 % extra LDS traffic), which keeps every tap address of every step a
// compile-time offset from two per-thread base registers -- a ring would need an address computation per tap row.
// Thread mapping: wave = row of the step's 8-row block, lane = column PAIR (2 l, 2 l + 1).  Full-density planes are stored
// de-interleaved by column parity (row = 64 even columns, then 64 odd ones), half-density ones (values at R/B sites, or at odd
// columns) compacted, so that every access is unit-stride over the lanes: no bank conflicts at wave-uniform rows.  A lane
// handles both columns of its pair in the full-density steps and the one R/B (or green) site of the pair otherwise.
// A strip is cut into vertical segments (one workgroup each, 10 + 10 rows of warm-up / drain overlap) so that the launch
// fills the chip.  The strips cover the whole frame: sites outside a step's border range hold 0 (Cols / Rows below), the
// [0, 7) ring is staged in pieces by the same workgroups (ring_piece in rcd.hip).  rcd_phases (rcd_interior) keeps the frames
// the strips do not fit.
#pragma once

namespace rs {

#if defined(TDK_EXPERIMENTS) && defined(TDK_RS_RB)
// experiment: TDK_RS_RB rows per step (16: half the barriers and slides per row, two workgroups per CU); rq::rcd_quad only -- the
// host takes it for every strip launch of such a build
constexpr int RB = TDK_RS_RB, NT = 512, TWS = 108, HALO = 10;
constexpr int WG_PER_CU = 2;
#elif defined(TDK_EXPERIMENTS) && defined(TDK_RS_WG_PER_CU)
// timing experiment only (wrong results): register budget for TDK_RS_WG_PER_CU workgroups per CU; the launcher is given a smaller
// LDS allocation through TDK_RCD_LDS_PAD (accesses beyond it are dropped by the hardware)
constexpr int RB = 8, NT = 512, TWS = 108, HALO = 10;
constexpr int WG_PER_CU = TDK_RS_WG_PER_CU;
#define TDK_RS_NO_LDS_ASSERT 1
#else
constexpr int RB = 8, NT = 512, TWS = 108, HALO = 10;
constexpr int WG_PER_CU = 3;
#endif

// planes: base (floats), live rows (rows older than the step's 8 new ones that a reader still needs), writer's lag
constexpr int PAD = 8;
constexpr int CFA_L = 13, VD_L = 2, HD_L = 1, VH_L = 7, LPF_L = 6, GRN_L = 6, P_L = 2, Q_L = 2, PQ_L = 4, COL_L = 6;
constexpr int CFA_W = 0, VD_W = 3, HD_W = 3, VH_W = 4, LPF_W = 1, GRN_W = 5, P_W = 3, Q_W = 3, PQ_W = 4, COL_W = 7;
constexpr int CFA_B = PAD;
constexpr int VD_B = CFA_B + (CFA_L + RB) * 128;
constexpr int HD_B = VD_B + (VD_L + RB) * 128;
constexpr int VH_B = HD_B + (HD_L + RB) * 128;
constexpr int LPF_B = VH_B + (VH_L + RB) * 128;
constexpr int GRN_B = LPF_B + (LPF_L + RB) * 64;
constexpr int P_B = GRN_B + (GRN_L + RB) * 64;
constexpr int Q_B = P_B + (P_L + RB) * 64;
constexpr int PQ_B = Q_B + (Q_L + RB) * 64;
constexpr int COL_B = PQ_B + (PQ_L + RB) * 64;
constexpr int LDS_FLOATS = COL_B + (COL_L + RB) * 64 + PAD;
constexpr int VERDICT_WORDS = 4 * 8;  // [step & 3][wave]
constexpr size_t LDS_BYTES = (size_t)(LDS_FLOATS + VERDICT_WORDS) * sizeof(float);
#ifndef TDK_RS_NO_LDS_ASSERT
static_assert(WG_PER_CU * LDS_BYTES <= 160 * 1024, "three workgroups per CU");
#endif

// lags of the steps (rows behind the newest CFA row)
constexpr int LAG_21 = 1, LAG_11 = 3, LAG_41 = 3, LAG_12 = 4, LAG_42 = 4, LAG_31 = 5, LAG_51 = 7, LAG_52 = 10;

constexpr int fdiv2(int v) { return v >= 0 ? v / 2 : -((1 - v) / 2); }  // floor(v / 2)

// Offset (floats, relative to the lane bases below) of a tap.  `lag` = lag of the READING step; a writer passes its own lag.
//  full-density plane: tap (dr, dc) seen from a site in a column of parity p
template <int BASE, int LIVE, int LAGW> constexpr int f128(int lag, int dr, int p, int dc) {
  return BASE + (LIVE + LAGW - lag + dr) * 128 + ((p + dc) & 1) * 64 + fdiv2(p + dc);
}
//  compacted plane: row dr, entry `shift` away from the lane's own entry
template <int BASE, int LIVE, int LAGW> constexpr int h64(int lag, int dr, int shift) { return BASE + (LIVE + LAGW - lag + dr) * 64 + shift; }

// The slide: float4 slots of all live rows, in plane order; slot k of a plane moves 8 rows up.
struct SlidePlane { int base, rowlen, live; };
constexpr SlidePlane SLIDE[10] = {{CFA_B, 128, CFA_L}, {VD_B, 128, VD_L}, {HD_B, 128, HD_L}, {VH_B, 128, VH_L}, {LPF_B, 64, LPF_L},
                                  {GRN_B, 64, GRN_L},  {P_B, 64, P_L},    {Q_B, 64, Q_L},    {PQ_B, 64, PQ_L},  {COL_B, 64, COL_L}};
constexpr int slide_slots() {
  int n = 0;
  for (int p = 0; p < 10; p++) n += SLIDE[p].live * SLIDE[p].rowlen / 4;
  return n;
}
constexpr int SLIDE_SLOTS = slide_slots();  // 1152
static_assert(SLIDE_SLOTS <= 3 * NT, "three slots per thread");

// A slide slot as one 16-byte LDS access.  hipcc takes the alignment of an access to `extern __shared__ float lds[]` from
// that declaration (4 bytes), whatever the pointer is cast to, and splits a float4 into two ds_read2_b32 / ds_write2_b32 whose
// lanes are 16 bytes apart: a 4-way bank conflict on every one of them (a quarter of this kernel's LDS cycles in round 3's
// counters).  Every plane base and row length is a multiple of 4 floats and the dynamic LDS base is 16-byte aligned (no
// static __shared__ in this kernel), so the promise below holds and the slide moves as ds_read_b128 / ds_write_b128.
__device__ __forceinline__ float4* slot16(float* p) { return reinterpret_cast<float4*>(__builtin_assume_aligned(p, 16)); }
static_assert(PAD % 4 == 0, "plane bases are multiples of 4 floats");

// destination offset (floats) of slide slot k and the distance to its source; dst < 0: no slot
__device__ __forceinline__ void slide_slot(int k, int& dst, int& dist) {
  dst = -1; dist = 0;
  int start = 0;
#pragma unroll
  for (int p = 0; p < 10; p++) {
    const int n = SLIDE[p].live * SLIDE[p].rowlen / 4;
    if (k >= start && k < start + n) { dst = SLIDE[p].base + (k - start) * 4; dist = RB * SLIDE[p].rowlen; }
    start += n;
  }
}

// Border rules (the `row/col >= k && <= size - k` ranges of rcd.cu's kernels): a site outside a step's range holds 0, as in the
// reference's zero-initialised planes.  The column tests are made once per thread and kept as 64-bit LANE MASKS (wave-uniform
// values: SGPR pairs), the row tests are wave-uniform and made per step; a site's rule is `row ? cols : 0`, scalar work.
// The select that applies it is written as VOP3 v_cndmask with the SGPR pair as its mask operand (keep()): left to hipcc, a
// mask that the scalar unit produced is copied to VCC for the short VOP2 form -- and a VOP2 v_cndmask reading a VCC that the
// SALU wrote costs ~23 SIMD cycles on gfx950 instead of ~4.4 (tests/hip_unit/select_bench.hip, profiles/r04/select_bench.txt);
// round 3's strips carried about fifteen of those per step.
using lmask = unsigned long long;
struct Cols {
  lmask c3[2];   // [3, w - 4]   steps 1.1, 4.1
  lmask c2a[2];  // [2, w - 3]   steps 1.2, 4.2
  lmask c2b[2];  // [2, w - 2]   step 2.1
  lmask c4a[2];  // [4, w - 5]   step 3.1
  lmask c4b[2];  // [4, w - 4]   step 5.1
  lmask img;     // the pair lies inside the frame
  bool o7[2];    // [7, w - 7) and one of the strip's own columns: the pixel is stored (per lane: it guards stores)
};
struct Rows {  // of the current block, for this wave
  bool r21, r11, r12, r31, r51, rimg41, rout;
};
__device__ __forceinline__ lmask rule(bool row_ok, lmask cols) { return row_ok ? cols : 0ull; }
// v where the lane's bit of m is set, +0 elsewhere
// (s_nop 1: two wait states between a VALU write of the mask -- a ballot straight from v_cmp -- and this read of it as a lane
// mask, which hipcc cannot see inside the asm; scalar-made masks need none, and the rules only exist in border blocks)
__device__ __forceinline__ float keep(lmask m, float v) {
  float r;
  asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
  return r;
}

// The three code variants of a step (MODE):
//   SLOW   every border rule, IEEE divisions: blocks whose 24 newest CFA rows fail the range check of the exact fast division
//   FASTM  every border rule, the exact fast division (tdk_fastdiv.h) except in step 4.2: border blocks with samples in range
//   INNER  no border rule at all and the fast division everywhere: blocks whose window keeps clear of every `>= k, <= size - k`
//          range of the nine steps (columns [gx0, gx0 + 128) inside [4, w - 5], the rows of all lags inside [4, h - 5]) and
//          therefore also of every stale p/q slot (columns 1, w - 3, w - 1, rows < 3 or > h - 4: step 4.1).  At 12 MP that is
//          36 of the 38 strips and all but the first / last two blocks of a strip's first / last segment.
//   ABORD  every border rule, approximate arithmetic (below)  \  fp16 results only (TDK_F16 output without TDK_RCD_EXACT): no
//   AINNER no border rule, approximate arithmetic               /  range check, no IEEE fallback, one code path per block kind
constexpr int SLOW = 0, FASTM = 1, INNER = 2, ABORD = 3, AINNER = 4;
constexpr bool has_rules(int mode) { return mode == SLOW || mode == FASTM || mode == ABORD; }
constexpr bool approx(int mode) { return mode >= ABORD; }

// ---- the arithmetic of the nine steps in two flavours.
// EXACT (AP = false): the oracle's operation order, no contraction, correctly rounded quotients -- the same bits as
// oracle/src/rcd.c, whatever the storage type of the result.
// APPROXIMATE (AP = true): what a result that is ROUNDED TO BINARY16 at the store can afford, and what the reference itself
// does -- it is an nvcc --use_fast_math build (setup.py:36: approximate division, contraction).  Quotients are a * v_rcp_f32(b)
// (1 ulp reciprocal; every denominator of the nine steps is >= 1e-10 by construction, so no range wrapper is needed), sums of
// products are fused, mix(a, b, t) is a + t (b - a).  Each differs from the exact form by a few fp32 ulps: measured against the
// oracle on the 12 MP test frames <= 4e-7 absolute before the store, i.e. the binary16 result differs from the oracle's in
// ~5e-5 of the values, by one binary16 ulp (the fp32 value sat next to a rounding boundary), and no selection flips were seen
// (tests/test_gpu_parity.py::test_rcd_fp16_fast_arithmetic).  23 % fewer VALU instructions per pixel than the exact flavour.
template <bool AP> __device__ __forceinline__ float hp7(float m3, float m2, float m1, float c, float p1, float p2, float p3) {
  if constexpr (AP) return __builtin_fmaf(6.0f, c, __builtin_fmaf(-3.0f, m2 + p2, ((m3 - m1) - p1) + p3));
  else return m3 - 3.0f * m2 - m1 + 6.0f * c - p1 - 3.0f * p2 + p3;
}
template <bool AP> __device__ __forceinline__ float dg7(float m3, float m2, float m1, float c, float p1, float p2, float p3) {
  if constexpr (AP) return __builtin_fmaf(6.0f, c, __builtin_fmaf(-3.0f, m2 + p2, ((m3 - m1) - p1) + p3));
  else return (m3 - m1 - p1 + p3) - 3.0f * (m2 + p2) + 6.0f * c;
}
template <bool AP> __device__ __forceinline__ float lpf9(float c, float n, float s, float w, float e, float nw, float ne, float sw, float se) {
  if constexpr (AP) return __builtin_fmaf(0.5f, (n + s) + (w + e), __builtin_fmaf(0.25f, (nw + ne) + (sw + se), c));
  else return c + 0.5f * (n + s + w + e) + 0.25f * (nw + ne + sw + se);
}
template <bool AP> __device__ __forceinline__ float dot2(float a, float b, float c, float d) {
  if constexpr (AP) return __builtin_fmaf(a, b, c * d);
  else return a * b + c * d;
}
template <bool AP> __device__ __forceinline__ float mixq(float a, float b, float t) {
  if constexpr (AP) return __builtin_fmaf(t, b - a, a);
  else return mixf(a, b, t);
}
__device__ __forceinline__ float div_rcp(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
// quotient of non-negative operands: steps 1.2, 3.1 (and 4.2 through qdiv42: the exact fast division only where no stale slot is near)
template <int MODE> __device__ __forceinline__ float qdiv(float a, float b) {
  if constexpr (approx(MODE)) return div_rcp(a, b);
  else return div_pos<MODE != SLOW>(a, b);
}
template <int MODE> __device__ __forceinline__ float qdiv42(float a, float b) {
  if constexpr (approx(MODE)) return div_rcp(a, b);
  else return div_pos<MODE == INNER>(a, b);
}

// One step of one wave.  PE = column parity of the R/B sites in this wave's rows of the EVEN-lag steps (odd-lag steps see the
// other parity: consecutive rows alternate).  b128 / b64: LDS + w * 128 + l / LDS + w * 64 + l.
template <int MODE, int PE, typename TI>
__device__ __forceinline__ void step_2_1_1_1_4_1(float* __restrict__ b128, float* __restrict__ b64, const Cols& cv, const Rows& rv, const TI* __restrict__ in,
                                                 int gx_odd, int gy41, int w, int h) {
  // ---- step 2.1 (lag 1): lpf at the R/B site of the pair
  {
    constexpr int L = LAG_21, p = PE ^ (L & 1);
    auto a = [&](int dr, int dc) { return b128[f128<CFA_B, CFA_L, CFA_W>(L, dr, p, dc)]; };
    const float v = lpf9<approx(MODE)>(a(0, 0), a(-1, 0), a(1, 0), a(0, -1), a(0, 1), a(-1, -1), a(-1, 1), a(1, -1), a(1, 1));
    if constexpr (has_rules(MODE)) b64[h64<LPF_B, LPF_L, LPF_W>(L, 0, 0)] = keep(rule(rv.r21, cv.c2b[p]), v);
    else b64[h64<LPF_B, LPF_L, LPF_W>(L, 0, 0)] = v;
  }
  // ---- step 1.1 (lag 3): v_diff / h_diff at both columns
  {
    constexpr int L = LAG_11;
#pragma unroll
    for (int p = 0; p < 2; p++) {
      auto a = [&](int dr, int dc) { return b128[f128<CFA_B, CFA_L, CFA_W>(L, dr, p, dc)]; };
      const float vd = sqf(hp7<approx(MODE)>(a(-3, 0), a(-2, 0), a(-1, 0), a(0, 0), a(1, 0), a(2, 0), a(3, 0)));
      const float hd = sqf(hp7<approx(MODE)>(a(0, -3), a(0, -2), a(0, -1), a(0, 0), a(0, 1), a(0, 2), a(0, 3)));
      if constexpr (has_rules(MODE)) {
        const lmask ok = rule(rv.r11, cv.c3[p]);
        b128[f128<VD_B, VD_L, VD_W>(L, 0, p, 0)] = keep(ok, vd);
        b128[f128<HD_B, HD_L, HD_W>(L, 0, p, 0)] = keep(ok, hd);
      } else {
        b128[f128<VD_B, VD_L, VD_W>(L, 0, p, 0)] = vd;
        b128[f128<HD_B, HD_L, HD_W>(L, 0, p, 0)] = hd;
      }
    }
  }
  // ---- step 4.1 (lag 3): p/q_diff at the odd column
  {
    constexpr int L = LAG_41;
    auto a = [&](int dr, int dc) { return b128[f128<CFA_B, CFA_L, CFA_W>(L, dr, 1, dc)]; };
    const float pd = sqf(dg7<approx(MODE)>(a(-3, -3), a(-2, -2), a(-1, -1), a(0, 0), a(1, 1), a(2, 2), a(3, 3)));
    const float qd = sqf(dg7<approx(MODE)>(a(-3, 3), a(-2, 2), a(-1, 1), a(0, 0), a(1, -1), a(2, -2), a(3, -3)));
    float pv = pd, qv = qd;
    if constexpr (has_rules(MODE)) {
      // slots step 4.1 does not write keep the same call's v_diff / h_diff of the shared buffer (rcd.cu:637-652; stale_diff in
      // rcd.hip); outside the frame: 0
      const lmask inside = rule(rv.rimg41, cv.img), stale = inside & ~rule(rv.r11, cv.c3[1]);
      if (stale != 0) {
        if ((stale >> (threadIdx.x & 63)) & 1) stale_pair(in, gy41, gx_odd, w, h, pv, qv);
      }
      pv = keep(inside, pv);
      qv = keep(inside, qv);
    }
    b64[h64<P_B, P_L, P_W>(L, 0, 0)] = pv;
    b64[h64<Q_B, Q_L, Q_W>(L, 0, 0)] = qv;
  }
}

template <int MODE, int PE>
__device__ __forceinline__ void step_1_2_4_2(float* __restrict__ b128, float* __restrict__ b64, const Cols& cv, const Rows& rv) {
  // ---- step 1.2 (lag 4): VH_dir at both columns
  {
    constexpr int L = LAG_12;
#pragma unroll
    for (int p = 0; p < 2; p++) {
      auto vd = [&](int dr) { return b128[f128<VD_B, VD_L, VD_W>(L, dr, p, 0)]; };
      auto hd = [&](int dc) { return b128[f128<HD_B, HD_L, HD_W>(L, 0, p, dc)]; };
      const float eps = 1e-10f;
      const float V_Stat = fmaxf(eps, vd(-1) + vd(0) + vd(1));
      const float H_Stat = fmaxf(eps, hd(-1) + hd(0) + hd(1));
      const float vh = qdiv<MODE>(V_Stat, V_Stat + H_Stat);
      if constexpr (has_rules(MODE)) b128[f128<VH_B, VH_L, VH_W>(L, 0, p, 0)] = keep(rule(rv.r12, cv.c2a[p]), vh);
      else b128[f128<VH_B, VH_L, VH_W>(L, 0, p, 0)] = vh;
    }
  }
  // ---- step 4.2 (lag 4): PQ_dir at the R/B site (column 2 l + p).  p/q slot of odd column 2 j + 1 = entry j; the slots of
  // (col - 1) | 1 on the neighbour rows: j = l - 1 + p (rcd.cu:166-182)
  {
    constexpr int L = LAG_42, p = PE ^ (L & 1), jm = p - 1;
    auto P = [&](int dr, int sh) { return b64[h64<P_B, P_L, P_W>(L, dr, sh)]; };
    auto Q = [&](int dr, int sh) { return b64[h64<Q_B, Q_L, Q_W>(L, dr, sh)]; };
    const float eps = 1e-10f;
    const float P_Stat = fmaxf(eps, P(-1, jm) + P(0, 0) + P(1, jm + 1));
    const float Q_Stat = fmaxf(eps, Q(-1, jm + 1) + Q(0, 0) + Q(1, jm));
    // plain division wherever a stale slot (see step 4.1) can be near: it holds values of samples no range check has seen
    const float pq = qdiv42<MODE>(P_Stat, P_Stat + Q_Stat);
    if constexpr (has_rules(MODE)) b64[h64<PQ_B, PQ_L, PQ_W>(L, 0, 0)] = keep(rule(rv.r12, cv.c2a[p]), pq);
    else b64[h64<PQ_B, PQ_L, PQ_W>(L, 0, 0)] = pq;
  }
}

// ---- step 3.1 (lag 5): green at the R/B site
template <int MODE, int PE>
__device__ __forceinline__ void step_3_1(float* __restrict__ b128, float* __restrict__ b64, const Cols& cv, const Rows& rv) {
  constexpr int L = LAG_31, p = PE ^ (L & 1);
  auto a = [&](int dr, int dc) { return b128[f128<CFA_B, CFA_L, CFA_W>(L, dr, p, dc)]; };
  auto vh = [&](int dr, int dc) { return b128[f128<VH_B, VH_L, VH_W>(L, dr, p, dc)]; };
  auto lp = [&](int dr, int sh) { return b64[h64<LPF_B, LPF_L, LPF_W>(L, dr, sh)]; };
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
  const float grn = mixq<approx(MODE)>(V_Est, H_Est, VH_Disc);
  if constexpr (has_rules(MODE)) b64[h64<GRN_B, GRN_L, GRN_W>(L, 0, 0)] = keep(rule(rv.r31, cv.c4a[p]), grn);
  else b64[h64<GRN_B, GRN_L, GRN_W>(L, 0, 0)] = grn;
}

// ---- step 5.1 (lag 7): the opposite colour at the R/B site.  `lanes_ok`: lanes whose numerators count in the wave's
// fast-division test (the outermost columns compute on garbage, which nothing that is stored ever reads).
template <int MODE, int PE>
__device__ __forceinline__ void step_5_1(float* __restrict__ b128, float* __restrict__ b64, lmask lanes_ok, const Cols& cv, const Rows& rv) {
  constexpr int L = LAG_51, p = PE ^ (L & 1);
  auto a = [&](int dr, int dc) { return b128[f128<CFA_B, CFA_L, CFA_W>(L, dr, p, dc)]; };
  auto pq = [&](int dr, int sh) { return b64[h64<PQ_B, PQ_L, PQ_W>(L, dr, sh)]; };
  // green of R/B site (row + dr, col + dc): rows of the same parity keep the lane's entry shifted by dc / 2, the other rows
  // hold their R/B sites on the other column parity: entry l + (p + dc - (1 - p)) / 2
  auto G = [&](int dr, int dc) { return b64[h64<GRN_B, GRN_L, GRN_W>(L, dr, (dr & 1) ? fdiv2(2 * p + dc - 1) : dc / 2)]; };
  constexpr int s = p - 1;  // entry of slot (col - 1) / 2 on the neighbour rows (rcd.cu:199-207)
  const float eps = 1e-5f;
  const float PQ_c = pq(0, 0);
  const float PQ_n = 0.25f * (pq(-1, s) + pq(-1, s + 1) + pq(1, s) + pq(1, s + 1));
  const float PQ_Disc = (fabsf(0.5f - PQ_c) < fabsf(0.5f - PQ_n)) ? PQ_n : PQ_c;
  const float g0 = G(0, 0);
  const float NW_Grad = eps + fabsf(a(-1, -1) - a(1, 1)) + fabsf(a(-1, -1) - a(-3, -3)) + fabsf(g0 - G(-2, -2));
  const float NE_Grad = eps + fabsf(a(-1, 1) - a(1, -1)) + fabsf(a(-1, 1) - a(-3, 3)) + fabsf(g0 - G(-2, 2));
  const float SW_Grad = eps + fabsf(a(-1, 1) - a(1, -1)) + fabsf(a(1, -1) - a(3, -3)) + fabsf(g0 - G(2, -2));
  const float SE_Grad = eps + fabsf(a(-1, -1) - a(1, 1)) + fabsf(a(1, 1) - a(3, 3)) + fabsf(g0 - G(2, 2));
  const float NW_Est = a(-1, -1) - G(-1, -1);
  const float NE_Est = a(-1, 1) - G(-1, 1);
  const float SW_Est = a(1, -1) - G(1, -1);
  const float SE_Est = a(1, 1) - G(1, 1);
  float num[2] = {dot2<approx(MODE)>(NW_Grad, SE_Est, SE_Grad, NW_Est), dot2<approx(MODE)>(NE_Grad, SW_Est, SW_Grad, NE_Est)};
  float den[2] = {NW_Grad + SE_Grad, NE_Grad + SW_Grad};
  const lmask ok = has_rules(MODE) ? rule(rv.r51, cv.c4b[p]) : ~0ull;
  float est[2];  // P_Est, Q_Est
  if constexpr (approx(MODE)) { est[0] = div_rcp(num[0], den[0]); est[1] = div_rcp(num[1], den[1]); }
  else div_signed<MODE != SLOW>(num, den, est, lanes_ok & ok);
  const float colv = g0 + mixq<approx(MODE)>(est[0], est[1], PQ_Disc);
  if constexpr (has_rules(MODE)) b64[h64<COL_B, COL_L, COL_W>(L, 0, 0)] = keep(ok, colv);
  else b64[h64<COL_B, COL_L, COL_W>(L, 0, 0)] = colv;
}

// ---- step 5.2 (lag 10) at the green site of the pair + the two finished pixels of the pair
template <int MODE, int PE, typename T>
__device__ __forceinline__ void step_5_2_out(float* __restrict__ b128, float* __restrict__ b64, bool red_row, T* __restrict__ dst, const Cols& cv, const Rows& rv) {
  const bool st_e = rv.rout && cv.o7[0], st_o = rv.rout && cv.o7[1];  // column 2 l / 2 l + 1 is stored
  const lmask lanes_ok = __builtin_amdgcn_ballot_w64(st_e || st_o);
  constexpr int L = LAG_52, p = PE ^ (L & 1), pg = 1 - p;  // R/B sites on parity p, the green site of the pair on pg
  auto a = [&](int dr, int dc) { return b128[f128<CFA_B, CFA_L, CFA_W>(L, dr, pg, dc)]; };
  auto vh = [&](int dr, int dc) { return b128[f128<VH_B, VH_L, VH_W>(L, dr, pg, dc)]; };
  auto grn = [&](int dr, int sh) { return b64[h64<GRN_B, GRN_L, GRN_W>(L, dr, sh)]; };
  auto col = [&](int dr, int sh) { return b64[h64<COL_B, COL_L, COL_W>(L, dr, sh)]; };
  const float eps = 1e-5f;
  const float VH_c = vh(0, 0);
  const float VH_n = 0.25f * (vh(-1, -1) + vh(-1, 1) + vh(1, -1) + vh(1, 1));
  const float VH_Disc = (fabsf(0.5f - VH_c) < fabsf(0.5f - VH_n)) ? VH_n : VH_c;
  const float g = a(0, 0);
  const float N1 = eps + fabsf(g - a(-2, 0));
  const float S1 = eps + fabsf(g - a(2, 0));
  const float W1 = eps + fabsf(g - a(0, -2));
  const float E1 = eps + fabsf(g - a(0, 2));
  // green at the four R/B neighbours: above / below the lane's own entry (those rows hold their R/B sites on this column's
  // parity), left / right the entries of columns col -+ 1 of this row
  const float gN = grn(-1, 0), gS = grn(1, 0), gW = grn(0, -p), gE = grn(0, 1 - p);
  // colour of this row's R/B sites (`own`): native left / right, from step 5.1 above / below; the other colour the other way round
  float num[4], den[4], est[4];  // V_Est, H_Est of `own`, then of the other colour
#pragma unroll
  for (int ci = 0; ci < 2; ci++) {
    float cN, cS, cW, cE, cN3, cS3, cW3, cE3;
    if (ci == 0) {
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
    num[2 * ci] = dot2<approx(MODE)>(N_Grad, S_Est, S_Grad, N_Est);
    den[2 * ci] = N_Grad + S_Grad;
    num[2 * ci + 1] = dot2<approx(MODE)>(E_Grad, W_Est, W_Grad, E_Est);
    den[2 * ci + 1] = E_Grad + W_Grad;
  }
  if constexpr (approx(MODE)) {
#pragma unroll
    for (int i = 0; i < 4; i++) est[i] = div_rcp(num[i], den[i]);
  } else {
    div_signed<MODE != SLOW>(num, den, est, lanes_ok);
  }
  const float own = fmaxf(g + mixq<approx(MODE)>(est[0], est[1], VH_Disc), 0.0f), oth = fmaxf(g + mixq<approx(MODE)>(est[2], est[3], VH_Disc), 0.0f);
  // the R/B pixel of the pair: native, green from step 3.1, other colour from step 5.1
  const float native = fmaxf(b128[f128<CFA_B, CFA_L, CFA_W>(L, 0, p, 0)], 0.0f), green = fmaxf(grn(0, 0), 0.0f), other = fmaxf(col(0, 0), 0.0f);
  const float gg = fmaxf(g, 0.0f);
  // {R, G, B} of the green pixel and of the R/B pixel; red_row is wave-uniform: a branch, not six selects
  auto store = [&](float gr, float gb, float rr, float rb) {
    // column 2 l first: the R/B pixel when the R/B sites of this row sit on even columns
    const float f0 = p == 0 ? rr : gr, f1 = p == 0 ? green : gg, f2 = p == 0 ? rb : gb;
    const float s0 = p == 0 ? gr : rr, s1 = p == 0 ? gg : green, s2 = p == 0 ? gb : rb;
    if (st_e && st_o) {
      if constexpr (sizeof(T) == 4) {
        struct alignas(8) px6 { float a, b, c, d, e, f; };
        *reinterpret_cast<px6*>(dst) = px6{f0, f1, f2, s0, s1, s2};
      } else {
        struct alignas(4) pair6 { __half2 a, b, c; };
        *reinterpret_cast<pair6*>(dst) = pair6{__floats2half2_rn(f0, f1), __floats2half2_rn(f2, s0), __floats2half2_rn(s1, s2)};
      }
    } else if (st_e) {  // the frame's last stored column is even (w - 8) ...
      st(dst, 0, f0); st(dst, 1, f1); st(dst, 2, f2);
    } else if (st_o) {  // ... its first one odd (7)
      st(dst, 3, s0); st(dst, 4, s1); st(dst, 5, s2);
    }
  };
  if (red_row) store(own, oth, native, other);
  else store(oth, own, other, native);
}

// Workgroup number -> (strip, segment), and which workgroups stage the ring pieces.
// Every workgroup of the launch is resident from the start (<= 3 per CU), and the issue arbiter serves the OLDEST wave of a SIMD
// first: the k-th workgroup of a CU in launch order ends in the k-th of three classes (73 / 97 / 115 us at 12 MP,
// profiles/r05/experiments/wg_lifetimes.txt), and the launch ends with its slowest workgroup.  The slow ones are known in advance
// -- the two border strips (every step runs the variant with border rules and stale slots: +25 %) and the first / last segment of
// the other strips (border rows) -- so they get the LOWEST numbers and with them the fast class; the ring pieces (a few us each)
// go to workgroups behind them, two per workgroup, so that none of them lands in the last class.
struct StripMap {
  int strip, seg;
  int ring_first, ring_count;  // workgroups [ring_first, ring_first + ring_count) stage the ring pieces, piece p on ring_first + p % ring_count
};
__device__ __forceinline__ StripMap strip_map(int b, int nstrips, int nsegs, int npieces) {
  StripMap m;
  const int nwg = nstrips * nsegs;
  if (nstrips < 3 || nsegs < 3) {  // small frames: plain order
    m.strip = b % nstrips; m.seg = b / nstrips;
    m.ring_first = 0; m.ring_count = nwg;
    return m;
  }
  const int ni = nstrips - 2, e1 = 2 * nsegs, e2 = 2 * ni;
  if (b < e1) { m.strip = (b & 1) ? nstrips - 1 : 0; m.seg = b >> 1; }
  else if (b < e1 + e2) { const int i = b - e1; m.strip = 1 + (i >> 1); m.seg = (i & 1) ? nsegs - 1 : 0; }
  else { const int i = b - e1 - e2; m.seg = 1 + i / ni; m.strip = 1 + i - (m.seg - 1) * ni; }
  m.ring_first = e1 + e2;
  const int rest = nwg - m.ring_first, want = (npieces + 1) / 2;
  m.ring_count = rest < want ? rest : (want > 0 ? want : 1);
  return m;
}

template <typename TI> struct Pair;
template <> struct Pair<float> {
  float2 v;
  __device__ __forceinline__ void fetch(const float* p) { v = *reinterpret_cast<const float2*>(p); }
  __device__ __forceinline__ float2 get() const { return v; }
};
template <> struct Pair<__half> {
  uint32_t v;
  __device__ __forceinline__ void fetch(const __half* p) { v = *reinterpret_cast<const uint32_t*>(p); }
  __device__ __forceinline__ float2 get() const { return __half22float2(__builtin_bit_cast(__half2, v)); }
};

__device__ __forceinline__ void wg_barrier() {  // orders LDS traffic only: the prefetched samples stay in flight across it
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Workgroup = one segment of one strip (grid: nstrips * nsegs), covering the whole frame; the [0, 7) border ring is staged in
// pieces (ring_piece in rcd.hip: 30 registers; border_pixel's nine-neighbour form cost this kernel two waves per SIMD).
// Requires: w even and >= TWS + 2 HALO, base pointer aligned for pair loads (host-checked).
// AP: the approximate arithmetic flavour (above) -- fp16 results only.
template <typename TI, typename T, bool AP = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6, 6))) void rcd_stream(const TI* __restrict__ in, T* __restrict__ out, int w, int h,
                                                                                              uint32_t pattern, int nstrips, int seg_rows, int nbx, int nby) {
  extern __shared__ float lds[];
  // the wave number through readfirstlane: everything derived from it (the row rules above all) is then scalar work for the
  // compiler, which cannot prove threadIdx.x >> 6 wave-uniform
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  // the [0, 7) border ring (independent of the strips: disjoint output pixels, input read-only): one piece per workgroup until
  // the pieces run out, staged through the plane area
  const StripMap sm_ = strip_map((int)blockIdx.x, nstrips, (int)gridDim.x / nstrips, 2 * (nbx + nby));
  if ((int)blockIdx.x >= sm_.ring_first && (int)blockIdx.x < sm_.ring_first + sm_.ring_count) {  // workgroup-uniform
    for (int b = (int)blockIdx.x - sm_.ring_first; b < 2 * (nbx + nby); b += sm_.ring_count) {
      ring_piece(in, out, w, h, pattern, nbx, nby, b, lds, NT);
      __syncthreads();
    }
  }

  const int strip = sm_.strip, seg = sm_.seg;
  // the last strip / segment is moved back so that it ends at the frame's edge (it recomputes what its neighbour also writes)
  const int xs = min(strip * TWS, w - TWS), ys = min(seg * seg_rows, h - seg_rows);
  const int gx0 = xs - HALO, gy0 = ys - HALO;  // frame position of window column 0 / row 0; both even
  const int nsteps = (seg_rows + 2 * HALO + RB - 1) / RB;
  float* b128 = lds + wv * 128 + l;
  float* b64 = lds + wv * 64 + l;
  uint32_t* verdict = reinterpret_cast<uint32_t*>(lds + LDS_FLOATS);

  // row parity: R/B column parity of window row (wv - lag) for even lag
  const int rowpar0 = cfa_color(0, 0, pattern) & 1, rowpar1 = cfa_color(1, 0, pattern) & 1;
  const int pe = ((gy0 + wv) & 1) ? rowpar1 : rowpar0;
  // colour of the R/B sites in this wave's output rows (lag 10: even)
  const bool red_row = cfa_color(gy0 + wv, pe, pattern) == 0;
  const lmask lanes_51 = __builtin_amdgcn_ballot_w64(l >= 3 && l < 61);  // the pairs whose step-5.1 colour a stored pixel can read (3 columns away)
  const bool own = l >= HALO / 2 && l < (HALO + TWS) / 2;  // the strip's own pairs

  Cols cv;
  const int gxe = gx0 + 2 * l;
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int gx = gxe + p;
    cv.c3[p] = __builtin_amdgcn_ballot_w64(gx >= 3 && gx <= w - 4);
    cv.c2a[p] = __builtin_amdgcn_ballot_w64(gx >= 2 && gx <= w - 3);
    cv.c2b[p] = __builtin_amdgcn_ballot_w64(gx >= 2 && gx <= w - 2);
    cv.c4a[p] = __builtin_amdgcn_ballot_w64(gx >= 4 && gx <= w - 5);
    cv.c4b[p] = __builtin_amdgcn_ballot_w64(gx >= 4 && gx <= w - 4);
    cv.o7[p] = own && gx >= 7 && gx < w - 7;
  }
  const bool pair_in = gxe >= 0 && gxe < w;  // w even: a pair is inside or outside as a whole
  cv.img = __builtin_amdgcn_ballot_w64(pair_in);
  // every column of the window inside every step's column range (and clear of the stale p/q slots): workgroup-uniform
  const bool inner_cols = gx0 >= 4 && gx0 + 127 <= w - 5;

  // slide slots of this thread
  int sd[3], sdist[3];
#pragma unroll
  for (int k = 0; k < 3; k++) slide_slot(tid + k * NT, sd[k], sdist[k]);
  static_assert(SLIDE_SLOTS >= 2 * NT && (SLIDE_SLOTS - 2 * NT) % 64 == 0, "slots 0 and 1 exist for every thread, slot 2 for whole waves");
  const bool third = wv < (SLIDE_SLOTS - 2 * NT) / 64;  // wave-uniform

  // samples of window row 8 b + wv, columns 2 l, 2 l + 1; outside the frame: 0
  Pair<TI> st;
  bool st_in = false;
  auto prefetch = [&](int b) {
    const int gy = gy0 + RB * b + wv;
    st_in = pair_in && gy >= 0 && gy < h;
    if (st_in) st.fetch(in + (size_t)gy * w + gxe);
  };
  prefetch(0);
  bool ok1 = false, ok2 = false;  // range verdicts of the two previous blocks

  for (int b = 0; b < nsteps; b++) {
    // ---- slide: every plane moves up by 8 rows (its live rows; the rest is rewritten in this step)
    if (b > 0) {
      const float4 s0 = *slot16(lds + sd[0] + sdist[0]);
      const float4 s1 = *slot16(lds + sd[1] + sdist[1]);
      float4 s2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (third) s2 = *slot16(lds + sd[2] + sdist[2]);
      wg_barrier();
      *slot16(lds + sd[0]) = s0;
      *slot16(lds + sd[1]) = s1;
      if (third) *slot16(lds + sd[2]) = s2;
    }
    // ---- the new CFA rows (max(0, in)), their range verdict, and the next block's samples on their way
    {
      float a0 = 0.0f, a1 = 0.0f;
      if (st_in) {
        const float2 s2 = st.get();
        a0 = fmaxf(0.0f, s2.x);
        a1 = fmaxf(0.0f, s2.y);
      }
      b128[f128<CFA_B, CFA_L, CFA_W>(0, 0, 0, 0)] = a0;
      b128[f128<CFA_B, CFA_L, CFA_W>(0, 0, 1, 0)] = a1;
      if constexpr (!AP) {
        Range rg;
        rg.add(a0);
        rg.add(a1);
        const bool wave_ok = __builtin_amdgcn_ballot_w64(!rg.ok()) == 0;
        if (l == 0) verdict[(b & 3) * 8 + wv] = wave_ok ? 1u : 0u;
      }
      if (b + 1 < nsteps) prefetch(b + 1);
    }
    wg_barrier();
    bool fast = true;
    if constexpr (!AP) {
      uint32_t all = 1u;
#pragma unroll
      for (int k = 0; k < 8; k++) all &= verdict[(b & 3) * 8 + k];
      const bool ok0 = __builtin_amdgcn_readfirstlane(all) != 0;
      fast = ok0 && ok1 && ok2;  // the 24 newest CFA rows >= the 21 rows any step of this block reads
      ok2 = ok1; ok1 = ok0;
    }
    // the rows of every step of this block (lags 1 .. 10 behind rows gy0 + 8 b .. + 7) inside every step's row range
    const bool inner = inner_cols && gy0 + RB * b - LAG_52 >= 4 && gy0 + RB * b + RB - 1 - LAG_21 <= h - 5;

    // frame rows of this wave's sites in the steps of this block, and their border rules
    const int gyb = gy0 + RB * b + wv;  // row of lag 0
    Rows rv;
    { const int gy = gyb - LAG_21; rv.r21 = gy >= 2 && gy <= h - 2; }
    { const int gy = gyb - LAG_11; rv.r11 = gy >= 3 && gy <= h - 4; rv.rimg41 = gy >= 0 && gy < h; }
    { const int gy = gyb - LAG_12; rv.r12 = gy >= 2 && gy <= h - 3; }
    { const int gy = gyb - LAG_31; rv.r31 = gy >= 4 && gy <= h - 5; }
    { const int gy = gyb - LAG_51; rv.r51 = gy >= 4 && gy <= h - 4; }
    const int orow = RB * b - LAG_52 + wv, gyo = gy0 + orow;
    rv.rout = orow >= HALO && orow < HALO + seg_rows && gyo >= 7 && gyo < h - 7;
    T* dst = out + ((size_t)gyo * w + gxe) * 3;

#if defined(TDK_EXPERIMENTS) && defined(TDK_RS_NO_PHASE_BARRIERS)  // timing experiment only (wrong results): what the four barriers between the steps cost
#define RS_PHASE_BARRIER() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local")
#else
#define RS_PHASE_BARRIER() wg_barrier()
#endif
#define RS_STEP(MODEV, PEV)                                                                                \
  do {                                                                                                     \
    step_2_1_1_1_4_1<MODEV, PEV, TI>(b128, b64, cv, rv, in, gxe + 1, gyb - LAG_41, w, h);                  \
    RS_PHASE_BARRIER();                                                                                    \
    step_1_2_4_2<MODEV, PEV>(b128, b64, cv, rv);                                                           \
    RS_PHASE_BARRIER();                                                                                    \
    step_3_1<MODEV, PEV>(b128, b64, cv, rv);                                                               \
    RS_PHASE_BARRIER();                                                                                    \
    step_5_1<MODEV, PEV>(b128, b64, lanes_51, cv, rv);                                                     \
    RS_PHASE_BARRIER();                                                                                    \
    step_5_2_out<MODEV, PEV, T>(b128, b64, red_row, dst, cv, rv);                                          \
  } while (0)
    if constexpr (AP) {
      if (inner) { if (pe) RS_STEP(AINNER, 1); else RS_STEP(AINNER, 0); }
      else { if (pe) RS_STEP(ABORD, 1); else RS_STEP(ABORD, 0); }
    } else {
      if (fast && inner) { if (pe) RS_STEP(INNER, 1); else RS_STEP(INNER, 0); }
      else if (fast) { if (pe) RS_STEP(FASTM, 1); else RS_STEP(FASTM, 0); }
      else { if (pe) RS_STEP(SLOW, 1); else RS_STEP(SLOW, 0); }
    }
#undef RS_STEP
#undef RS_PHASE_BARRIER
  }
}

}  // namespace rs
