// laplacian.hip -- local Laplacian filter (6 gamma levels, fp16 pyramid storage, fp32 math).
//
// Replaces reference csrc/local_contrast/laplacian.cu:50-635 (LaplacianImpl<6>::process:
// pad_input_half, gauss_reduce_half, process_curve_half, laplacian_assemble<6>,
// write_back_half; ~90 launches on a 2.5x padded image).  Semantics kept: level sizes
// dl(x, l) = (x + 2^l - 1) >> l, levels = min(30, floor(log2(min(W, H)))), replicate padding
// by 2^(levels-1), every stored value rounded to binary16, 5x5 binomial reduce at 2c with
// c clamped to [1, size-2], parity-dependent expand taps x4, clamp_boundary, curve
// (laplacian.cu:266-290), the coarsest input level stored in the output pyramid (:526).
//
// MI355X design -- 10 launches at 12 MP (the reference: ~90; round 1 of this library: 32):
//  * level 0 exists nowhere in memory.  The padded input is clamp-indexed straight from the fp32
//    image (rounded to binary16 on the fly, exactly what pad_input_half stores), and level 0 of the six
//    gamma pyramids is binary16(curve(input)), evaluated where it is consumed;
//  * level1_kernel: level 1 of all seven pyramids (input + six gammas) in one pass over the image:
//    curves into LDS, separable reduce.  60 % of the 2.5x padded plane is pure padding, where the
//    replicated input depends on one coordinate only (or none): tiles that lie inside a padding band
//    evaluate one row / column / sample of curves and run the same reduce arithmetic on it;
//  * reduce_pair_kernel: two pyramid levels per launch -- a workgroup rebuilds the level-(l+1) window its
//    level-(l+2) tile reduces from (in LDS, binary16-rounded like the stored level), and stores its own part
//    of level l+1;
//  * deep_reduce_kernel / deep_assemble_kernel: once a level has <= 32 K pixels, all remaining levels run
//    inside ONE workgroup per pyramid (reduce) / one workgroup (assemble) with workgroup barriers between
//    the levels instead of a launch per level;
//  * assemble is tiled: the <= 35 x 11 coarse cells a 64 x 16 fine tile expands from are staged in
//    LDS for the output pyramid and all six gamma pyramids (the reference gathers 3 x 4..9 halves
//    per pixel from global memory); the level-0 assemble writes the fp32 result directly (write_back);
//    every level is assembled only on the rectangle the level below reads;
//  * pointers are computed from one layout struct passed by value (the reference uploads pointer tables
//    to process-global __device__ symbols before every level, laplacian.cu:43-45,574-575: not stream- or
//    multi-instance safe); workspace handed in by the caller (no allocation inside process()).
// Every value that the reference stores is still rounded to binary16 at the same point.
#include "tdk_common.h"

namespace {

constexpr int NG = 6;
constexpr int NP = NG + 1;  // pyramids that are reduced: the input's and the six gammas'
constexpr int MAX_LEVELS = 30;
constexpr int DEEP_PIXELS = 32768;  // levels at most this large are reduced inside single workgroups (one per pyramid)
// ... and levels at most this large assembled by ONE workgroup: the assemble has a single output pyramid, so its single-workgroup
// form pays for every pixel with one 1024-thread workgroup's latency -- level 5 of a 12 MP frame (13 K pixels to assemble) took
// 30 of that launch's 47 us and takes 11 as a tiled launch of its own
constexpr int DEEP_ASSEMBLE_PIXELS = 8192;

inline int dl(int x, int level) { return (x + (1 << level) - 1) >> level; }

__device__ __forceinline__ float hld(const __half* p, int x, int y, int w) { return __half2float(p[(size_t)y * w + x]); }
__device__ __forceinline__ void hst(__half* p, int x, int y, int w, float v) { p[(size_t)y * w + x] = __float2half_rn(v); }
__device__ __forceinline__ float round_half(float v) { return __half2float(__float2half_rn(v)); }

// Workspace layout: (2 + NG) pyramids of `pyr_elems` halves each -- 0: padded input, 1: output, 2 + k: gamma k;
// level l of a pyramid starts `off[l]` elements in.  Passed to every kernel by value.
struct Layout {
  __half* base;
  size_t pyr_elems;
  size_t off[MAX_LEVELS + 1];
  int levels, pad, bw, bh, w, h;
  __host__ __device__ int lw(int l) const { return (bw + (1 << l) - 1) >> l; }
  __host__ __device__ int lh(int l) const { return (bh + (1 << l) - 1) >> l; }
  __host__ __device__ __half* at(int pyramid, int l) const { return base + (size_t)pyramid * pyr_elems + off[l]; }
  // reduce chains: z = 0 is the input's pyramid, whose coarsest level lives in the output pyramid
  // (laplacian.cu:526); z = 1..6 the gamma pyramids
  __host__ __device__ __half* chain(int z, int l) const { return z == 0 ? at(l == levels - 1 ? 1 : 0, l) : at(1 + z, l); }
};

// binary16(padded input) at padded coordinates (pad_input_half, laplacian.cu:70-90)
template <typename T> __device__ __forceinline__ float padded0(const T* __restrict__ in, const Layout& L, int x, int y) {
  const int cx = min(max(x - L.pad, 0), L.w - 1), cy = min(max(y - L.pad, 0), L.h - 1);
  return round_half(ld<T>(in, (size_t)cy * L.w + cx));  // binary16 storage: the value itself
}

__device__ __forceinline__ int clampc(int p, int n) {  // the reduce's centre clamp (laplacian.cu:185-190)
  int c = p;
  if (p >= n - 1) c = n - 2;
  if (c <= 0) c = 1;
  return c;
}

// laplacian.cu:177-207: 5 x 5 binomial at (2 cx, 2 cy) of the finer level
template <typename F> __device__ __forceinline__ float reduce25(F fine, int px, int py, int cw, int ch) {
  const int cx = clampc(px, cw), cy = clampc(py, ch);
  const float w5[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
  float acc = 0.0f;
#pragma unroll
  for (int j = -2; j <= 2; j++)
#pragma unroll
    for (int i = -2; i <= 2; i++) acc += fine(2 * cx + i, 2 * cy + j) * w5[i + 2] * w5[j + 2];
  return acc;
}

// The same reduce when the finer level is binary16 data in memory: v_dot2c_f32_f16 multiplies two binary16 pairs
// exactly and accumulates in fp32 -- the stored halves feed the multiplier directly (no conversions) and a row
// of five taps is 2 dot instructions + 1 FMA.  The weights w_i w_j (k / 256, k in {1, 4, 6, 16, 24, 36}) are exact in
// binary16, every product is exact, and the 25 terms are summed in fp32 row by row: the result differs from the
// reference's left-to-right sum of twice-rounded products by a few fp32 ulps, far below the binary16 rounding it
// goes through next.
typedef _Float16 hpair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot5(const __half* __restrict__ p, float wj, float acc) {  // p[0..4] . (1 4 6 4 1) / 16 * wj
  uint32_t d01, d23;
  __builtin_memcpy(&d01, p, 4);
  __builtin_memcpy(&d23, p + 2, 4);
  const hpair w01 = {(_Float16)(wj * (1.0f / 16.0f)), (_Float16)(wj * (4.0f / 16.0f))};
  const hpair w23 = {(_Float16)(wj * (6.0f / 16.0f)), (_Float16)(wj * (4.0f / 16.0f))};
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(hpair, d01), w01, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(hpair, d23), w23, acc, false);
  return __builtin_fmaf(__half2float(p[4]), wj * (1.0f / 16.0f), acc);
}
__device__ __forceinline__ float reduce25_half(const __half* __restrict__ fine, int fw, int px, int py, int cw, int ch) {
  const int cx = clampc(px, cw), cy = clampc(py, ch);
  const float w5[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
  const __half* p = fine + (size_t)(2 * cy - 2) * fw + (2 * cx - 2);
  float acc = 0.0f;
#pragma unroll
  for (int j = 0; j < 5; j++) acc = dot5(p + (size_t)j * fw, w5[j], acc);
  return acc;
}

// laplacian.cu:266-290 in the oracle's operation order (oracle/src/laplacian.c curve()), no FMA contraction, the quotient
// c / (2 ssigma) correctly rounded: q = c rc, r = c - q d exactly (fma), q + r rc (Markstein) with rc = +-1 / (2 sigma) rounded
// on the host -- exact for every finite c this kernel can see (|c| <= 2 sigma on this branch and >= 2^-26 or 0: x is a
// binary16 value, g a multiple of 1/12) as long as sigma itself is ordinary (checked per launch: CurveK::plain_div otherwise).
// Round 3 evaluated a factored form under contraction (12 instead of ~26 instructions): a fraction of a percent of the pixels
// then saw a binary16 rounding flip somewhere in their pyramid.  With this form the whole filter is bit-identical to the oracle
// on every size / parameter set of the tests and of profiles/bounds_probe.py (level-1 kernel 128 -> 169 us, level-0 assemble
// 93 -> 103 us at 12 MP: profiles/r04/experiments/laplacian_faithful_curve.txt) -- parity first.  The clarity term keeps the
// hardware exp2 (the reference is a --use_fast_math build); per-launch constants are hoisted into CurveK.
struct CurveK {
  float sigma, two_sigma, inv_two_sigma, shadows, highlights, clarity, neg_inv_e;  // neg_inv_e = -log2(e) / (2 sigma^2 / 3)
  bool plain_div;
};
__device__ __forceinline__ CurveK make_curve(float sigma, float shadows, float highlights, float clarity) {
  CurveK k;
  k.sigma = sigma; k.two_sigma = 2 * sigma; k.inv_two_sigma = 1.0f / (2.0f * sigma);
  k.shadows = shadows; k.highlights = highlights; k.clarity = clarity;
  k.neg_inv_e = -1.44269504088896341f / (2.0f * sigma * sigma / 3.0f);
  k.plain_div = !(sigma >= 0x1p-20f && sigma <= 0x1p20f);
  return k;
}
template <bool HAS_CLARITY> __device__ __forceinline__ float curve(float x, float g, const CurveK& k) {
  const float c = x - g;
  // c > 0 ? sigma : -sigma; at c == 0 either sign gives val = g (t = 0 annihilates both terms), so the sign bit of c serves
  const float ssigma = copysignf(k.sigma, c);
  const float shadhi = c > 0.0f ? k.shadows : k.highlights;
  const float lin = g + ssigma + shadhi * (c - ssigma);  // linear part
#ifndef TDK_LAP_NO_SHORTCUT
  // Every lane of the wave beyond 2 sigma of this gamma centre (neighbouring samples are: with sigma = 0.2 a third of the (sample, gamma)
  // pairs of an image): the bezier branch below would be computed and discarded by the select -- skip it, the result is the same `lin`.
  if (__builtin_amdgcn_ballot_w64(!(fabsf(c) > k.two_sigma)) == 0) {
    float v = lin;
    if constexpr (HAS_CLARITY) v += k.clarity * c * __builtin_amdgcn_exp2f(c * c * k.neg_inv_e);
    return v;
  }
#endif
  // blend in via quadratic bezier
  const float d = 2.0f * ssigma;
  float q;
  if (__builtin_expect(k.plain_div, 0)) {
    q = c / d;
  } else {
    const float rc = copysignf(k.inv_two_sigma, c);
    const float q0 = c * rc;
    q = __builtin_fmaf(__builtin_fmaf(-q0, d, c), rc, q0);
  }
  const float t = fminf(fmaxf(q, 0.0f), 1.0f);
  const float t2 = t * t;
  const float mt = 1.0f - t;
  const float bez = g + d * mt * t + t2 * (ssigma + ssigma * shadhi);  // ssigma * 2.0f == d
  float val = (fabsf(c) > k.two_sigma) ? lin : bez;
  if constexpr (HAS_CLARITY) val += k.clarity * c * __builtin_amdgcn_exp2f(c * c * k.neg_inv_e);  // hardware exp2 (fast-math build)
  return val;
}
__device__ __forceinline__ float gamma_centre(int k) { return ((float)k + 0.5f) / (float)NG; }

// ---------------------------------------------------------------- level 0 -> level 1, all seven pyramids
// A 512-thread workgroup owns 32 x 16 level-1 cells.  Phase 1: the (2*32+3) x (2*16+3) level-0 window, input and
// six curves, rounded to binary16 into LDS.  Phase 2: horizontal 5-tap sums at the tile's coarse columns, then their
// vertical combination and store (separable: the summation order differs from the 2-D form -- far below the
// binary16 rounding of the result).  A window that lies inside a padding band holds one distinct column (left / right
// band), one distinct row (top / bottom) or one sample (corners): only those are evaluated, and phase 2 reads them
// through the same index map, i.e. runs the same arithmetic on the same values as the full window would.
constexpr int A_TW = 32, A_TH = 16, A_FW = 2 * A_TW + 3, A_FH = 2 * A_TH + 3, A_FS = A_FW + 1, A_NT = 512;

template <bool HAS_CLARITY, typename T>
__global__ __launch_bounds__(A_NT) void level1_kernel(const T* __restrict__ in, Layout L, float sigma, float shadows, float highlights, float clarity) {
  __shared__ __half fine[NP][A_FH * A_FS];
  const CurveK ck = make_curve(sigma, shadows, highlights, clarity);
  const int cw = L.lw(1), ch = L.lh(1);
  const int CX0 = blockIdx.x * A_TW, CY0 = blockIdx.y * A_TH;
  const int cxa = clampc(CX0, cw), cxb = clampc(min(CX0 + A_TW, cw) - 1, cw);
  const int cya = clampc(CY0, ch), cyb = clampc(min(CY0 + A_TH, ch) - 1, ch);
  const int fx0 = 2 * cxa - 2, fy0 = 2 * cya - 2;
  const int fx1 = 2 * cxb + 2, fy1 = 2 * cyb + 2;  // last fine column / row any tap of this tile reads
  // distinct source columns / rows of the window (padded coordinates clamp into the image)
  const bool one_col = (fx1 - L.pad <= 0) || (fx0 - L.pad >= L.w - 1);
  const bool one_row = (fy1 - L.pad <= 0) || (fy0 - L.pad >= L.h - 1);
  const int ncol = one_col ? 1 : A_FW, nrow = one_row ? 1 : A_FH;
  for (int i = threadIdx.x; i < ncol * nrow; i += A_NT) {
    const int r = one_col ? i : (one_row ? 0 : i / A_FW), c = one_col ? 0 : i - r * A_FW;
    const float v = padded0(in, L, min(fx0 + c, fx1), min(fy0 + r, fy1));
    fine[0][r * A_FS + c] = __float2half_rn(v);
#pragma unroll
    for (int k = 0; k < NG; k++) fine[1 + k][r * A_FS + c] = __float2half_rn(curve<HAS_CLARITY>(v, gamma_centre(k), ck));
  }
  __syncthreads();
  // Phase 2: thread = (coarse column, pyramid, upper / lower 8 rows of the tile).  It forms the horizontal 5-tap sums of
  // the 19 window rows behind its 8 outputs once, in registers, and combines them vertically -- no second LDS array
  // (the workgroup's LDS is the 33 KB window: four workgroups per CU) and no second barrier.
  const float w5[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
  const int item = threadIdx.x;
  if (item >= A_TW * 2 * NP) return;
  const int pxl = item & (A_TW - 1), hk = item / A_TW, half = hk & 1, k = hk >> 1;
  const int px = CX0 + pxl;
  if (px >= cw) return;
  const int c0 = one_col ? 0 : 2 * clampc(px, cw) - fx0;  // even; the rows start 4-B aligned: two aligned dwords + one half per row
  auto hsum = [&](int r) {  // horizontal sum of window row r at this column
    const __half* row = &fine[k][(one_row ? 0 : r) * A_FS];
    if (one_col) {  // five equal taps, same instructions as the general case
      const __half same[6] = {row[0], row[0], row[0], row[0], row[0], row[0]};
      return dot5(same, 1.0f, 0.0f);
    }
    return dot5(row + c0 - 2, 1.0f, 0.0f);
  };
  __half* dst = L.chain(k, 1);
  const int PY0 = CY0 + 8 * half;
  if (CY0 >= 1 && PY0 + 7 <= ch - 2) {  // no centre clamp in this half: output o reads window rows 16 half + 2 o .. + 4
    float hv[19];
    if (one_row) {
      const float v = hsum(0);
#pragma unroll
      for (int i = 0; i < 19; i++) hv[i] = v;
    } else {
#pragma unroll
      for (int i = 0; i < 19; i++) hv[i] = hsum(16 * half + i);
    }
#pragma unroll
    for (int o = 0; o < 8; o++) {
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < 5; j++) acc += hv[2 * o + j] * w5[j];
      hst(dst, px, PY0 + o, cw, acc);
    }
  } else {  // first / last tile rows of the level
    for (int o = 0; o < 8; o++) {
      const int py = PY0 + o;
      if (py >= ch) break;
      const int ly = 2 * clampc(py, ch) - fy0;
      float acc = 0.0f;
#pragma unroll
      for (int j = -2; j <= 2; j++) acc += hsum(ly + j) * w5[j + 2];
      hst(dst, px, py, cw, acc);
    }
  }
}

// ---------------------------------------------------------------- one level: l -> l + 1 (blockIdx.z = pyramid)
__global__ __launch_bounds__(256) void reduce_kernel(Layout L, int l) {
  const int cw = L.lw(l + 1), ch = L.lh(l + 1), fw = L.lw(l);
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cw || y >= ch) return;
  const __half* fine = L.chain(blockIdx.z, l);
  hst(L.chain(blockIdx.z, l + 1), x, y, cw, reduce25_half(fine, fw, x, y, cw, ch));
}

// ---------------------------------------------------------------- two levels: l -> l + 1 -> l + 2
// A 256-thread workgroup owns 16 x 16 cells of level l + 2.  It stages the <= 79 x 79 cells of level l behind them in
// LDS (8-B loads; per-tap global loads made this kernel texture-addresser bound: 338 vector-memory instructions per
// workgroup against 25 now), computes the 38 x 38 window of level l + 1 they (and its edge cases) reduce from,
// rounded to binary16 like the stored level, stores the 32 x 32 cells of level l + 1 it owns, and reduces the window to
// its level-(l+2) cells.  Window cell (wr, wc) holds level l + 1 at (2 Y0 - 4 + wr, 2 X0 - 4 + wc) clamped into the level:
// every tap the centre clamp can ask for is inside.
constexpr int P_T = 16, P_WIN = 2 * P_T + 6, P_WS = P_WIN + 1, P_FW = 2 * P_WIN + 3, P_FQ = (P_FW + 4) / 4;  // 38, 39, 79, 20 quads per row
// row pitch of the staged fine cells in halves.  (With this natural pitch lanes 38 .. 63 of a wave collide two-way with lanes
// 0 .. 37 -- 42 % of the kernel's LDS cycles are bank conflicts; a pitch of 104 removes them and changes nothing, 72.6 against
// 69.4 us: the kernel waits on its three dependent phases, not on the LDS.  profiles/r04/experiments/laplacian_small_levels.txt)
constexpr int P_FS = 80;
static_assert(P_FS >= 4 * P_FQ && P_FS % 4 == 0, "8-byte staging stores");

__global__ __launch_bounds__(256) void reduce_pair_kernel(Layout L, int l) {
  __shared__ __align__(8) __half fwin[P_FW * P_FS];
  __shared__ float win[P_WIN * P_WS];
  const int z = blockIdx.z;
  const int fw = L.lw(l);
  const int w1 = L.lw(l + 1), h1 = L.lh(l + 1), w2 = L.lw(l + 2), h2 = L.lh(l + 2);
  const __half* fine = L.chain(z, l);
  __half* mid = L.chain(z, l + 1);
  const int X0 = blockIdx.x * P_T, Y0 = blockIdx.y * P_T;
  const int ox = 2 * X0 - 4, oy = 2 * Y0 - 4;
  auto into = [](int v, int n) { return min(max(v, 0), n - 1); };
  // level-l cells any tap of the window reads
  const int fx0 = 2 * clampc(into(ox, w1), w1) - 2, fx1 = 2 * clampc(into(ox + P_WIN - 1, w1), w1) + 2;
  const int fy0 = 2 * clampc(into(oy, h1), h1) - 2, fy1 = 2 * clampc(into(oy + P_WIN - 1, h1), h1) + 2;
  if (fx1 - fx0 == P_FW - 1 && fy1 - fy0 == P_FW - 1) {  // full window: 4 cells per load (the 80th column is read but never used)
    // all of a thread's loads in flight before the first one is stored (a load -> store loop pays one memory round trip per
    // round, seven in a row: 69 -> 63 us.  The same change in the assemble kernels' staging and per-pixel loads made THEM 12 %
    // slower -- more registers, and their waves already overlap each other's round trips; profiles/r04/experiments/laplacian_small_levels.txt)
    constexpr int NL = (P_FW * P_FQ + 255) / 256;
    uint2 u[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const int i = threadIdx.x + 256 * k, r = i / P_FQ, q = i - r * P_FQ;
      if (i < P_FW * P_FQ) __builtin_memcpy(&u[k], fine + (size_t)(fy0 + r) * fw + fx0 + 4 * q, 8);
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const int i = threadIdx.x + 256 * k, r = i / P_FQ, q = i - r * P_FQ;
      if (i < P_FW * P_FQ) *reinterpret_cast<uint2*>(&fwin[r * P_FS + 4 * q]) = u[k];
    }
  } else {
    for (int i = threadIdx.x; i < P_FW * P_FW; i += 256) {
      const int r = i / P_FW, c = i - r * P_FW;
      fwin[r * P_FS + c] = fine[(size_t)min(fy0 + r, fy1) * fw + min(fx0 + c, fx1)];
    }
  }
  __syncthreads();
  const float w5[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
  for (int i = threadIdx.x; i < P_WIN * P_WIN; i += 256) {
    const int wr = i / P_WIN, wc = i - wr * P_WIN;
    const int x1 = into(ox + wc, w1), y1 = into(oy + wr, h1);
    const __half* p = &fwin[(2 * clampc(y1, h1) - 2 - fy0) * P_FS + (2 * clampc(x1, w1) - 2 - fx0)];  // even column: 4-B aligned
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 5; j++) acc = dot5(p + j * P_FS, w5[j], acc);
    const float v = round_half(acc);
    win[wr * P_WS + wc] = v;
    const bool own = wc >= 4 && wc < 4 + 2 * P_T && wr >= 4 && wr < 4 + 2 * P_T && ox + wc < w1 && oy + wr < h1;
    if (own) mid[(size_t)y1 * w1 + x1] = __float2half_rn(v);  // v is a binary16 value: exact
  }
  __syncthreads();
  const int x2 = X0 + (threadIdx.x & (P_T - 1)), y2 = Y0 + threadIdx.x / P_T;
  if (x2 >= w2 || y2 >= h2) return;
  hst(L.chain(z, l + 2), x2, y2, w2, reduce25([&](int x1, int y1) { return win[(y1 - oy) * P_WS + (x1 - ox)]; }, x2, y2, w2, h2));
}

// ---------------------------------------------------------------- the small levels of one pyramid in one workgroup
// Levels first + 1 .. levels - 1 from level `first`; blockIdx.x = pyramid.  A level is written to global memory and
// read back by the same workgroup after a barrier (workgroup-scope release / acquire through __syncthreads).
__global__ __launch_bounds__(1024) void deep_reduce_kernel(Layout L, int first) {
  const int z = blockIdx.x;
  for (int l = first; l + 1 < L.levels; l++) {
    const int cw = L.lw(l + 1), ch = L.lh(l + 1), fw = L.lw(l);
    const __half* fine = L.chain(z, l);
    __half* coarse = L.chain(z, l + 1);
    for (int i = threadIdx.x; i < cw * ch; i += 1024) {
      const int y = i / cw, x = i - y * cw;
      hst(coarse, x, y, cw, reduce25_half(fine, fw, x, y, cw, ch));
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---------------------------------------------------------------- assemble
// laplacian.cu:53-65
__host__ __device__ __forceinline__ int clamp_boundary(int q, int n) {
  if (n & 1) { if (q > n - 2) q = n - 2; } else { if (q > n - 3) q = n - 3; }
  if (q <= 0) q = 1;
  return q;
}

// laplacian.cu:111-141 on any coarse-level accessor: an even coordinate uses taps -1, 0, 1 with weights 1 6 1, an
// odd one taps 0, 1 with weights 4 4.  Same taps, weights and summation order as the reference's loops; the tap
// an odd coordinate skips gets weight 0 (adds an exact +0) instead of a branch.
template <typename F> __device__ __forceinline__ float expand4(F coarse, int x, int y) {
  const int cx = x / 2, cy = y / 2;
  const bool x_odd = x & 1, y_odd = y & 1;
  const float wx[3] = {x_odd ? 0.0f : 1.0f / 16.0f, x_odd ? 4.0f / 16.0f : 6.0f / 16.0f, x_odd ? 4.0f / 16.0f : 1.0f / 16.0f};
  const float wy[3] = {y_odd ? 0.0f : 1.0f / 16.0f, y_odd ? 4.0f / 16.0f : 6.0f / 16.0f, y_odd ? 4.0f / 16.0f : 1.0f / 16.0f};
  float c = 0.0f;
#pragma unroll
  for (int i = -1; i <= 1; i++)
#pragma unroll
    for (int j = -1; j <= 1; j++) c += coarse(cx + i, cy + j) * wx[i + 1] * wy[j + 1];
  return 4.0f * c;
}

// One pixel of laplacian_assemble<6> (laplacian.cu:221-252).  `v` = the level's input value, `out_c(i, j)` / `gam_c(k,
// i, j)` = the coarser level of the output / gamma-k pyramid, `gam_f(k)` = gamma k's value at this pixel.
template <typename OC, typename GC, typename GF>
__device__ __forceinline__ float assemble_px(float v, int qx, int qy, OC out_c, GC gam_c, GF gam_f) {
  float val = expand4(out_c, qx, qy);
  // the reference's search loop over the (increasing) gamma centres, as four compares
  int hi = 1;
#pragma unroll
  for (int q = 1; q < NG - 1; q++) hi += (gamma_centre(q) <= v) ? 1 : 0;
  const int lo = hi - 1;
  const float a = fminf(fmaxf(v * NG - ((float)lo + .5f), 0.0f), 1.0f);
  const float l0 = gam_f(lo) - expand4([&](int i, int j) { return gam_c(lo, i, j); }, qx, qy);
  const float l1 = gam_f(lo + 1) - expand4([&](int i, int j) { return gam_c(lo + 1, i, j); }, qx, qy);
  return val + (l0 * (1.0f - a) + l1 * a);
}

// Tiled: a 256-thread workgroup owns 64 x 16 fine pixels; the coarse cells their expands touch (<= 35 x 11) of the
// output pyramid and of all six gamma pyramids are staged once in LDS as floats.  LEVEL0: the fine level is the
// image itself -- input = binary16(image), gamma k = binary16(curve(input)), and the result goes out as fp32
// (write_back_half, laplacian.cu:92-108) for the pixels inside the image.
constexpr int ATW = 64, ATH = 16, ACW = ATW / 2 + 3, ACH = ATH / 2 + 3, ACS = ACW + 1;

template <bool LEVEL0, bool HAS_CLARITY, typename T>
__global__ __launch_bounds__(256) void assemble_tiled_kernel(Layout L, int l, const T* __restrict__ image, T* __restrict__ result, float sigma,
                                                             float shadows, float highlights, float clarity, int tile_x0, int tile_y0) {
  __shared__ float tiles[NP * ACH * ACS];
  const CurveK ck = make_curve(sigma, shadows, highlights, clarity);
  const int fw = L.lw(l), fh = L.lh(l);
  const int X0 = (blockIdx.x + tile_x0) * ATW, Y0 = (blockIdx.y + tile_y0) * ATH;
  const int cw = (fw - 1) / 2 + 1, chh = (fh - 1) / 2 + 1;
  const int X1 = min(X0 + ATW, fw) - 1, Y1 = min(Y0 + ATH, fh) - 1;
  // coarse cells touched: clamp_boundary is monotone, taps are cx - 1 .. cx + 1
  const int tx0 = max(clamp_boundary(X0, fw) / 2 - 1, 0), tx1 = min(clamp_boundary(X1, fw) / 2 + 1, cw - 1);
  const int ty0 = max(clamp_boundary(Y0, fh) / 2 - 1, 0), ty1 = min(clamp_boundary(Y1, fh) / 2 + 1, chh - 1);
  // the full ACW x ACH window is loaded (coordinates clamped into the level): constant trip counts and divisors
#pragma unroll
  for (int k = 0; k < NP; k++) {
    const __half* src = L.at(1 + k, l + 1);  // output pyramid, then the gamma pyramids
    for (int i = threadIdx.x; i < ACW * ACH; i += 256) {
      const int r = i / ACW, c = i - r * ACW;
      tiles[k * (ACH * ACS) + r * ACS + c] = hld(src, min(tx0 + c, tx1), min(ty0 + r, ty1), cw);
    }
  }
  __syncthreads();
  const int x = X0 + (threadIdx.x & 63);
  if (x >= fw) return;
  if (LEVEL0 && (x < L.pad || x >= L.pad + L.w)) return;
  const int qx = clamp_boundary(x, fw);
  for (int yy = threadIdx.x >> 6; yy < ATH; yy += 4) {
    const int y = Y0 + yy;
    if (y >= fh) break;
    if (LEVEL0 && (y < L.pad || y >= L.pad + L.h)) continue;
    const int qy = clamp_boundary(y, fh);
    // index clamps only matter for taps of weight 0
    auto tile = [&](int k, int i, int j) { return tiles[k * (ACH * ACS) + max(j - ty0, 0) * ACS + max(i - tx0, 0)]; };
    float v;
    if constexpr (LEVEL0) v = round_half(ld<T>(image, (size_t)(y - L.pad) * L.w + (x - L.pad)));
    else v = hld(L.at(0, l), x, y, fw);
    const float val = assemble_px(
        v, qx, qy, [&](int i, int j) { return tile(0, i, j); }, [&](int k, int i, int j) { return tile(1 + k, i, j); },
        [&](int k) {
          if constexpr (LEVEL0) return round_half(curve<HAS_CLARITY>(v, gamma_centre(k), ck));
          else return hld(L.at(2 + k, l), x, y, fw);
        });
    if constexpr (LEVEL0) st<T>(result, (size_t)(y - L.pad) * L.w + (x - L.pad), round_half(val));
    else hst(L.at(1, l), x, y, fw, val);
  }
}

// The small levels `top` .. `bottom` (descending) of the output pyramid in ONE workgroup, each on its needed
// rectangle.  The coarser level of all seven pyramids is first staged in LDS (binary16, <= 7 x 16 K cells): 27 taps per
// pixel straight from global memory made the kernel texture-addresser bound on its single CU.
struct Rect { int x0, x1, y0, y1; };  // inclusive
struct Rects { Rect r[MAX_LEVELS + 1]; };

__global__ __launch_bounds__(1024) void deep_assemble_kernel(Layout L, Rects need, int top, int bottom) {
  extern __shared__ __half coarse[];  // [NP][cw * chh]
  for (int l = top; l >= bottom; l--) {
    const int fw = L.lw(l), fh = L.lh(l), cw = L.lw(l + 1), chh = L.lh(l + 1), cells = cw * chh;
    for (int k = 0; k < NP; k++) {
      const __half* src = L.at(1 + k, l + 1);  // output pyramid (written by the previous iteration / the reduce chain), then the gammas
      for (int i = threadIdx.x; i < cells; i += 1024) coarse[k * cells + i] = src[i];
    }
    __syncthreads();
    const Rect rc = need.r[l];
    const int rw = rc.x1 - rc.x0 + 1, rh = rc.y1 - rc.y0 + 1;
    for (int i = threadIdx.x; i < rw * rh; i += 1024) {
      const int yy = i / rw, x = rc.x0 + (i - yy * rw), y = rc.y0 + yy;
      const float v = hld(L.at(0, l), x, y, fw);
      auto cell = [&](int k, int ci, int cj) { return __half2float(coarse[k * cells + max(cj, 0) * cw + max(ci, 0)]); };  // -1 only under weight 0
      const float val = assemble_px(
          v, clamp_boundary(x, fw), clamp_boundary(y, fh), [&](int ci, int cj) { return cell(0, ci, cj); },
          [&](int k, int ci, int cj) { return cell(1 + k, ci, cj); }, [&](int k) { return hld(L.at(2 + k, l), x, y, fw); });
      hst(L.at(1, l), x, y, fw, val);
    }
    __threadfence_block();
    __syncthreads();
  }
}

Layout make_layout(int w, int h) {
  Layout L;
  const int m = w < h ? w : h;
  int lg = 0;
  while ((1 << (lg + 1)) <= m) lg++;
  L.levels = lg < MAX_LEVELS ? lg : MAX_LEVELS;
  L.pad = L.levels >= 1 ? 1 << (L.levels - 1) : 0;
  L.w = w;
  L.h = h;
  L.bw = w + 2 * L.pad;
  L.bh = h + 2 * L.pad;
  L.base = nullptr;
  size_t off = 0;
  for (int l = 0; l <= MAX_LEVELS; l++) {
    L.off[l] = off;
    // level 0 of every pyramid is virtual (see the header): it takes no space
    if (l >= 1 && l < L.levels) off += tdk_align_up((size_t)dl(L.bw, l) * dl(L.bh, l), 128);
  }
  L.pyr_elems = off;
  return L;
}

}  // namespace

TDK_EXPORT size_t tdk_laplacian_workspace_bytes(int width, int height, int num_gamma) {
  if (width <= 0 || height <= 0 || num_gamma != NG) return 0;
  const Layout L = make_layout(width, height);
  return tdk_align_up((size_t)(2 + NG) * L.pyr_elems * sizeof(__half), 256);  // padded + output + 6 gamma pyramids, levels >= 1
}

namespace {
template <typename T>
int laplacian_t(const T* lum_in, T* lum_out, void* workspace, int width, int height, int num_gamma, float sigma, float shadows, float highlights,
                float clarity, tdk_stream_t stream) {
  TDK_REQUIRE(num_gamma == NG, "Unsupported gamma count: %d", num_gamma);
  TDK_REQUIRE(lum_in && lum_out && workspace, "tdk_laplacian: null pointer");
  TDK_REQUIRE(width >= 4 && height >= 4, "tdk_laplacian: image %dx%d too small", width, height);
  hipStream_t s = tdk_stream(stream);
  Layout L = make_layout(width, height);  // levels >= 2 for a 4 x 4 image
  L.base = reinterpret_cast<__half*>(workspace);
  const int top = L.levels - 1;             // coarsest level
  auto small = [&](int l) { return (int64_t)L.lw(l) * L.lh(l) <= DEEP_PIXELS; };

  // ---- reduce side: level 1 of the seven pyramids, then pairs of levels, then everything small in one launch
  const dim3 g1(tdk_div_up(L.lw(1), A_TW), tdk_div_up(L.lh(1), A_TH));
  if (clarity != 0.0f) TDK_LAUNCH("tdk_laplacian(level1)", (level1_kernel<true, T>), g1, dim3(A_NT), 0, s, lum_in, L, sigma, shadows, highlights, clarity);
  else TDK_LAUNCH("tdk_laplacian(level1)", (level1_kernel<false, T>), g1, dim3(A_NT), 0, s, lum_in, L, sigma, shadows, highlights, clarity);
  int l = 1;  // finest level that exists so far
  while (l < top && !small(l + 1)) {
    if (l + 2 <= top) {  // also when level l + 2 is already small: it costs less here than as the single-workgroup kernel's first level
      TDK_LAUNCH(l == 1 ? "tdk_laplacian(reduce pair 2,3)" : "tdk_laplacian(reduce pair 4+)", reduce_pair_kernel, dim3(tdk_div_up(L.lw(l + 2), P_T), tdk_div_up(L.lh(l + 2), P_T), NP), dim3(256), 0, s, L, l);
      l += 2;
    } else {
      TDK_LAUNCH("tdk_laplacian(reduce)", reduce_kernel, dim3(tdk_div_up(L.lw(l + 1), 64), tdk_div_up(L.lh(l + 1), 4), NP), dim3(256), 0, s, L, l);
      l += 1;
    }
  }
  if (l < top) TDK_LAUNCH("tdk_laplacian(deep reduce)", deep_reduce_kernel, dim3(NP), dim3(1024), 0, s, L, l);

  // ---- assemble side.  The output pyramid is only consumed downwards, and only the image rectangle of level 0 is
  // ever read.  So level l is assembled only on the rectangle the level below expands from (clamp_boundary, taps
  // cx - 1 .. cx + 1): most of the 2.5x padded area is never assembled at the fine levels.
  Rects need;
  need.r[0] = Rect{L.pad, L.pad + width - 1, L.pad, L.pad + height - 1};
  for (int k = 0; k + 1 < L.levels; k++) {
    const int fw = L.lw(k), fh = L.lh(k), cw = (fw - 1) / 2 + 1, chh = (fh - 1) / 2 + 1;
    auto lo = [](int v) { return v > 0 ? v : 0; };
    const Rect f = need.r[k];
    need.r[k + 1] = Rect{lo(clamp_boundary(f.x0, fw) / 2 - 1), (clamp_boundary(f.x1, fw) / 2 + 1 < cw - 1) ? clamp_boundary(f.x1, fw) / 2 + 1 : cw - 1,
                         lo(clamp_boundary(f.y0, fh) / 2 - 1), (clamp_boundary(f.y1, fh) / 2 + 1 < chh - 1) ? clamp_boundary(f.y1, fh) / 2 + 1 : chh - 1};
  }
  int a = top - 1;  // next level to assemble (the coarsest output level is the coarsest input level)
  auto small_a = [&](int l) { return (int64_t)L.lw(l) * L.lh(l) <= DEEP_ASSEMBLE_PIXELS; };
  if (a >= 1 && small_a(a)) {
    int bottom = a;
    auto fits = [&](int lv) { return (size_t)NP * L.lw(lv + 1) * L.lh(lv + 1) * sizeof(__half) <= 150 * 1024; };  // its coarser level in LDS, 7 pyramids
    while (bottom - 1 >= 1 && small_a(bottom - 1) && fits(bottom - 1)) bottom--;
    const size_t lds = (size_t)NP * L.lw(bottom + 1) * L.lh(bottom + 1) * sizeof(__half);  // the largest coarser level staged
    TDK_MAX_LDS_ONCE(deep_assemble_kernel, "tdk_laplacian(hipFuncSetAttribute)");
    TDK_LAUNCH("tdk_laplacian(deep assemble)", deep_assemble_kernel, dim3(1), dim3(1024), lds, s, L, need, a, bottom);
    a = bottom - 1;
  }
  for (; a >= 0; a--) {
    const Rect rc = need.r[a];
    const int tx0 = rc.x0 / ATW, tx1 = rc.x1 / ATW, ty0 = rc.y0 / ATH, ty1 = rc.y1 / ATH;
    const dim3 g(tx1 - tx0 + 1, ty1 - ty0 + 1);
    if (a == 0 && clarity != 0.0f)
      TDK_LAUNCH("tdk_laplacian(assemble 0)", (assemble_tiled_kernel<true, true, T>), g, dim3(256), 0, s, L, a, lum_in, lum_out, sigma, shadows, highlights, clarity, tx0, ty0);
    else if (a == 0)
      TDK_LAUNCH("tdk_laplacian(assemble 0)", (assemble_tiled_kernel<true, false, T>), g, dim3(256), 0, s, L, a, lum_in, lum_out, sigma, shadows, highlights, clarity, tx0, ty0);
    else
      TDK_LAUNCH(a == 1 ? "tdk_laplacian(assemble 1)" : a == 2 ? "tdk_laplacian(assemble 2)" : "tdk_laplacian(assemble 3+)", (assemble_tiled_kernel<false, false, T>), g,
                 dim3(256), 0, s, L, a, lum_in, lum_out, sigma, shadows, highlights, clarity, tx0, ty0);
  }
  return TDK_OK;
}
}  // namespace

/* dtype: storage type of lum_in and lum_out.  The reference pads the image to binary16 and writes binary16 values back as float32
 * (laplacian.cu:70-108), so a binary16 image in and out is the SAME result, not a rounded one. */
TDK_EXPORT int tdk_laplacian_ex(const void* lum_in, void* lum_out, void* workspace, int width, int height, int num_gamma, float sigma, float shadows,
                                float highlights, float clarity, int dtype, tdk_stream_t stream) {
  if (dtype == TDK_F16)
    return laplacian_t<__half>(reinterpret_cast<const __half*>(lum_in), reinterpret_cast<__half*>(lum_out), workspace, width, height, num_gamma, sigma, shadows,
                               highlights, clarity, stream);
  TDK_REQUIRE(dtype == TDK_F32, "unsupported dtype tag %d", dtype);
  return laplacian_t<float>(reinterpret_cast<const float*>(lum_in), reinterpret_cast<float*>(lum_out), workspace, width, height, num_gamma, sigma, shadows,
                            highlights, clarity, stream);
}

TDK_EXPORT int tdk_laplacian(const float* lum_in, float* lum_out, void* workspace, int width, int height, int num_gamma, float sigma,
                             float shadows, float highlights, float clarity, tdk_stream_t stream) {
  return tdk_laplacian_ex(lum_in, lum_out, workspace, width, height, num_gamma, sigma, shadows, highlights, clarity, TDK_F32, stream);
}
