// rcd.hip -- Ratio Corrected Demosaic, fused.
//
// Replaces reference csrc/debayer/rcd.cu:30-681 (RCDImpl::process: 13 launches over eight
// full-resolution fp32 scratch planes, ~150 B/px of HBM traffic).  Semantics: the reference's
// FIRST call on a fresh workspace, as a pure function of the input (oracle/src/rcd.c restates
// it literally):
//   * every scratch plane starts at zero, so sites outside a step's write range read as 0;
//   * the half-density planes are addressed with flat `idx / 2`: p/q_diff written at every odd
//     column (row, c') and read by step 4.2 at (row-1, oc(col-1)), (row, oc(col)),
//     (row+1, oc(col-1)+2) [P] / (row-1, oc(col-1)+2), (row, oc(col)), (row+1, oc(col-1)) [Q]
//     with oc(c) = c | 1  (rcd.cu:157-181);
//   * p/q slots step 4.1 does not write (c' in {1, W-3, W-1}, rows < 3 or > H-4) still hold
//     the same call's v_diff / h_diff at flat position row*(W/2) + (c'-1)/2 of the shared
//     buffer (rcd.cu:637-652) -- reproduced by stale_diff() below;
//   * PQ_dir at (row+-1) is looked up through slots (col-1)/2 and (col-1)/2 + 1 (rcd.cu:199-207).
//
// MI355X design: one 1024-thread workgroup per 64 x 64 output tile.  The CFA tile plus a 10-px
// halo (the dependency radius of step 5.2 back to the raw data) is read once, coalesced, into
// LDS; the nine RCD steps then run back to back on five LDS planes (cfa | v_diff->p/q_diff |
// h_diff->step-5.1 colour | VH_dir | lpf->PQ_dir + green@R/B) with plane-lifetime reuse.  In the
// last phase a wave owns a 2-row band: it sorts its finished pixels by row and writes each row
// as one contiguous run straight to HBM (no staging).  Tiles clear of the image border run a
// variant without border guards.  HBM sees the compulsory 4 B/px read (+halo from L2) and
// 12 B/px write.  LDS: 5 x 84 x 86 x 4 B = 144.5 KB, i.e. one workgroup (16 waves) per CU -- a
// four-plane / 64 x 32 / two-workgroup variant was measured slower (more halo work).  The [0,7)
// border ring (3x3 average + PPG-style green / red-blue, rcd.cu:285-493 and ppg.cu:342-389) is
// computed by the first ~100 workgroups of the same launch.  Arithmetic: same operation order as
// the oracle, no FMA contraction, IEEE divides -> bit-exact.
#include "tdk_fastdiv.h"
#include "tdk_stencils.h"
#if defined(TDK_EXPERIMENTS) && defined(TDK_RQ_FAKE_LAB)
#include "tdk_color.h"
#endif

#ifdef TDK_RCD_TIMING
// experiments: per-phase clock64() deltas of one workgroup, summed over its tiles (profiles/rcd_phase_exp.py)
__device__ unsigned long long g_rcd_phase_cycles[16];
#define RCD_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x == (gridDim.x > 1000u ? 1500u : 3u)) { const unsigned long long t_ = clock64(); atomicAdd(&g_rcd_phase_cycles[k], t_ - rcd_t0); rcd_t0 = t_; } } while (0)
#define RCD_T0_PARAM , unsigned long long& rcd_t0
#define RCD_T0_ARG , rcd_t0
// experiments: [2 b], [2 b + 1] = wall_clock64() (100 MHz) at the start / end of workgroup b of the strip kernels: which workgroups are the launch's tail
__device__ unsigned long long g_rcd_wg_times[2 * 1024];
extern "C" __attribute__((visibility("default"))) int tdk_debug_rcd_wg_times(unsigned long long* out2048) {
  return hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_rcd_wg_times), sizeof(unsigned long long) * 2048) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int tdk_debug_rcd_phase_cycles(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rcd_phase_cycles), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rcd_phase_cycles), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
#define RCD_MARK(k)
#define RCD_T0_PARAM
#define RCD_T0_ARG
#endif

namespace {

constexpr int TW = 64, TH = 64, HALO = 10;
constexpr int RW = TW + 2 * HALO, RH = TH + 2 * HALO;  // 84 x 84 working region
constexpr int S = RW + 2;                               // LDS row stride: even, so lanes that alternate between two rows stay on distinct banks
constexpr int PLANE = RH * S;
constexpr int NT = 1024;

// ---------------------------------------------------------------- divisions
// FAST tiles (interior tiles whose CFA samples all lie in {0} U [CFA_MIN, CFA_MAX], decided per tile after the
// load phase) run the 15 divisions per pixel pair as the bare core of the IEEE expansion (tdk_fastdiv.h: same
// bits, 25 instead of ~55 issue cycles).  Preconditions, with m = CFA_MIN = 2^-24, M = CFA_MAX = 2^16:
//   * every denominator is eps + a sum of non-negative terms (or >= 2e-10 for the direction statistics) and is
//     bounded by small multiples of M or M^2: 2^-33 < b < 2^44;
//   * the numerators of steps 1.2 / 4.2 are >= 1e-10; those of step 3.1 are products of non-negative factors
//     that are 0 or >= m^2 / 2 (cfa * 2 lpf) and, one level up, 0 or >= eps * 2^-68 = 2^-85 with step 3.1's eps = 1e-5
//     (a gradient >= eps times an estimate >= m * m / (3 * 2^19): a 2^-24 sample between 2^16 neighbours; the case
//     `tiny_beside_huge` of tests/test_gpu_parity.py builds it) -- never -0, never below 2^-102;
//   * the numerators of steps 5.1 / 5.2 are signed sums of products of differences and can cancel to anything:
//     div_signed() checks them per wave (|a| >= 2^-80, which also excludes +-0) and falls back to `/`.
// Tiles that fail the range check (denormal / huge / NaN samples), border tiles (their stale p/q slots alias
// samples from outside the tile) and the border ring keep the compiler's IEEE division.
constexpr float CFA_MIN = 0x1p-24f, CFA_MAX = 0x1p16f;

template <bool FAST> __device__ __forceinline__ float div_pos(float a, float b) {
  if constexpr (FAST) return tdk::div_core(a, b);
  else return a / b;
}

// `lanes`: the lanes whose quotients anything reads (a 64-bit lane mask); the others may hold garbage on either path and do
// not count in the wave's test.
template <bool FAST, int N> __device__ __forceinline__ void div_signed(const float (&a)[N], const float (&b)[N], float (&q)[N], unsigned long long lanes = ~0ull) {
  if constexpr (FAST) {
    float mn = fabsf(a[0]);
#pragma unroll
    for (int i = 1; i < N; i++) mn = fminf(mn, fabsf(a[i]));
    if ((__builtin_amdgcn_ballot_w64(!(mn >= tdk::DIV_CORE_MIN_NUM)) & lanes) == 0) {
#pragma unroll
      for (int i = 0; i < N; i++) q[i] = tdk::div_core(a[i], b[i]);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) q[i] = a[i] / b[i];
}

// the same with the wave's test made on a value the caller has prepared: mn = the smallest numerator magnitude of the lane's
// quotients that anything reads (1 where nothing does)
template <bool FAST, int N> __device__ __forceinline__ void div_signed_min(const float (&a)[N], const float (&b)[N], float (&q)[N], float mn) {
  if constexpr (FAST) {
    if (__builtin_amdgcn_ballot_w64(!(mn >= tdk::DIV_CORE_MIN_NUM)) == 0) {
#pragma unroll
      for (int i = 0; i < N; i++) q[i] = tdk::div_core(a[i], b[i]);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) q[i] = a[i] / b[i];
}

// v_diff / h_diff of the raw image at (fr, fc), or 0 outside step 1.1's range (rcd.cu:63-75)
template <typename T>
__device__ float diff_1_1(const T* __restrict__ in, int fr, int fc, int w, int h, bool vertical) {
  if (fr < 3 || fr > h - 4 || fc < 3 || fc > w - 4) return 0.0f;
  const int st = vertical ? w : 1;
  const size_t idx = (size_t)fr * w + fc;
  float c[7];
#pragma unroll
  for (int k = -3; k <= 3; k++) c[k + 3] = fmaxf(0.0f, ld(in, idx + (ptrdiff_t)k * st));
  return sqf(c[0] - 3.0f * c[1] - c[2] + 6.0f * c[3] - c[4] - 3.0f * c[5] + c[6]);
}

// Content of the shared VP/HQ buffer at the p/q slot of odd-column site (row, col) when step
// 4.1 did not write it.
// flat = row * (w / 2) + (col - 1) / 2 with w even and 0 <= (col - 1) / 2 < w / 2, so flat / w = row >> 1 and
// flat % w = (row & 1) * (w / 2) + (col - 1) / 2 -- no division (the 64-bit quotient and remainder this replaced were ~400
// instructions per call for the whole wave, in every step of the strips' border columns: their workgroups were the tail of the
// launch, profiles/r05/experiments/rcd_wg_times.txt).
template <typename T>
__device__ float stale_diff(const T* __restrict__ in, int row, int col, int w, int h, bool p_plane) {
  return diff_1_1(in, row >> 1, (row & 1) * (w >> 1) + ((col - 1) >> 1), w, h, p_plane);
}
// both planes of one slot: the 13 samples are loaded together (one memory round trip), then the two high-pass values
template <typename T>
__device__ void stale_pair(const T* __restrict__ in, int row, int col, int w, int h, float& pv, float& qv) {
  const int fr = row >> 1, fc = (row & 1) * (w >> 1) + ((col - 1) >> 1);
  pv = 0.0f; qv = 0.0f;
  if (fr < 3 || fr > h - 4 || fc < 3 || fc > w - 4) return;
  const size_t idx = (size_t)fr * w + fc;
  float v[7], u[7];
#pragma unroll
  for (int k = -3; k <= 3; k++) { v[k + 3] = ld(in, idx + (ptrdiff_t)k * w); u[k + 3] = ld(in, idx + k); }
#pragma unroll
  for (int k = 0; k < 7; k++) { v[k] = fmaxf(0.0f, v[k]); u[k] = fmaxf(0.0f, u[k]); }
  pv = sqf(v[0] - 3.0f * v[1] - v[2] + 6.0f * v[3] - v[4] - 3.0f * v[5] + v[6]);
  qv = sqf(u[0] - 3.0f * u[1] - u[2] + 6.0f * u[3] - u[4] - 3.0f * u[5] + u[6]);
}

// ---------------------------------------------------------------- border ring [0, 7)
// Intermediate image of the reference's border path at (x, y): ring < 3 -> 3x3 same-colour
// average (ppg.cu:342-389); otherwise native sample + PPG-style green (rcd.cu:285-385).
template <typename T>
__device__ f3 border_temp(const T* __restrict__ in, int x, int y, int w, int h, uint32_t pattern) {
  if (x < 0 || y < 0 || x >= w || y >= h) return mk3(0.0f, 0.0f, 0.0f);
  if (x < 3 || y < 3 || x >= w - 3 || y >= h - 3)
    return border_average([&](int xx, int yy) { return ld(in, (size_t)yy * w + xx); }, x, y, w, h, pattern);
  auto rd = [&](int xx, int yy) { return (xx >= 0 && yy >= 0 && xx < w && yy < h) ? fmaxf(0.0f, ld(in, (size_t)yy * w + xx)) : 0.0f; };
  const int c = cfa_color(y, x, pattern);
  f3 v = mk3(0.0f, 0.0f, 0.0f);
  const float pc = rd(x, y);
  if (c == 0) v.x = pc;
  else if (c == 2) v.z = pc;
  else v.y = pc;
  if (c != 1) {
    float hx[7], vy[7];
#pragma unroll
    for (int d = -3; d <= 3; d++) { hx[d + 3] = rd(x + d, y); vy[d + 3] = rd(x, y + d); }
    v.y = ppg_green(hx, vy);
  }
  return mk3(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f));
}

// Ring pixel number i: RBAND full rows (top + bottom) then CBAND columns of the middle rows.
template <typename TI, typename T>
__device__ void border_pixel(const TI* __restrict__ in, T* __restrict__ out, int w, int h, uint32_t pattern, int64_t i) {
  const int rband = min(h, 14), cband = min(w, 14);
  const int mid_rows = max(h - 14, 0);
  const int64_t n_rows = (int64_t)rband * w, n_cols = (int64_t)mid_rows * cband;
  if (i >= n_rows + n_cols) return;
  int x, y;
  if (i < n_rows) {
    const int r = (int)(i / w);
    x = (int)(i - (int64_t)r * w);
    y = (r < 7) ? r : h - rband + r;
  } else {
    const int64_t j = i - n_rows;
    const int r = (int)(j / cband), cidx = (int)(j - (int64_t)r * cband);
    y = 7 + r;
    x = (cidx < 7) ? cidx : w - cband + cidx;
  }
  // rcd_border_redblue (rcd.cu:387-493): 3x3 neighbourhood of max(0, temp)
  f3 nbv[3][3];
#pragma unroll
  for (int dy = -1; dy <= 1; dy++)
#pragma unroll
    for (int dx = -1; dx <= 1; dx++) {
      const f3 t = border_temp(in, x + dx, y + dy, w, h, pattern);
      nbv[dy + 1][dx + 1] = mk3(fmaxf(0.0f, t.x), fmaxf(0.0f, t.y), fmaxf(0.0f, t.z));
    }
  f3 col = nbv[1][1];
  if (y > 0 && x > 0 && x < w - 1 && y < h - 1) {
    auto nb = [&](int dx, int dy) { return nbv[dy + 1][dx + 1]; };
    col = ppg_redblue(nb, col, cfa_color(y, x, pattern), cfa_color(y, x + 1, pattern) == 0);
  }
  const size_t p = (size_t)y * w + x;
  st(out, p * 3, fmaxf(col.x, 0.0f));
  st(out, p * 3 + 1, fmaxf(col.y, 0.0f));
  st(out, p * 3 + 2, fmaxf(col.z, 0.0f));
}

// stand-alone ring kernel: images too small to have an interior
template <typename TI, typename T>
__global__ __launch_bounds__(256) void rcd_border(const TI* __restrict__ in, T* __restrict__ out, int w, int h, uint32_t pattern) {
  border_pixel(in, out, w, h, pattern, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// The ring in two passes through LDS (the strips launch, tdk_rcd_stream.h): a piece = 32 x 7 pixels of the top / bottom bands or
// 7 x 32 of the left / right bands (rows [7, h - 7)); the intermediate image of border_temp() is computed ONCE per position of
// the piece and its 1-px surround (border_pixel evaluates it nine times per pixel: 25 us of dependent loads at 12 MP as a
// kernel of its own), then the red/blue fill from the staged values.  Same expressions as border_pixel.  Needs w > 14, h > 14.
constexpr int RING_LEN = 32;  // (32 + 2) x 9 = 306 staged positions
constexpr int RING_TMP = (RING_LEN + 2) * 9;

// piece number b of the ring on `nthreads` threads (all of them must call: barriers inside); tmp: 3 * RING_TMP floats of LDS
template <typename TI, typename T>
__device__ __forceinline__ void ring_piece(const TI* __restrict__ in, T* __restrict__ out, int w, int h, uint32_t pattern, int nbx, int nby, int b,
                                           float* __restrict__ tmp, int nthreads) {
  int x0, y0, bw, bh;
  if (b < 2 * nbx) {  // top, then bottom band
    x0 = (b % nbx) * RING_LEN; y0 = b < nbx ? 0 : h - 7;
    bw = min(RING_LEN, w - x0); bh = 7;
  } else {            // left, then right band
    b -= 2 * nbx;
    x0 = b < nby ? 0 : w - 7; y0 = 7 + (b % nby) * RING_LEN;
    bw = 7; bh = min(RING_LEN, h - 7 - y0);
  }
  const int tw = bw + 2, th = bh + 2;
  for (int i = threadIdx.x; i < tw * th; i += nthreads) {
    const int ty = i / tw, tx = i - ty * tw;
    const f3 t = border_temp(in, x0 - 1 + tx, y0 - 1 + ty, w, h, pattern);
    tmp[i] = fmaxf(0.0f, t.x); tmp[RING_TMP + i] = fmaxf(0.0f, t.y); tmp[2 * RING_TMP + i] = fmaxf(0.0f, t.z);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bw * bh; i += nthreads) {
    const int py = i / bw, px = i - py * bw, x = x0 + px, y = y0 + py;
    const int c0 = (py + 1) * tw + px + 1;
    auto nb = [&](int dx, int dy) { const int q = c0 + dy * tw + dx; return mk3(tmp[q], tmp[RING_TMP + q], tmp[2 * RING_TMP + q]); };
    f3 col = nb(0, 0);
    if (y > 0 && x > 0 && x < w - 1 && y < h - 1) col = ppg_redblue(nb, col, cfa_color(y, x, pattern), cfa_color(y, x + 1, pattern) == 0);
    const size_t p = (size_t)y * w + x;
    st(out, p * 3, fmaxf(col.x, 0.0f));
    st(out, p * 3 + 1, fmaxf(col.y, 0.0f));
    st(out, p * 3 + 2, fmaxf(col.z, 0.0f));
  }
}

// One 64 x 64 tile.  INTERIOR = the tile and its 10-px halo keep clear of every image-border rule
// (all the `row/col >= k && <= size - k` guards of the nine steps hold for every site the tile
// touches): the guards compile away, which removes ~10 % of the instructions of 93 % of the tiles.
template <typename TI, typename T, bool INTERIOR, bool FAST>
__device__ __forceinline__ void rcd_phases(const TI* __restrict__ in, T* __restrict__ out, int w, int h, uint32_t pattern, int tile_x, int tile_y,
                                           float* __restrict__ lds RCD_T0_PARAM) {
  RCD_MARK(0);
#ifdef TDK_RCD_STOP
  if (TDK_RCD_STOP == 0) return;
#endif
  float* pA = lds;               // cfa
  float* pB = lds + PLANE;       // v_diff, then p_diff at odd columns and q_diff at (odd - 1)
  float* pC = lds + 2 * PLANE;   // h_diff, then step-5.1 colour at R/B sites
  float* pD = lds + 3 * PLANE;   // VH_dir
  float* pE = lds + 4 * PLANE;   // R/B sites: lpf then PQ_dir; their green partner (c ^ 1): green from step 3.1

  const int tid = threadIdx.x;
  const int x0 = tile_x * TW, y0 = tile_y * TH;
  const int gx0 = x0 - HALO, gy0 = y0 - HALO;  // global coords of local (0, 0); both even
  const int rowpar0 = cfa_color(0, 0, pattern) & 1, rowpar1 = cfa_color(1, 0, pattern) & 1;
  auto rb_par = [&](int gy) { return (gy & 1) ? rowpar1 : rowpar0; };  // column parity of the R/B sites of a row

  // ---- P1: step 1.1 on the halo-7 region
  {
    constexpr int K = 7, SW = TW + 2 * K, SH = TH + 2 * K;
    for (int i = tid; i < SW * SH; i += NT) {
      const int rr = i / SW, cc = i - rr * SW;
      const int r = rr + HALO - K, c = cc + HALO - K;
      const int gx = gx0 + c, gy = gy0 + r;
      float vd = 0.0f, hd = 0.0f;
      if (INTERIOR || (gy >= 3 && gy <= h - 4 && gx >= 3 && gx <= w - 4)) {
        const float* a = pA + r * S + c;
        vd = sqf(a[-3 * S] - 3.0f * a[-2 * S] - a[-S] + 6.0f * a[0] - a[S] - 3.0f * a[2 * S] + a[3 * S]);
        hd = sqf(a[-3] - 3.0f * a[-2] - a[-1] + 6.0f * a[0] - a[1] - 3.0f * a[2] + a[3]);
      }
      pB[r * S + c] = vd;
      pC[r * S + c] = hd;
    }
  }
  __syncthreads();

  RCD_MARK(1);
#ifdef TDK_RCD_STOP
  if (TDK_RCD_STOP == 1) return;
#endif
  // ---- P2: step 1.2 (VH_dir, halo 6) and step 2.1 (lpf at R/B sites, halo 7)
  {
    constexpr int K = 6, SW = TW + 2 * K, SH = TH + 2 * K;
    for (int i = tid; i < SW * SH; i += NT) {
      const int rr = i / SW, cc = i - rr * SW;
      const int r = rr + HALO - K, c = cc + HALO - K;
      const int gx = gx0 + c, gy = gy0 + r;
      float vh = 0.0f;
      if (INTERIOR || (gy >= 2 && gy <= h - 3 && gx >= 2 && gx <= w - 3)) {
        const int q = r * S + c;
        const float eps = 1e-10f;
        const float V_Stat = fmaxf(eps, pB[q - S] + pB[q] + pB[q + S]);
        const float H_Stat = fmaxf(eps, pC[q - 1] + pC[q] + pC[q + 1]);
        vh = div_pos<FAST>(V_Stat, V_Stat + H_Stat);
      }
      pD[r * S + c] = vh;
    }
  }
  {
    // checkerboard lane mapping: consecutive lanes take consecutive columns of a 2-row band, each
    // on the row whose R/B sites have that column parity -> conflict-free LDS rows (stride S even)
    constexpr int K = 7, SW = TW + 2 * K, SHB = (TH + 2 * K) / 2;
    for (int i = tid; i < SW * SHB; i += NT) {
      const int bnd = i / SW, cc = i - bnd * SW;
      const int c = (HALO - K) + cc;
      const int ra = (HALO - K) + 2 * bnd;
      const int r = ra + (((c & 1) ^ rb_par(gy0 + ra)) & 1);
      const int gy = gy0 + r;
      const int gx = gx0 + c;
      float v = 0.0f;
      if (INTERIOR || (gy >= 2 && gy <= h - 2 && gx >= 2 && gx <= w - 2)) {
        const float* a = pA + r * S + c;
        v = a[0] + 0.5f * (a[-S] + a[S] + a[-1] + a[1]) + 0.25f * (a[-S - 1] + a[-S + 1] + a[S - 1] + a[S + 1]);
      }
      pE[r * S + c] = v;
    }
  }
  __syncthreads();

  RCD_MARK(2);
#ifdef TDK_RCD_STOP
  if (TDK_RCD_STOP == 2) return;
#endif
  // ---- P3: step 3.1 (green at R/B sites, halo 5 -> pE at the green partner) and
  //          step 4.1 (p/q_diff at odd columns, halo 6 -> pB)
  {
    // checkerboard lane mapping: consecutive lanes take consecutive columns of a 2-row band, each
    // on the row whose R/B sites have that column parity -> conflict-free LDS rows (stride S even)
    constexpr int K = 5, SW = TW + 2 * K, SHB = (TH + 2 * K) / 2;
    for (int i = tid; i < SW * SHB; i += NT) {
      const int bnd = i / SW, cc = i - bnd * SW;
      const int c = (HALO - K) + cc;
      const int ra = (HALO - K) + 2 * bnd;
      const int r = ra + (((c & 1) ^ rb_par(gy0 + ra)) & 1);
      const int gy = gy0 + r;
      const int gx = gx0 + c;
      float g = 0.0f;
      if (INTERIOR || (gy >= 4 && gy <= h - 5 && gx >= 4 && gx <= w - 5)) {
        const int q = r * S + c;
        const float* a = pA + q;
        const float* L = pE + q;
        const float eps = 1e-5f;
        const float VH_c = pD[q];
        const float VH_n = 0.25f * (pD[q - S - 1] + pD[q - S + 1] + pD[q + S - 1] + pD[q + S + 1]);
        const float VH_Disc = (fabsf(0.5f - VH_c) < fabsf(0.5f - VH_n)) ? VH_n : VH_c;
        const float cfai = a[0];
        const float N_Grad = eps + fabsf(a[-S] - a[S]) + fabsf(cfai - a[-2 * S]) + fabsf(a[-S] - a[-3 * S]) + fabsf(a[-2 * S] - a[-4 * S]);
        const float S_Grad = eps + fabsf(a[S] - a[-S]) + fabsf(cfai - a[2 * S]) + fabsf(a[S] - a[3 * S]) + fabsf(a[2 * S] - a[4 * S]);
        const float W_Grad = eps + fabsf(a[-1] - a[1]) + fabsf(cfai - a[-2]) + fabsf(a[-1] - a[-3]) + fabsf(a[-2] - a[-4]);
        const float E_Grad = eps + fabsf(a[1] - a[-1]) + fabsf(cfai - a[2]) + fabsf(a[1] - a[3]) + fabsf(a[2] - a[4]);
        const float lpfi = L[0];
        const float N_Est = div_pos<FAST>(a[-S] * (lpfi + lpfi), eps + lpfi + L[-2 * S]);
        const float S_Est = div_pos<FAST>(a[S] * (lpfi + lpfi), eps + lpfi + L[2 * S]);
        const float W_Est = div_pos<FAST>(a[-1] * (lpfi + lpfi), eps + lpfi + L[-2]);
        const float E_Est = div_pos<FAST>(a[1] * (lpfi + lpfi), eps + lpfi + L[2]);
        const float V_Est = div_pos<FAST>(S_Grad * N_Est + N_Grad * S_Est, N_Grad + S_Grad);
        const float H_Est = div_pos<FAST>(W_Grad * E_Est + E_Grad * W_Est, E_Grad + W_Grad);
        g = mixf(V_Est, H_Est, VH_Disc);
      }
      pE[r * S + (c ^ 1)] = g;
    }
  }
  {
    constexpr int K = 6, SW2 = (TW + 2 * K) / 2, SH = TH + 2 * K;
    for (int i = tid; i < SW2 * SH; i += NT) {
      const int rr = i / SW2, ci = i - rr * SW2;
      const int r = rr + HALO - K;
      const int c = (HALO - K) + 2 * ci + 1;  // odd local column == odd global column
      const int gx = gx0 + c, gy = gy0 + r;
      float pd = 0.0f, qd = 0.0f;
      if (INTERIOR || (gx >= 0 && gy >= 0 && gx < w && gy < h)) {
        if (INTERIOR || (gy >= 3 && gy <= h - 4 && gx >= 3 && gx <= w - 4)) {
          const float* a = pA + r * S + c;
          pd = sqf((a[-3 * S - 3] - a[-S - 1] - a[S + 1] + a[3 * S + 3]) - 3.0f * (a[-2 * S - 2] + a[2 * S + 2]) + 6.0f * a[0]);
          qd = sqf((a[-3 * S + 3] - a[-S + 1] - a[S - 1] + a[3 * S - 3]) - 3.0f * (a[-2 * S + 2] + a[2 * S - 2]) + 6.0f * a[0]);
        } else {
          stale_pair(in, gy, gx, w, h, pd, qd);
        }
      }
      pB[r * S + c] = pd;
      pB[r * S + c - 1] = qd;
    }
  }
  __syncthreads();

  RCD_MARK(3);
#ifdef TDK_RCD_STOP
  if (TDK_RCD_STOP == 3) return;
#endif
  // ---- P4: step 4.2 (PQ_dir at R/B sites, halo 4 -> pE, replacing lpf)
  {
    // checkerboard lane mapping: consecutive lanes take consecutive columns of a 2-row band, each
    // on the row whose R/B sites have that column parity -> conflict-free LDS rows (stride S even)
    constexpr int K = 4, SW = TW + 2 * K, SHB = (TH + 2 * K) / 2;
    for (int i = tid; i < SW * SHB; i += NT) {
      const int bnd = i / SW, cc = i - bnd * SW;
      const int c = (HALO - K) + cc;
      const int ra = (HALO - K) + 2 * bnd;
      const int r = ra + (((c & 1) ^ rb_par(gy0 + ra)) & 1);
      const int gy = gy0 + r;
      const int gx = gx0 + c;
      float pq = 0.0f;
      if (INTERIOR || (gy >= 2 && gy <= h - 3 && gx >= 2 && gx <= w - 3)) {
        const int oc0 = c | 1, ocm = (c - 1) | 1;  // odd-column aliases of col and col-1 (local parity == global parity)
        const float eps = 1e-10f;
        const float P_Stat = fmaxf(eps, pB[(r - 1) * S + ocm] + pB[r * S + oc0] + pB[(r + 1) * S + ocm + 2]);
        const float Q_Stat = fmaxf(eps, pB[(r - 1) * S + ocm + 2 - 1] + pB[r * S + oc0 - 1] + pB[(r + 1) * S + ocm - 1]);
        pq = div_pos<FAST>(P_Stat, P_Stat + Q_Stat);
      }
      pE[r * S + c] = pq;
    }
  }
  __syncthreads();

  RCD_MARK(4);
#ifdef TDK_RCD_STOP
  if (TDK_RCD_STOP == 4) return;
#endif
  // ---- P5: step 5.1 (opposite colour at R/B sites, halo 3 -> pC)
  {
    // checkerboard lane mapping: consecutive lanes take consecutive columns of a 2-row band, each
    // on the row whose R/B sites have that column parity -> conflict-free LDS rows (stride S even)
    constexpr int K = 3, SW = TW + 2 * K, SHB = (TH + 2 * K) / 2;
    for (int i = tid; i < SW * SHB; i += NT) {
      const int bnd = i / SW, cc = i - bnd * SW;
      const int c = (HALO - K) + cc;
      const int ra = (HALO - K) + 2 * bnd;
      const int r = ra + (((c & 1) ^ rb_par(gy0 + ra)) & 1);
      const int gy = gy0 + r;
      const int par = c & 1;
      const int gx = gx0 + c;
      float val = 0.0f;
      if (INTERIOR || (gy >= 4 && gy <= h - 4 && gx >= 4 && gx <= w - 4)) {
        const int q = r * S + c;
        const float* a = pA + q;  // rgbc at the diagonal neighbours is their native sample
        // green plane: own site and same-class sites keep their partner parity; the diagonal
        // (other-class) sites sit on rows of opposite parity -> partner = column ^ 1 there too
        auto G = [&](int dr, int dc) { return pE[(r + dr) * S + ((c + dc) ^ 1)]; };
        // PQ_dir of row r+-1 through slots (col-1)/2 and (col-1)/2 + 1: R/B column of that row = 2*slot + parity(row+-1)
        const int parn = par ^ 1;  // rows r-1 and r+1 hold the other R/B class
        const int slot_c = ((c - 1) & ~1) + parn;  // local column of slot (col-1)/2 on the neighbour rows
        const float eps = 1e-5f;
        const float PQ_c = pE[q];
        const float PQ_n = 0.25f * (pE[(r - 1) * S + slot_c] + pE[(r - 1) * S + slot_c + 2] + pE[(r + 1) * S + slot_c] + pE[(r + 1) * S + slot_c + 2]);
        const float PQ_Disc = (fabsf(0.5f - PQ_c) < fabsf(0.5f - PQ_n)) ? PQ_n : PQ_c;
        const float g0 = G(0, 0);
        const float NW_Grad = eps + fabsf(a[-S - 1] - a[S + 1]) + fabsf(a[-S - 1] - a[-3 * S - 3]) + fabsf(g0 - G(-2, -2));
        const float NE_Grad = eps + fabsf(a[-S + 1] - a[S - 1]) + fabsf(a[-S + 1] - a[-3 * S + 3]) + fabsf(g0 - G(-2, 2));
        const float SW_Grad = eps + fabsf(a[-S + 1] - a[S - 1]) + fabsf(a[S - 1] - a[3 * S - 3]) + fabsf(g0 - G(2, -2));
        const float SE_Grad = eps + fabsf(a[-S - 1] - a[S + 1]) + fabsf(a[S + 1] - a[3 * S + 3]) + fabsf(g0 - G(2, 2));
        const float NW_Est = a[-S - 1] - G(-1, -1);
        const float NE_Est = a[-S + 1] - G(-1, 1);
        const float SW_Est = a[S - 1] - G(1, -1);
        const float SE_Est = a[S + 1] - G(1, 1);
        const float num[2] = {NW_Grad * SE_Est + SE_Grad * NW_Est, NE_Grad * SW_Est + SW_Grad * NE_Est};
        const float den[2] = {NW_Grad + SE_Grad, NE_Grad + SW_Grad};
        float est[2];  // P_Est, Q_Est
        div_signed<FAST>(num, den, est);
        val = g0 + mixf(est[0], est[1], PQ_Disc);
      }
      pC[r * S + c] = val;
    }
  }
  __syncthreads();

  RCD_MARK(5);
#ifdef TDK_RCD_STOP
  if (TDK_RCD_STOP == 5) return;
#endif
  // ---- P6: step 5.2 at green sites + write_output (margin 7), half a tile (32 rows) at a time.
  // One lane per COLUMN of a 2-row band (checkerboard, as above) produces the band's green site and
  // R/B site of that column; the two pixels are then sorted by ROW, so that for each row the wave
  // holds 64 consecutive pixels and writes them straight to HBM as one contiguous run (fp32: 12 B
  // per lane; fp16: even lanes write their own and their right neighbour's pixel as 12 B).  No LDS
  // staging, no barrier inside the phase.
  {
    for (int half = 0; half < 2; half++) {
      {
        const int bnd = tid >> 6, cc = tid & 63;
        const int ty_a = half * 32 + 2 * bnd;            // tile-local first row of the band
        const int c = cc + HALO;
        const int par_a = rb_par(y0 + ty_a);             // R/B column parity of row ty_a
        const int rb_off = ((c & 1) ^ par_a) & 1;        // row (0/1 inside the band) of this column's R/B site
        float rbpx[3], gpx[3];
        // --- R/B site: native, green from step 3.1, other colour from step 5.1
        {
          const int ty = ty_a + rb_off, r = ty + HALO, q = r * S + c;
          const int row_color = cfa_color(y0 + ty, c & 1, pattern);  // colour of this row's R/B sites
          const float native = pA[q], green = pE[r * S + (c ^ 1)], other = pC[q];
          rbpx[0] = fmaxf(row_color == 0 ? native : other, 0.0f);
          rbpx[1] = fmaxf(green, 0.0f);
          rbpx[2] = fmaxf(row_color == 0 ? other : native, 0.0f);
        }
        // --- green site: step 5.2
        {
          const int ty = ty_a + (rb_off ^ 1), r = ty + HALO, q = r * S + c;
          const int row_color = cfa_color(y0 + ty, (c & 1) ^ 1, pattern);  // colour of this row's R/B sites (left/right neighbours)
          const float eps = 1e-5f;
          const float VH_c = pD[q];
          const float VH_n = 0.25f * (pD[q - S - 1] + pD[q - S + 1] + pD[q + S - 1] + pD[q + S + 1]);
          const float VH_Disc = (fabsf(0.5f - VH_c) < fabsf(0.5f - VH_n)) ? VH_n : VH_c;
          const float g = pA[q];
          const float N1 = eps + fabsf(g - pA[q - 2 * S]);
          const float S1 = eps + fabsf(g - pA[q + 2 * S]);
          const float W1 = eps + fabsf(g - pA[q - 2]);
          const float E1 = eps + fabsf(g - pA[q + 2]);
          // green at the four R/B neighbours (stored at their partner column)
          const float gN = pE[(r - 1) * S + (c ^ 1)], gS = pE[(r + 1) * S + (c ^ 1)];
          const float gW = pE[r * S + ((c - 1) ^ 1)], gE = pE[r * S + ((c + 1) ^ 1)];
          // colour `row_color` is native left/right (this row's R/B sites) and from step 5.1
          // above/below; the other colour the other way round.
          float num[4], den[4], est[4];  // V_Est, H_Est of rgb0 then of rgb2
#pragma unroll
          for (int ci = 0; ci < 2; ci++) {
            const int col = ci * 2;  // rgbc = rgb0 then rgb2 (rcd.cu:256-258)
            const bool native_h = (col == row_color);
            const float* ph = native_h ? pA : pC;  // horizontal neighbours
            const float* pv = native_h ? pC : pA;  // vertical neighbours
            const float cN = pv[q - S], cS = pv[q + S], cW = ph[q - 1], cE = ph[q + 1];
            const float cN3 = pv[q - 3 * S], cS3 = pv[q + 3 * S], cW3 = ph[q - 3], cE3 = ph[q + 3];
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
            num[2 * ci] = N_Grad * S_Est + S_Grad * N_Est;
            den[2 * ci] = N_Grad + S_Grad;
            num[2 * ci + 1] = E_Grad * W_Est + W_Grad * E_Est;
            den[2 * ci + 1] = E_Grad + W_Grad;
          }
          div_signed<FAST>(num, den, est);
          gpx[0] = fmaxf(g + mixf(est[0], est[1], VH_Disc), 0.0f);
          gpx[1] = fmaxf(g, 0.0f);
          gpx[2] = fmaxf(g + mixf(est[2], est[3], VH_Disc), 0.0f);
        }
        // --- sort by row and write: row ty_a holds this column's R/B pixel when rb_off == 0, else its green pixel
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
          const bool take_rb = (rb_off == rr);
          const float v0 = take_rb ? rbpx[0] : gpx[0], v1 = take_rb ? rbpx[1] : gpx[1], v2 = take_rb ? rbpx[2] : gpx[2];
          const int x = x0 + cc, y = y0 + ty_a + rr;
          const size_t p = (size_t)y * w + x;
          if constexpr (INTERIOR) {
            if constexpr (sizeof(T) == 4) {
              struct alignas(4) px3 { float a, b, c; };
              *reinterpret_cast<px3*>(reinterpret_cast<float*>(out) + p * 3) = px3{v0, v1, v2};
            } else {
              // the right neighbour's pixel via DPP quad_perm(1, 0, 3, 2): lane i reads lane i ^ 1
              const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v0), 0xB1, 0xF, 0xF, true));
              const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v1), 0xB1, 0xF, 0xF, true));
              const float n2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v2), 0xB1, 0xF, 0xF, true));
              if ((cc & 1) == 0) {  // x0 and w are even: p is even, the 12-B pair starts on a 4-B boundary
                const __half2 h0 = __floats2half2_rn(v0, v1), h1 = __floats2half2_rn(v2, n0), h2 = __floats2half2_rn(n1, n2);
                struct alignas(4) pair6 { __half2 a, b, c; };
                *reinterpret_cast<pair6*>(reinterpret_cast<__half*>(out) + p * 3) = pair6{h0, h1, h2};
              }
            }
          } else if (x < w && y >= 7 && y < h - 7 && x >= 7 && x < w - 7) {
            st(out, p * 3, v0); st(out, p * 3 + 1, v1); st(out, p * 3 + 2, v2);
          }
        }
      }
    }
  }
  RCD_MARK(6);
}

// ---------------------------------------------------------------- load phase
// The 84 x 84 sample window of an interior tile as 8-B (fp32) / 4-B (fp16) pairs: 84 rows x 42 pairs = 3528
// loads per tile, at most PAIRS_PT = 4 per thread.  The persistent workgroup issues them for its NEXT tile
// right after the load barrier of the current one, so they are in flight during the six compute phases: with one
// 1024-thread workgroup per CU nothing else can hide that latency (measured: the load phase alone was 46 us of
// the 225 us kernel).  Needs w even (always) and an 8-B / 4-B aligned base pointer (`wide_ok`).
constexpr int PAIRS_ROW = RW / 2, PAIRS = PAIRS_ROW * RH, PAIRS_PT = (PAIRS + NT - 1) / NT;

template <typename TI> struct Staged;
template <> struct Staged<float> {
  float2 v[PAIRS_PT];
  __device__ __forceinline__ void fetch(const float* p, int k) { v[k] = *reinterpret_cast<const float2*>(p); }
  __device__ __forceinline__ float2 get(int k) const { return v[k]; }
};
template <> struct Staged<__half> {
  uint32_t v[PAIRS_PT];
  __device__ __forceinline__ void fetch(const __half* p, int k) { v[k] = *reinterpret_cast<const uint32_t*>(p); }
  __device__ __forceinline__ float2 get(int k) const { return __half22float2(__builtin_bit_cast(__half2, v[k])); }
};

template <typename TI>
__device__ __forceinline__ void prefetch_tile(Staged<TI>& st, const TI* __restrict__ in, int w, int tile_x, int tile_y) {
  const int gx0 = tile_x * TW - HALO, gy0 = tile_y * TH - HALO;
#pragma unroll
  for (int k = 0; k < PAIRS_PT; k++) {
    const int j = (int)threadIdx.x + k * NT;
    if (j < PAIRS) {
      const int r = j / PAIRS_ROW, cp = j - r * PAIRS_ROW;
      st.fetch(in + (size_t)(gy0 + r) * w + gx0 + 2 * cp, k);
    }
  }
}

struct Range {  // of the staged samples, as bit patterns: samples are >= +0, so integer order == float order
  uint32_t lo = 0xffffffffu, hi = 0u;  // min over (bits - 1) [zero wraps to the top: zeros are allowed], max over bits
  __device__ __forceinline__ void add(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    lo = min(lo, u - 1u);
    hi = max(hi, u);
  }
  __device__ __forceinline__ bool ok() const { return lo >= __builtin_bit_cast(uint32_t, CFA_MIN) - 1u && hi <= __builtin_bit_cast(uint32_t, CFA_MAX); }
};

// P0: cfa = max(0, in) into plane A (zero outside the image; a NaN cannot survive v_max_f32 with 0), from the
// staged pairs when the tile was prefetched, else by scalar loads.  Interior tiles also decide the division
// flavour: every thread tracks the range of what it stages, each wave posts its verdict, and after the load
// barrier every thread folds the 16 verdicts -- a workgroup-uniform answer.  Returns after that barrier.
template <typename TI, bool INTERIOR>
__device__ __forceinline__ bool load_tile(const TI* __restrict__ in, int w, int h, int tile_x, int tile_y, float* __restrict__ lds, const Staged<TI>* st) {
  float* pA = lds;
  const int tid = threadIdx.x;
  const int gx0 = tile_x * TW - HALO, gy0 = tile_y * TH - HALO;
  Range rg;
  if (INTERIOR && st) {
#pragma unroll
    for (int k = 0; k < PAIRS_PT; k++) {
      const int j = tid + k * NT;
      if (j < PAIRS) {
        const int r = j / PAIRS_ROW, cp = j - r * PAIRS_ROW;
        const float2 s2 = st->get(k);
        const float a = fmaxf(0.0f, s2.x), b = fmaxf(0.0f, s2.y);
        *reinterpret_cast<float2*>(pA + r * S + 2 * cp) = make_float2(a, b);  // S even: 8-B aligned
        rg.add(a);
        rg.add(b);
      }
    }
  } else {
    for (int i = tid; i < RW * RH; i += NT) {
      const int r = i / RW, c = i - r * RW;
      const int gx = gx0 + c, gy = gy0 + r;
      const float v = (INTERIOR || (gx >= 0 && gy >= 0 && gx < w && gy < h)) ? fmaxf(0.0f, ld(in, (size_t)gy * w + gx)) : 0.0f;
      pA[r * S + c] = v;
      if constexpr (INTERIOR) rg.add(v);
    }
  }
  if constexpr (INTERIOR) {
    uint32_t* verdict = reinterpret_cast<uint32_t*>(lds + 5 * PLANE);  // NT / 64 = 16 words behind the planes
    const bool wave_ok = __builtin_amdgcn_ballot_w64(!rg.ok()) == 0;
    if ((tid & 63) == 0) verdict[tid >> 6] = wave_ok ? 1u : 0u;
    __syncthreads();
    uint32_t all = 1u;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) all &= verdict[k];
    return __builtin_amdgcn_readfirstlane(all) != 0;
  } else {
    __syncthreads();
    return false;
  }
}

#include "tdk_rcd_stream.h"
// The register-blocked strips (a lane owns four columns, own-column 16-byte LDS reads, neighbour taps by DPP): the same bits as
// rs::rcd_stream with 55 % fewer LDS and 6 % fewer VALU instructions, on HALF the waves (half-wave rows: 12 waves per CU in the
// same LDS, 112 VGPRs each).  With the GPU to itself it is 12 % slower than rs::rcd_stream (183 against 164 us: three waves per
// SIMD no longer cover the dependent chains, profiles/r04/experiments/rcd_quad.txt); with other frames' kernels in flight on
// other streams it is the faster choice for the whole (+3 % frames per second on the 12 MP chain,
// profiles/r04/experiments/coresidency.txt): it leaves five wave slots and 176 VGPRs per SIMD, where the streaming kernels of
// the other frames (luminance, Wiener finish, tone map, metrics: 150 us per frame) run next to it instead of after it.
// Callers that keep several frames in flight ask for it with TDK_RCD_CONCURRENT.
#include "tdk_rcd_quad.h"

// Persistent workgroups (one per CU: the five planes fill its LDS).  Work list: `nborder` chunks of the border
// ring (independent of the tiles: disjoint output pixels, input read-only), then the tiles; workgroup b takes
// tiles b, b + G, b + 2G, ...  The column of a tile is rotated by 7 per tile row so that the slower border-column
// tiles spread over the workgroups instead of landing on the same two.
template <typename TI, typename T>   // TI: storage type of the mosaic, T: of the RGB result
__global__ __launch_bounds__(NT) void rcd_interior(const TI* __restrict__ in, T* __restrict__ out, int w, int h, uint32_t pattern, int wide_ok, int nborder,
                                                    int tiles_x, int tiles_y) {
  extern __shared__ float lds[];
  const int G = (int)gridDim.x;
  for (int i = (int)blockIdx.x; i < nborder; i += G) border_pixel(in, out, w, h, pattern, (int64_t)i * NT + threadIdx.x);

  const int ntiles = tiles_x * tiles_y;
  auto locate = [&](int t, int& tx, int& ty) -> bool {  // tile number -> position; true: clear of every image-border rule
    ty = t / tiles_x;
    tx = (t - ty * tiles_x + 7 * ty) % tiles_x;
    return tx >= 1 && ty >= 1 && tx * TW + TW + HALO <= w && ty * TH + TH + HALO <= h;
  };
#ifdef TDK_RCD_TIMING
  unsigned long long rcd_t0 = clock64();
#endif
  Staged<TI> st;
  int t = (int)blockIdx.x, tx = 0, ty = 0;
  bool interior = false, staged = false;
  if (t < ntiles) {
    interior = locate(t, tx, ty);
    if (interior && wide_ok) { prefetch_tile(st, in, w, tx, ty); staged = true; }
  }
  while (t < ntiles) {
    const bool fast = interior ? load_tile<TI, true>(in, w, h, tx, ty, lds, staged ? &st : nullptr) : load_tile<TI, false>(in, w, h, tx, ty, lds, nullptr);
    // next tile: position, and its samples on their way while this one computes
    const int tn = t + G;
    int txn = 0, tyn = 0;
    bool interior_n = false, staged_n = false;
    if (tn < ntiles) {
      interior_n = locate(tn, txn, tyn);
      if (interior_n && wide_ok) { prefetch_tile(st, in, w, txn, tyn); staged_n = true; }
    }
    if (!interior) rcd_phases<TI, T, false, false>(in, out, w, h, pattern, tx, ty, lds RCD_T0_ARG);
    else if (fast) rcd_phases<TI, T, true, true>(in, out, w, h, pattern, tx, ty, lds RCD_T0_ARG);
    else rcd_phases<TI, T, true, false>(in, out, w, h, pattern, tx, ty, lds RCD_T0_ARG);
    __syncthreads();  // the last phase reads planes the next load phase overwrites
    t = tn; tx = txn; ty = tyn; interior = interior_n; staged = staged_n;
  }
}

template <typename TI, typename T>
int launch_mixed(const void* bayer, void* rgb, int w, int h, uint32_t pattern, unsigned flags, hipStream_t s) {
  const TI* in = reinterpret_cast<const TI*>(bayer);
  T* out = reinterpret_cast<T*>(rgb);
  const int wide_ok = tdk_aligned(bayer, 2 * sizeof(TI));  // sample pairs load as one 8-B / 4-B access (w is even)
  constexpr size_t lds_bytes = (size_t)5 * PLANE * sizeof(float) + (NT / 64) * sizeof(uint32_t);  // planes + the per-wave range verdicts
  {
    const int rc = tdk_raise_lds_limit(reinterpret_cast<const void*>(&rcd_interior<TI, T>), (int)lds_bytes, "tdk_rcd(hipFuncSetAttribute)");
    if (rc != TDK_OK) return rc;
  }
  const int rband = h < 14 ? h : 14, cband = w < 14 ? w : 14;
  const int64_t nring = (int64_t)rband * w + (int64_t)(h > 14 ? h - 14 : 0) * cband;
  if (w > 14 && h > 14) {
    const int nborder = (int)tdk_div_up64(nring, NT), tiles_x = tdk_div_up(w, TW), tiles_y = tdk_div_up(h, TH);
    // Frames that hold a strip (and whose samples load as pairs) go to rcd_stream, ring included.
    // (the strips also store pixel pairs: 8-B / 4-B aligned output; an offset view takes the tile kernel)
    bool stream = wide_ok && tdk_aligned(rgb, 2 * sizeof(T)) && w >= rs::TWS + 2 * rs::HALO && h >= 64 && !(flags & TDK_RCD_TILE_KERNEL);
#ifdef TDK_EXPERIMENTS
    if (const char* e = getenv("TDK_RCD_STREAM")) stream = stream && atoi(e) != 0;
#endif
    if (stream) {
      const int nstrips = tdk_div_up(w, rs::TWS);
      // segments: as many workgroups as the chip holds at once (3 per CU), but no segment shorter than 64 rows (20 rows of
      // warm-up / drain per segment)
      int nsegs = (rs::WG_PER_CU * tdk_device_cus()) / nstrips;
#ifdef TDK_EXPERIMENTS
      if (const char* e = getenv("TDK_RCD_NSEGS")) nsegs = atoi(e);  // fewer, longer segments: less warm-up per row, fewer workgroups than slots
#endif
      if (nsegs < 1) nsegs = 1;
      if (nsegs > h / 64) nsegs = h / 64;
      int seg_rows = (tdk_div_up(h, nsegs) + 1) & ~1;  // even: segment origins keep the CFA phase
      // a single segment of an odd-height frame: its origin is row 0 whatever its length, so it may be odd (two
      // overlapping segments of h - 1 rows would each walk nearly the whole frame)
      if (seg_rows > h) seg_rows = nsegs == 1 ? h : (h & ~1);
      nsegs = tdk_div_up(h, seg_rows);
      const int nwg = nstrips * nsegs;
      const int nbx = tdk_div_up(w, RING_LEN), nby = tdk_div_up(h - 14, RING_LEN);  // ring pieces: a share of the strip workgroups takes one each
      size_t strip_lds = rs::LDS_BYTES;
#ifdef TDK_EXPERIMENTS
      if (const char* e = getenv("TDK_RCD_LDS_PAD")) strip_lds += (size_t)atoi(e);  // fewer resident workgroups per CU (occupancy experiment)
#endif
      bool quad = (flags & TDK_RCD_CONCURRENT) != 0;
#if defined(TDK_EXPERIMENTS) && defined(TDK_RS_RB)
      quad = true;
#endif
#if defined(TDK_EXPERIMENTS) && !defined(TDK_RS_RB)
      if (const char* e = getenv("TDK_RCD_QUAD")) {  // columns per lane of the register-tap variant: 2, else 4
        if (atoi(e) == 2) {
          const int rcq = tdk_raise_lds_limit(reinterpret_cast<const void*>(&rq::rcd_quad<2, TI, T>), 160 * 1024, "tdk_rcd(hipFuncSetAttribute)");
          if (rcq != TDK_OK) return rcq;
          TDK_LAUNCH("tdk_rcd", (rq::rcd_quad<2, TI, T>), dim3((unsigned)nwg), dim3(rq::Geo<2>::NT), strip_lds, s, in, out, w, h, pattern, nstrips, seg_rows, nbx, nby);
          return TDK_OK;
        }
        quad = atoi(e) != 0;
      }
#endif
      // binary16 results: the approximate arithmetic flavour of the strips (tdk_rcd_stream.h), unless the caller asks for the
      // oracle's bits rounded once (TDK_RCD_EXACT)
      if constexpr (sizeof(T) == 2) {
        if (!(flags & TDK_RCD_EXACT)) {
          if (quad) {
            const int rcq = tdk_raise_lds_limit(reinterpret_cast<const void*>(&rq::rcd_quad<4, TI, T, true>), 160 * 1024, "tdk_rcd(hipFuncSetAttribute)");
            if (rcq != TDK_OK) return rcq;
            TDK_LAUNCH("tdk_rcd(concurrent)", (rq::rcd_quad<4, TI, T, true>), dim3((unsigned)nwg), dim3(rq::Geo<4>::NT), strip_lds, s, in, out, w, h, pattern, nstrips, seg_rows, nbx, nby);
            return TDK_OK;
          }
          const int rca = tdk_raise_lds_limit(reinterpret_cast<const void*>(&rs::rcd_stream<TI, T, true>), 160 * 1024, "tdk_rcd(hipFuncSetAttribute)");
          if (rca != TDK_OK) return rca;
          TDK_LAUNCH("tdk_rcd", (rs::rcd_stream<TI, T, true>), dim3((unsigned)nwg), dim3(rs::NT), strip_lds, s, in, out, w, h, pattern, nstrips, seg_rows, nbx, nby);
          return TDK_OK;
        }
      }
      if (quad) {
        const int rcq = tdk_raise_lds_limit(reinterpret_cast<const void*>(&rq::rcd_quad<4, TI, T>), 160 * 1024, "tdk_rcd(hipFuncSetAttribute)");
        if (rcq != TDK_OK) return rcq;
        TDK_LAUNCH("tdk_rcd(concurrent)", (rq::rcd_quad<4, TI, T>), dim3((unsigned)nwg), dim3(rq::Geo<4>::NT), strip_lds, s, in, out, w, h, pattern, nstrips, seg_rows, nbx, nby);
        return TDK_OK;
      }
      const int rc = tdk_raise_lds_limit(reinterpret_cast<const void*>(&rs::rcd_stream<TI, T>), 160 * 1024, "tdk_rcd(hipFuncSetAttribute)");
      if (rc != TDK_OK) return rc;
      TDK_LAUNCH("tdk_rcd", (rs::rcd_stream<TI, T>), dim3((unsigned)nwg), dim3(rs::NT), strip_lds, s, in, out, w, h, pattern, nstrips, seg_rows, nbx, nby);
      return TDK_OK;
    }
    const int ntiles_launch = tiles_x * tiles_y;
    int grid = tdk_device_cus();  // one resident workgroup per CU
#ifdef TDK_EXPERIMENTS
    if (const char* e = getenv("TDK_RCD_GRID")) grid = atoi(e) > 0 ? atoi(e) : nborder + ntiles_launch;  // 0 = one item per workgroup
#endif
    if (grid > nborder + ntiles_launch) grid = nborder + ntiles_launch;
    TDK_LAUNCH("tdk_rcd", (rcd_interior<TI, T>), dim3((unsigned)grid), dim3(NT), lds_bytes, s, in, out, w, h, pattern, wide_ok, nborder, tiles_x, tiles_y);
  } else {
    TDK_LAUNCH("tdk_rcd(border)", (rcd_border<TI, T>), dim3((unsigned)tdk_div_up64(nring, 256)), dim3(256), 0, s, in, out, w, h, pattern);
  }
  return TDK_OK;
}

template <typename T> int launch(const void* bayer, void* rgb, int w, int h, uint32_t pattern, unsigned flags, hipStream_t s) {
  return launch_mixed<T, T>(bayer, rgb, w, h, pattern, flags, s);
}

}  // namespace

TDK_EXPORT size_t tdk_rcd_workspace_bytes(int, int) { return 0; }

TDK_EXPORT int tdk_rcd_ex(const void* bayer, void* rgb, void* /*workspace*/, int width, int height, uint32_t pattern, int dtype, unsigned flags,
                          tdk_stream_t stream) {
  TDK_REQUIRE(bayer && rgb, "tdk_rcd: null pointer");
  TDK_REQUIRE(width > 0 && height > 0, "tdk_rcd: invalid size %dx%d", width, height);
  TDK_REQUIRE((width & 1) == 0, "tdk_rcd: width must be even (the reference packs half-density planes as idx/2)");
  TDK_REQUIRE(pattern == TDK_PATTERN_RGGB || pattern == TDK_PATTERN_BGGR || pattern == TDK_PATTERN_GRBG || pattern == TDK_PATTERN_GBRG,
              "tdk_rcd: invalid Bayer pattern 0x%08x", pattern);
  TDK_REQUIRE((flags & ~(TDK_RCD_TILE_KERNEL | TDK_RCD_CONCURRENT | TDK_RCD_EXACT)) == 0, "tdk_rcd: unknown flags 0x%x", flags);
  TDK_DISPATCH_DTYPE(dtype, T, return launch<T>(bayer, rgb, width, height, pattern, flags, tdk_stream(stream)));
  return TDK_OK;
}

TDK_EXPORT int tdk_rcd(const void* bayer, void* rgb, void* workspace, int width, int height, uint32_t pattern, int dtype, tdk_stream_t stream) {
  return tdk_rcd_ex(bayer, rgb, workspace, width, height, pattern, dtype, 0u, stream);
}

// decode12_float -> apply_white_balance -> RCD.process as one call.  Two launches: a streaming kernel that decodes and
// white-balances the mosaic into the caller's fp32 scratch plane (codec.hip), then the tile kernel.  Decoding inside the
// tile staging was built and measured SLOWER than this (profiles/rcd_packed_exp.py: 284 us + ring against 17 + 230 us):
// the RCD kernel runs one 1024-thread workgroup per CU, so everything in its load phase is exposed latency, while a
// streaming kernel runs at HBM speed.
int tdk_decode12_wb_plane(const uint8_t* packed, float* mosaic, const float* gains, int width, int height, uint32_t pattern, int ids_format, hipStream_t s);

TDK_EXPORT size_t tdk_decode12_wb_rcd_workspace_bytes(int width, int height) {
  return width > 0 && height > 0 ? tdk_align_up((size_t)width * height * sizeof(float), 256) : 0;
}

TDK_EXPORT int tdk_decode12_wb_rcd_ex(const uint8_t* packed, void* rgb, void* workspace, const float* gains, int width, int height, uint32_t pattern,
                                      int ids_format, int out_dtype, unsigned flags, tdk_stream_t stream) {
  TDK_REQUIRE(packed && rgb && workspace, "tdk_decode12_wb_rcd: null pointer");
  TDK_REQUIRE((flags & ~(TDK_RCD_TILE_KERNEL | TDK_RCD_CONCURRENT | TDK_RCD_EXACT)) == 0, "tdk_decode12_wb_rcd: unknown flags 0x%x", flags);
  TDK_REQUIRE(width > 0 && height > 0, "tdk_decode12_wb_rcd: invalid size %dx%d", width, height);
  TDK_REQUIRE((width & 1) == 0, "tdk_decode12_wb_rcd: width must be even (pixel pairs share three bytes; RCD packs half-density planes as idx/2)");
  TDK_REQUIRE(pattern == TDK_PATTERN_RGGB || pattern == TDK_PATTERN_BGGR || pattern == TDK_PATTERN_GRBG || pattern == TDK_PATTERN_GBRG,
              "tdk_decode12_wb_rcd: invalid Bayer pattern 0x%08x", pattern);
  float* mosaic = reinterpret_cast<float*>(workspace);
  const int rc = tdk_decode12_wb_plane(packed, mosaic, gains, width, height, pattern, ids_format, tdk_stream(stream));
  if (rc != TDK_OK) return rc;
  if (out_dtype == TDK_F32) return launch_mixed<float, float>(mosaic, rgb, width, height, pattern, flags, tdk_stream(stream));
  if (out_dtype == TDK_F16) return launch_mixed<float, __half>(mosaic, rgb, width, height, pattern, flags, tdk_stream(stream));
  tdk_set_error("unsupported dtype tag %d", out_dtype);
  return TDK_ERR_INVALID_ARGUMENT;
}

TDK_EXPORT int tdk_decode12_wb_rcd(const uint8_t* packed, void* rgb, void* workspace, const float* gains, int width, int height, uint32_t pattern,
                                   int ids_format, int out_dtype, tdk_stream_t stream) {
  return tdk_decode12_wb_rcd_ex(packed, rgb, workspace, gains, width, height, pattern, ids_format, out_dtype, 0u, stream);
}
