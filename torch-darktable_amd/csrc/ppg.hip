// ppg.hip -- Pattern Pixel Grouping demosaic.
//
// Replaces reference csrc/debayer/ppg.cu:21-476 (PPGImpl::process: border_interpolate,
// pre_median, ppg_demosaic_green, ppg_demosaic_redblue -- 3-4 launches, a zero-filled output
// and a persistent float3 temp image).
//
// MI355X design: ONE fused kernel.  A 512-thread workgroup owns a 64 x 32 output tile:
//   1. raw CFA tile + 4-px halo -> LDS (zero outside the image), coalesced row reads;
//   2. the reference's intermediate "green + sparse R/B" image is produced only for the
//      66 x 34 region the tile needs, straight into three LDS planes (3-px image ring from
//      the 3x3 same-colour average, everything else from the gradient-selected green);
//   3. red/blue fill reads that LDS region and each thread emits 4 consecutive pixels as
//      three 16-B stores.
// HBM traffic is the compulsory 4 + 12 B/px (fp32) instead of ~52 B/px; the intermediate
// never leaves the CU.  Arithmetic is add/mul/abs/min/max only, same term order as the oracle,
// so results are bit-exact.  The optional pre-median stays a separate plane-to-plane kernel
// (its 5x5 cross footprint would double the halo for a feature the presets leave off).
#include "tdk_stencils.h"

namespace {

constexpr int TW = 64, TH = 32;
constexpr int PNT = 512;                    // threads per workgroup: 4 workgroups x 8 waves per CU (LDS 39 KB each)
constexpr int RH = 4;                       // raw halo: 1 (red/blue) + 3 (green)
constexpr int RW_ = TW + 2 * RH, RHT = TH + 2 * RH;   // 72 x 40
constexpr int GW = TW + 2, GH = TH + 2;     // 66 x 34 intermediate region
// LDS layouts, chosen for the lane strides of the phases (b32 accesses of lanes 2 or 4 columns apart are 2- / 4-way
// bank conflicts in a row-major tile: measured 58 % of this kernel's LDS cycles):
//   raw tile: even columns, then odd columns of a row -- the green phase walks one CFA class, i.e. every other column;
//   intermediate planes: columns de-interleaved by 4 -- the red/blue phase gives a lane 4 consecutive pixels.
// With these, consecutive lanes touch consecutive words in both phases.  Row stride 72 in both (== 8 mod 32: the
// 16-lane halves of a wave that sit two rows apart in the last phase land on disjoint banks); 40 896 B in all, so four
// workgroups still share a CU.
constexpr int RS = RW_, RHALF = RW_ / 2;    // raw: row stride 72, odd columns start at +36
constexpr int GS = 72, GQ = 17;             // planes: row stride 72, quarter q (= column mod 4) starts at + 17 q
__device__ __forceinline__ int raw_at(int r, int c) { return r * RS + (c >> 1) + (c & 1) * RHALF; }
__device__ __forceinline__ int pln_at(int r, int c) { return r * GS + (c >> 2) + (c & 3) * GQ; }

// One 64 x 32 tile.  INTERIOR = the tile, its 1-px intermediate ring and its 4-px raw halo stay at least
// 3 px inside the image: no in-image tests, no 3x3 border-average path.
template <typename T, bool INTERIOR>
__device__ __forceinline__ void ppg_tile(const T* __restrict__ src, const T* __restrict__ orig, T* __restrict__ out, int width, int height,
                                         uint32_t pattern, int vec_ok, float* __restrict__ raw, float* __restrict__ pr, float* __restrict__ pg,
                                         float* __restrict__ pb, int tx, int ty) {
  const int x0 = tx * TW, y0 = ty * TH;

  {
    // all global loads of the thread are issued before the first LDS store (a load -> store loop
    // would expose one memory latency per iteration)
    constexpr int NLD = (RW_ * RHT + PNT - 1) / PNT;
    float tmp[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const int i = threadIdx.x + k * PNT;
      const int r = i / RW_, c = i - r * RW_;
      const int gx = x0 - RH + c, gy = y0 - RH + r;
      tmp[k] = (i < RW_ * RHT && (INTERIOR || (gx >= 0 && gy >= 0 && gx < width && gy < height))) ? ld(src, (size_t)gy * width + gx) : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      const int i = threadIdx.x + k * PNT;
      const int r = i / RW_, c = i - r * RW_;
      if (i < RW_ * RHT) raw[raw_at(r, c)] = tmp[k];
    }
  }
  __syncthreads();

  // The 66 x 34 region is walked one CFA site class (row parity, column parity) at a time, so every
  // lane of a wave sits on the same kind of site: the green interpolation at red/blue sites runs
  // with all lanes active instead of under a checkerboard mask, and green sites only copy.
  constexpr int CW = GW / 2, CH = GH / 2;  // 33 x 17 sites per class
#pragma unroll
  for (int cls = 0; cls < 4; cls++) {  // unrolled: the column parity of a class decides which half of a raw row a tap reads
    const int rp = cls >> 1, cp = cls & 1;
    const int cc = cfa_color(y0 - 1 + rp, x0 - 1 + cp, pattern);  // CFA colour of the whole class (scalar: x0, y0 are even)
    for (int i = threadIdx.x; i < CW * CH; i += PNT) {
      const int rr = i / CW, ci = i - rr * CW, r = 2 * rr + rp, c = 2 * ci + cp;
      const int gx = x0 - 1 + c, gy = y0 - 1 + r;
      f3 v = mk3(0.0f, 0.0f, 0.0f);
      if (INTERIOR || (gx >= 0 && gy >= 0 && gx < width && gy < height)) {
        if (!INTERIOR && (gx < 3 || gy < 3 || gx >= width - 3 || gy >= height - 3)) {
          // the 3x3 neighbourhood of a ring pixel lies inside the staged raw tile (4-px halo); without a pre-median the staged
          // plane IS the original mosaic, so the nine samples come from LDS instead of nine global loads per site
          if (src == orig) v = border_average([&](int xx, int yy) { return raw[raw_at(yy - (y0 - RH), xx - (x0 - RH))]; }, gx, gy, width, height, pattern);
          else v = border_average([&](int xx, int yy) { return ld(orig, (size_t)yy * width + xx); }, gx, gy, width, height, pattern);
        } else {
          // raw column c + RH - 1 + d = 2 ci + (cp + RH - 1 + d): half and offset inside it are compile-time per tap
          const float* rrow = raw + (r + RH - 1) * RS + ci;
          auto tap = [&](int d, int dr) { const int k = cp + RH - 1 + d; return rrow[dr * RS + (k >> 1) + (k & 1) * RHALF]; };
          const float pc = tap(0, 0);
          if (cc == 0) v.x = pc;
          else if (cc == 2) v.z = pc;
          else v.y = pc;
          if (cc != 1) {
            float h[7], vv[7];
#pragma unroll
            for (int d = -3; d <= 3; d++) {
              h[d + 3] = tap(d, 0);
              vv[d + 3] = tap(0, d);
            }
            v.y = ppg_green(h, vv);
          }
          v = mk3(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f));
        }
      }
      pr[pln_at(r, c)] = v.x;
      pg[pln_at(r, c)] = v.y;
      pb[pln_at(r, c)] = v.z;
    }
  }
  __syncthreads();

  const int lx = (threadIdx.x & 15) * 4;
  {
    // waves 0-3 take the even rows, waves 4-7 the odd rows: pixel k of every lane of a wave is the same site class
    const int ly = 2 * ((threadIdx.x >> 4) & 15) + (threadIdx.x >> 8);
    const int x = x0 + lx, y = y0 + ly;
    if (!INTERIOR && (x >= width || y >= height)) return;
    // this wave's rows all have the parity of y: the site class of pixel k is a scalar
    const int yu = __builtin_amdgcn_readfirstlane(y);
    float px[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int gx = x + k;
      // plane column lx + k + 1 + dx with lx a multiple of 4: quarter and offset inside it are compile-time per (k, dx)
      const int rowb = (ly + 1) * GS + (lx >> 2);
      auto at = [&](int dx, int dy) { const int kk = k + 1 + dx; return rowb + dy * GS + (kk >> 2) + (kk & 3) * GQ; };
      const int base = at(0, 0);
      f3 col = mk3(pr[base], pg[base], pb[base]);
      if (INTERIOR || (gx < width && !(gx == 0 || y == 0 || gx == width - 1 || y == height - 1))) {
        auto nb = [&](int dx, int dy) { const int q = at(dx, dy); return mk3(pr[q], pg[q], pb[q]); };
        col = ppg_redblue(nb, col, cfa_color(yu, k, pattern), cfa_color(yu, k + 1, pattern) == 0);  // x0 + lx is a multiple of 4
      }
      px[3 * k] = fmaxf(col.x, 0.0f);
      px[3 * k + 1] = fmaxf(col.y, 0.0f);
      px[3 * k + 2] = fmaxf(col.z, 0.0f);
    }
    store_rgb4(out, x, y, width, vec_ok, px);
  }
}

template <typename T>
__global__ __launch_bounds__(PNT) void ppg_fused(const T* __restrict__ src, const T* __restrict__ orig, T* __restrict__ out, int width,
                                                 int height, uint32_t pattern, int vec_ok) {
  __shared__ float raw[RHT * RS];
  __shared__ float pr[GH * GS], pg[GH * GS], pb[GH * GS];
  static_assert((RHT * RS + 3 * GH * GS) * sizeof(float) <= 40 * 1024, "four workgroups per CU");
  // Tile order: the frame's outer ring of tiles FIRST, then the interior row by row.  Border tiles take the slower variant
  // (in-image tests, 3x3 border averages); in plain row-major order the whole last tile row -- all border tiles -- is the end
  // of the launch and runs on a nearly empty GPU (a size-independent tail of ~45 us at 12 and 50 MP in round 4's numbers).
  const int ntx = (width + TW - 1) / TW, nty = (height + TH - 1) / TH;
  int tx, ty;
  {
    const int b = (int)blockIdx.x;
    const int nring = (ntx > 2 && nty > 2) ? 2 * ntx + 2 * (nty - 2) : ntx * nty;
    if (b >= nring) { const int i = b - nring; ty = 1 + i / (ntx - 2); tx = 1 + i - (ty - 1) * (ntx - 2); }
    else if (ntx <= 2 || nty <= 2) { ty = b / ntx; tx = b - ty * ntx; }
    else if (b < ntx) { ty = 0; tx = b; }
    else if (b < 2 * ntx) { ty = nty - 1; tx = b - ntx; }
    else { const int i = b - 2 * ntx; ty = 1 + (i >> 1); tx = (i & 1) ? ntx - 1 : 0; }
  }
  const int x0 = tx * TW, y0 = ty * TH;
  const bool interior = x0 >= 8 && y0 >= 8 && x0 + TW + 8 <= width && y0 + TH + 8 <= height;
  if (interior) ppg_tile<T, true>(src, orig, out, width, height, pattern, vec_ok, raw, pr, pg, pb, tx, ty);
  else ppg_tile<T, false>(src, orig, out, width, height, pattern, vec_ok, raw, pr, pg, pb, tx, ty);
}

// reference ppg.cu:21-113; threshold already divided by 100
template <typename T>
__global__ __launch_bounds__(256) void pre_median_kernel(const T* __restrict__ in, T* __restrict__ out, int width, int height,
                                                         uint32_t pattern, float threshold) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= width || y >= height) return;
  auto rd = [&](int xx, int yy) { return (xx >= 0 && yy >= 0 && xx < width && yy < height) ? ld(in, (size_t)yy * width + xx) : 0.0f; };
  const float center = rd(x, y);
  float med[9];
  int cnt = 0;
  {
    constexpr int DX[9] = {0, -1, 1, -2, 0, 2, -1, 1, 0};
    constexpr int DY[9] = {-2, -1, -1, 0, 0, 0, 1, 1, 2};
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const float v = rd(x + DX[k], y + DY[k]);
      if (fabsf(v - center) < threshold) { med[k] = v; cnt++; }
      else med[k] = 64.0f + v;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int ii = i + 1; ii < 9; ii++)
      if (med[i] > med[ii]) { const float t = med[i]; med[i] = med[ii]; med[ii] = t; }
  float color = center;
  if (cfa_color(y, x, pattern) & 1) {
    // med[(cnt - 1) / 2] with a register-resident array: select instead of dynamic indexing
    float pick = med[0];
    const int idx = (cnt - 1) / 2;
#pragma unroll
    for (int k = 1; k < 9; k++) pick = (idx == k) ? med[k] : pick;
    const float target = (cnt == 1) ? (med[4] - 64.0f) : pick;
    const float delta = target - center;
    color = center + fminf(fmaxf(delta, -threshold), threshold);
  }
  st(out, (size_t)y * width + x, fmaxf(color, 0.0f));
}

template <typename T>
int launch(const void* bayer, void* rgb, void* workspace, int width, int height, uint32_t pattern, float median_threshold, hipStream_t s) {
  const T* in = reinterpret_cast<const T*>(bayer);
  const T* src = in;
  if (median_threshold > 0.0f) {
    T* med = reinterpret_cast<T*>(workspace);
    TDK_LAUNCH("tdk_ppg(pre_median)", pre_median_kernel<T>, dim3(tdk_div_up(width, 64), tdk_div_up(height, 4)), dim3(256), 0, s, in, med, width, height,
                       pattern, median_threshold / 100.0f);
    src = med;
  }
  const int vec_ok = (width % 4 == 0) && tdk_aligned(rgb, 16);
  TDK_LAUNCH("tdk_ppg", ppg_fused<T>, dim3(tdk_div_up(width, TW) * tdk_div_up(height, TH)), dim3(PNT), 0, s, src, in, reinterpret_cast<T*>(rgb),
                     width, height, pattern, vec_ok);
  return TDK_OK;
}

}  // namespace

TDK_EXPORT size_t tdk_ppg_workspace_bytes(int width, int height, float median_threshold) {
  if (!(median_threshold > 0.0f) || width <= 0 || height <= 0) return 0;
  return tdk_align_up((size_t)width * height * sizeof(float), 256);
}

TDK_EXPORT int tdk_ppg(const void* bayer, void* rgb, void* workspace, int width, int height, uint32_t pattern, float median_threshold,
                       int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(bayer && rgb, "tdk_ppg: null pointer");
  TDK_REQUIRE(width > 0 && height > 0, "tdk_ppg: invalid size %dx%d", width, height);
  TDK_REQUIRE(pattern == TDK_PATTERN_RGGB || pattern == TDK_PATTERN_BGGR || pattern == TDK_PATTERN_GRBG || pattern == TDK_PATTERN_GBRG,
              "tdk_ppg: invalid Bayer pattern 0x%08x", pattern);
  TDK_REQUIRE(!(median_threshold > 0.0f) || workspace, "tdk_ppg: median_threshold > 0 needs a workspace");
  TDK_DISPATCH_DTYPE(dtype, T, return launch<T>(bayer, rgb, workspace, width, height, pattern, median_threshold, tdk_stream(stream)));
  return TDK_OK;
}
