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
// MI355X design:
//  * the six gamma pointer tables are passed by value as kernel arguments (the reference uploads
//    them to process-global __device__ symbols before every level, laplacian.cu:43-45,574-575:
//    not stream- or multi-instance safe);
//  * level 0 of the six gamma pyramids (six full-resolution curve images: 6 x 63 MB written and
//    read back twice at 12 MP) is never materialised: curves_reduce6_kernel evaluates the curves
//    into LDS and reduces them to level 1 in the same pass, and the level-0 assemble recomputes
//    binary16(curve(input)) for the two gammas that bracket each pixel;
//  * assemble is tiled: the <= 35 x 11 coarse cells a 64 x 16 fine tile expands from are staged in
//    LDS for the output pyramid and all six gamma pyramids (the reference gathers 3 x 4..9 halves
//    per pixel from global memory);
//  * the six per-gamma reduces of the deeper levels run in one launch (blockIdx.z = gamma);
//  * workspace handed in by the caller (no allocation inside process()).
// Every value that the reference stores is still rounded to binary16 at the same point.
#include "tdk_common.h"

namespace {

constexpr int NG = 6;
constexpr int MAX_LEVELS = 30;

inline int dl(int x, int level) { return (x + (1 << level) - 1) >> level; }

__device__ __forceinline__ float hld(const __half* p, int x, int y, int w) { return __half2float(p[(size_t)y * w + x]); }
__device__ __forceinline__ void hst(__half* p, int x, int y, int w, float v) { p[(size_t)y * w + x] = __float2half_rn(v); }

struct Ptr6 {
  __half* p[NG];
};
struct CPtr6 {
  const __half* p[NG];
};

__global__ __launch_bounds__(256) void pad_kernel(const float* __restrict__ in, __half* __restrict__ padded, int w, int h, int pad, int bw, int bh) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= bw || y >= bh) return;
  const int cx = min(max(x - pad, 0), w - 1), cy = min(max(y - pad, 0), h - 1);
  hst(padded, x, y, bw, in[(size_t)cy * w + cx]);
}

__device__ __forceinline__ float reduce_at(const __half* __restrict__ fine, int fw, int px, int py, int cw, int ch) {
  int cx = px, cy = py;
  if (px >= cw - 1) cx = cw - 2;
  if (py >= ch - 1) cy = ch - 2;
  if (cx <= 0) cx = 1;
  if (cy <= 0) cy = 1;
  const float w5[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
  float acc = 0.0f;
#pragma unroll
  for (int j = -2; j <= 2; j++)
#pragma unroll
    for (int i = -2; i <= 2; i++) acc += hld(fine, 2 * cx + i, 2 * cy + j, fw) * w5[i + 2] * w5[j + 2];
  return acc;
}

// laplacian.cu:177-207; blockIdx.z selects one of up to six (fine, coarse) pairs
__global__ __launch_bounds__(256) void reduce_kernel(CPtr6 fine, Ptr6 coarse, int fw, int cw, int ch) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cw || y >= ch) return;
  hst(coarse.p[blockIdx.z], x, y, cw, reduce_at(fine.p[blockIdx.z], fw, x, y, cw, ch));
}

// laplacian.cu:266-290.  Per-launch constants are hoisted into CurveK (reciprocals instead of the
// per-sample divisions: within an ulp of the reference's fast-math divides, far below the binary16
// rounding every result goes through); both pieces are evaluated and selected (they are a few
// instructions each).
struct CurveK {
  float sigma, two_sigma, inv_two_sigma, shadows, highlights, clarity, neg_inv_e;  // neg_inv_e = -log2(e) / (2 sigma^2 / 3)
};
__device__ __forceinline__ CurveK make_curve(float sigma, float shadows, float highlights, float clarity) {
  CurveK k;
  k.sigma = sigma; k.two_sigma = 2 * sigma; k.inv_two_sigma = 1.0f / (2.0f * sigma);
  k.shadows = shadows; k.highlights = highlights; k.clarity = clarity;
  k.neg_inv_e = -1.44269504088896341f / (2.0f * sigma * sigma / 3.0f);
  return k;
}
__device__ __forceinline__ float curve(float x, float g, const CurveK& k) {
  const float c = x - g;
  const bool pos = c > 0.0f;
  const float ssigma = pos ? k.sigma : -k.sigma;
  const float shadhi = pos ? k.shadows : k.highlights;
  const float outer = g + ssigma + shadhi * (c - ssigma);
  const float t = fminf(fabsf(c) * k.inv_two_sigma, 1.0f);  // c / (2 ssigma) is never negative
  const float inner = g + ssigma * 2.0f * (1.0f - t) * t + t * t * (ssigma + ssigma * shadhi);
  float val = (fabsf(c) > k.two_sigma) ? outer : inner;
  val += k.clarity * c * __builtin_amdgcn_exp2f(c * c * k.neg_inv_e);  // hardware exp2 (the reference is a --use_fast_math build)
  return val;
}

// Level 0 -> level 1 of the six gamma pyramids in one pass: the padded input is read once, the six
// remap curves are evaluated per fine pixel into LDS (rounded to binary16, exactly what the
// reference stores as level 0 of each gamma pyramid), then reduced separably (rows, then columns).  The six full-resolution curve images
// (6 x 63 MB written and read back at 12 MP) never exist.
constexpr int RTW = 32, RTH = 8, RFW = 2 * RTW + 3, RFH = 2 * RTH + 3, RFS = RFW + 1;

__global__ __launch_bounds__(256) void curves_reduce6_kernel(const __half* __restrict__ padded, Ptr6 coarse, int fw, int cw, int ch, float sigma,
                                                             float shadows, float highlights, float clarity) {
  __shared__ __half fine[NG][RFH * RFS];
  __shared__ float hrow[NG][RFH * RTW];
  const CurveK ck = make_curve(sigma, shadows, highlights, clarity);
  const int CX0 = blockIdx.x * RTW, CY0 = blockIdx.y * RTH;
  auto clampc = [](int p, int n) { int c = p; if (p >= n - 1) c = n - 2; if (c <= 0) c = 1; return c; };  // reduce_at's centre clamp
  const int cxa = clampc(CX0, cw), cxb = clampc(min(CX0 + RTW, cw) - 1, cw);
  const int cya = clampc(CY0, ch), cyb = clampc(min(CY0 + RTH, ch) - 1, ch);
  const int fx0 = 2 * cxa - 2, fy0 = 2 * cya - 2;
  const int fx1 = 2 * cxb + 2, fy1 = 2 * cyb + 2;  // last fine column / row any tap of this tile reads
  for (int i = threadIdx.x; i < RFW * RFH; i += 256) {  // full window, clamped: constant divisor
    const int r = i / RFW, c = i - r * RFW;
    const float v = hld(padded, min(fx0 + c, fx1), min(fy0 + r, fy1), fw);
#pragma unroll
    for (int k = 0; k < NG; k++) fine[k][r * RFS + c] = __float2half_rn(curve(v, ((float)k + 0.5f) / (float)NG, ck));
  }
  __syncthreads();
  // separable 5 x 5: horizontal sums of every fine row at the tile's 32 coarse columns, then the
  // vertical combination (9 + 9 instead of 25 multiply-adds per output; the summation order differs
  // from reduce_at's -- far below the binary16 rounding of the result)
  const float w5[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
  for (int i = threadIdx.x; i < NG * RFH * RTW; i += 256) {
    const int pxl = i & (RTW - 1), kr = i / RTW, k = kr / RFH, r = kr - k * RFH;
    const int px = min(CX0 + pxl, cw - 1);
    const __half* row = &fine[k][r * RFS + 2 * clampc(px, cw) - fx0];
    float acc = 0.0f;
#pragma unroll
    for (int t = -2; t <= 2; t++) acc += __half2float(row[t]) * w5[t + 2];
    hrow[k][r * RTW + pxl] = acc;
  }
  __syncthreads();
  const int pxl = threadIdx.x & (RTW - 1);
  const int px = CX0 + pxl, py = CY0 + threadIdx.x / RTW;
  if (px >= cw || py >= ch) return;
  const int ly = 2 * clampc(py, ch) - fy0;
#pragma unroll
  for (int k = 0; k < NG; k++) {
    float acc = 0.0f;
#pragma unroll
    for (int j = -2; j <= 2; j++) acc += hrow[k][(ly + j) * RTW + pxl] * w5[j + 2];
    hst(coarse.p[k], px, py, cw, acc);
  }
}

// laplacian.cu:111-141
__device__ __forceinline__ float expand_gaussian(const __half* __restrict__ coarse, int x, int y, int cw) {
  const float w5[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};
  const int cx = x / 2, cy = y / 2;
  const int x_odd = x & 1, y_odd = y & 1;
  float c = 0.0f;
  for (int i = x_odd ? 0 : -1; i <= 1; i++)
    for (int j = y_odd ? 0 : -1; j <= 1; j++) {
      const float p = hld(coarse, cx + i, cy + j, cw);
      const int wi = x_odd ? (2 * i + 1) : (2 * i + 2);
      const int wj = y_odd ? (2 * j + 1) : (2 * j + 2);
      c += p * w5[wi] * w5[wj];
    }
  return 4.0f * c;
}

// laplacian.cu:53-65
__host__ __device__ __forceinline__ int clamp_boundary(int q, int n) {
  if (n & 1) { if (q > n - 2) q = n - 2; } else { if (q > n - 3) q = n - 3; }
  if (q <= 0) q = 1;
  return q;
}

// expand_gaussian (above) reading a float tile in LDS: tile(x, y) = coarse(x + tx0, y + ty0).
// Same taps, weights and summation order; the parity-dependent tap set is a predicate instead of a
// loop bound (an even coordinate uses taps -1, 0, 1 with weights 1 6 1; an odd one taps 0, 1 with 4 4).
__device__ __forceinline__ float expand_lds(const float* __restrict__ tile, int x, int y, int tx0, int ty0, int ts) {
  const int cx = x / 2 - tx0, cy = y / 2 - ty0;
  const bool x_odd = x & 1, y_odd = y & 1;
  const float wx[3] = {1.0f / 16.0f, x_odd ? 4.0f / 16.0f : 6.0f / 16.0f, x_odd ? 4.0f / 16.0f : 1.0f / 16.0f};
  const float wy[3] = {1.0f / 16.0f, y_odd ? 4.0f / 16.0f : 6.0f / 16.0f, y_odd ? 4.0f / 16.0f : 1.0f / 16.0f};
  float c = 0.0f;
#pragma unroll
  for (int i = -1; i <= 1; i++)
#pragma unroll
    for (int j = -1; j <= 1; j++) {
      const bool take = !(x_odd && i == -1) && !(y_odd && j == -1);
      const float p = tile[max(cy + j, 0) * ts + max(cx + i, 0)];  // clamped index only matters for taps not taken
      const float t = c + p * wx[i + 1] * wy[j + 1];
      c = take ? t : c;
    }
  return 4.0f * c;
}

// laplacian.cu:221-252, tiled: a 256-thread workgroup owns 64 x 16 fine pixels; the coarse cells
// their expands touch (<= 35 x 11) of the output pyramid and of all six gamma pyramids are staged
// once in LDS as floats (the reference gathers 3 x 4..9 halves per pixel from global memory).
// LEVEL0: the fine level of the gamma pyramids is never stored -- it is binary16(curve(input)),
// recomputed here for the two bracketing gammas.
constexpr int ATW = 64, ATH = 16, ACW = ATW / 2 + 3, ACH = ATH / 2 + 3, ACS = ACW + 1;

template <bool LEVEL0>
__global__ __launch_bounds__(256) void assemble_tiled_kernel(const __half* __restrict__ input, const __half* __restrict__ out_coarse,
                                                             __half* __restrict__ out_fine, CPtr6 g_fine, CPtr6 g_coarse, int fw, int fh, float sigma,
                                                             float shadows, float highlights, float clarity, int tile_x0, int tile_y0) {
  __shared__ float tiles[(NG + 1) * ACH * ACS];
  const CurveK ck = make_curve(sigma, shadows, highlights, clarity);
  const int X0 = (blockIdx.x + tile_x0) * ATW, Y0 = (blockIdx.y + tile_y0) * ATH;
  const int cw = (fw - 1) / 2 + 1, chh = (fh - 1) / 2 + 1;
  const int X1 = min(X0 + ATW, fw) - 1, Y1 = min(Y0 + ATH, fh) - 1;
  // coarse cells touched: clamp_boundary is monotone, taps are cx - 1 .. cx + 1
  const int tx0 = max(clamp_boundary(X0, fw) / 2 - 1, 0), tx1 = min(clamp_boundary(X1, fw) / 2 + 1, cw - 1);
  const int ty0 = max(clamp_boundary(Y0, fh) / 2 - 1, 0), ty1 = min(clamp_boundary(Y1, fh) / 2 + 1, chh - 1);
  // the full ACW x ACH window is loaded (coordinates clamped into the level): constant trip counts and divisors
#pragma unroll
  for (int k = 0; k <= NG; k++) {
    const __half* src = (k == 0) ? out_coarse : g_coarse.p[k - 1];
    for (int i = threadIdx.x; i < ACW * ACH; i += 256) {
      const int r = i / ACW, c = i - r * ACW;
      tiles[k * (ACH * ACS) + r * ACS + c] = hld(src, min(tx0 + c, tx1), min(ty0 + r, ty1), cw);
    }
  }
  __syncthreads();
  const int x = X0 + (threadIdx.x & 63);
  if (x >= fw) return;
  const int qx = clamp_boundary(x, fw);
  for (int yy = threadIdx.x >> 6; yy < ATH; yy += 4) {
    const int y = Y0 + yy;
    if (y >= fh) break;
    const int qy = clamp_boundary(y, fh);
    float val = expand_lds(tiles, qx, qy, tx0, ty0, ACS);
    const float v = hld(input, x, y, fw);
    // the reference's search loop over the (increasing) gamma centres, as four compares
    int hi = 1;
#pragma unroll
    for (int q = 1; q < NG - 1; q++) hi += (((float)q + .5f) / (float)NG <= v) ? 1 : 0;
    const int lo = hi - 1;
    const float a = fminf(fmaxf(v * NG - ((float)lo + .5f), 0.0f), 1.0f);
    float fine0, fine1;
    if constexpr (LEVEL0) {
      fine0 = __half2float(__float2half_rn(curve(v, ((float)lo + 0.5f) / (float)NG, ck)));
      fine1 = __half2float(__float2half_rn(curve(v, ((float)lo + 1.5f) / (float)NG, ck)));
    } else {
      const __half *f0 = g_fine.p[0], *f1 = g_fine.p[1];
#pragma unroll
      for (int k = 1; k < NG - 1; k++)
        if (lo == k) { f0 = g_fine.p[k]; f1 = g_fine.p[k + 1]; }
      fine0 = hld(f0, x, y, fw);
      fine1 = hld(f1, x, y, fw);
    }
    const float l0 = fine0 - expand_lds(tiles + (1 + lo) * (ACH * ACS), qx, qy, tx0, ty0, ACS);
    const float l1 = fine1 - expand_lds(tiles + (2 + lo) * (ACH * ACS), qx, qy, tx0, ty0, ACS);
    val += l0 * (1.0f - a) + l1 * a;
    hst(out_fine, x, y, fw, val);
  }
}

__global__ __launch_bounds__(256) void write_back_kernel(const __half* __restrict__ processed, float* __restrict__ out, int w, int h, int pad, int bw) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  out[(size_t)y * w + x] = hld(processed, x + pad, y + pad, bw);
}

struct Layout {
  int levels, pad, bw, bh;
  size_t level_off[MAX_LEVELS + 1];  // element offset of level l inside one pyramid
  size_t pyr_elems;
};

Layout make_layout(int w, int h) {
  Layout L;
  const int m = w < h ? w : h;
  int lg = 0;
  while ((1 << (lg + 1)) <= m) lg++;
  L.levels = lg < MAX_LEVELS ? lg : MAX_LEVELS;
  L.pad = L.levels >= 1 ? 1 << (L.levels - 1) : 0;
  L.bw = w + 2 * L.pad;
  L.bh = h + 2 * L.pad;
  size_t off = 0;
  for (int l = 0; l < L.levels; l++) {
    L.level_off[l] = off;
    off += tdk_align_up((size_t)dl(L.bw, l) * dl(L.bh, l), 128);
  }
  L.level_off[L.levels] = off;
  L.pyr_elems = off;
  return L;
}

inline dim3 grid2(int w, int h) { return dim3(tdk_div_up(w, 64), tdk_div_up(h, 4)); }

}  // namespace

TDK_EXPORT size_t tdk_laplacian_workspace_bytes(int width, int height, int num_gamma) {
  if (width <= 0 || height <= 0 || num_gamma != NG) return 0;
  const Layout L = make_layout(width, height);
  return tdk_align_up((size_t)(2 + NG) * L.pyr_elems * sizeof(__half), 256);  // padded + output + 6 processed pyramids
}

TDK_EXPORT int tdk_laplacian(const float* lum_in, float* lum_out, void* workspace, int width, int height, int num_gamma, float sigma,
                             float shadows, float highlights, float clarity, tdk_stream_t stream) {
  TDK_REQUIRE(num_gamma == NG, "Unsupported gamma count: %d", num_gamma);
  TDK_REQUIRE(lum_in && lum_out && workspace, "tdk_laplacian: null pointer");
  TDK_REQUIRE(width >= 4 && height >= 4, "tdk_laplacian: image %dx%d too small", width, height);
  hipStream_t s = tdk_stream(stream);
  const Layout L = make_layout(width, height);
  __half* base = reinterpret_cast<__half*>(workspace);
  auto padded = [&](int l) { return base + L.level_off[l]; };
  auto output = [&](int l) { return base + L.pyr_elems + L.level_off[l]; };
  auto proc = [&](int k, int l) { return base + (size_t)(2 + k) * L.pyr_elems + L.level_off[l]; };

  TDK_LAUNCH("tdk_laplacian(pad)", pad_kernel, grid2(L.bw, L.bh), dim3(256), 0, s, lum_in, padded(0), width, height, L.pad, L.bw, L.bh);

  for (int l = 1; l < L.levels; l++) {
    const int cw = dl(L.bw, l), ch = dl(L.bh, l), fw = dl(L.bw, l - 1);
    CPtr6 f{}; Ptr6 c{};
    f.p[0] = padded(l - 1);
    c.p[0] = (l == L.levels - 1) ? output(l) : padded(l);
    dim3 g = grid2(cw, ch);
    TDK_LAUNCH("tdk_laplacian(reduce)", reduce_kernel, g, dim3(256), 0, s, f, c, fw, cw, ch);
  }

  if (L.levels >= 2) {  // gamma pyramids, level 0 -> 1 (level 0 itself is recomputed where it is needed)
    const int cw = dl(L.bw, 1), ch = dl(L.bh, 1);
    Ptr6 c;
    for (int k = 0; k < NG; k++) c.p[k] = proc(k, 1);
    TDK_LAUNCH("tdk_laplacian(curves+reduce6)", curves_reduce6_kernel, dim3(tdk_div_up(cw, RTW), tdk_div_up(ch, RTH)), dim3(256), 0, s, padded(0), c, L.bw, cw,
               ch, sigma, shadows, highlights, clarity);
  }
  for (int l = 2; l < L.levels; l++) {
    const int cw = dl(L.bw, l), ch = dl(L.bh, l), fw = dl(L.bw, l - 1);
    CPtr6 f; Ptr6 c;
    for (int k = 0; k < NG; k++) { f.p[k] = proc(k, l - 1); c.p[k] = proc(k, l); }
    dim3 g = grid2(cw, ch);
    g.z = NG;
    TDK_LAUNCH("tdk_laplacian(reduce6)", reduce_kernel, g, dim3(256), 0, s, f, c, fw, cw, ch);
  }

  // The output pyramid is only consumed downwards, and only write_back reads level 0 -- inside the
  // un-padded image.  So level l is assembled only on the rectangle the level below expands from
  // (clamp_boundary, taps cx - 1 .. cx + 1), starting from the image rectangle at level 0: most of
  // the 2.5x padded area is never assembled at the fine levels.
  struct Rect { int x0, x1, y0, y1; };  // inclusive
  Rect need[MAX_LEVELS + 1];
  need[0] = Rect{L.pad, L.pad + width - 1, L.pad, L.pad + height - 1};
  for (int l = 0; l + 1 < L.levels; l++) {
    const int fw = dl(L.bw, l), fh = dl(L.bh, l), cw = (fw - 1) / 2 + 1, chh = (fh - 1) / 2 + 1;
    auto lo = [](int v) { return v > 0 ? v : 0; };
    need[l + 1] = Rect{lo(clamp_boundary(need[l].x0, fw) / 2 - 1), (clamp_boundary(need[l].x1, fw) / 2 + 1 < cw - 1) ? clamp_boundary(need[l].x1, fw) / 2 + 1 : cw - 1,
                       lo(clamp_boundary(need[l].y0, fh) / 2 - 1), (clamp_boundary(need[l].y1, fh) / 2 + 1 < chh - 1) ? clamp_boundary(need[l].y1, fh) / 2 + 1 : chh - 1};
  }
  for (int l = L.levels - 2; l >= 0; l--) {
    const int pw = dl(L.bw, l), ph = dl(L.bh, l);
    CPtr6 gf, gc;
    for (int k = 0; k < NG; k++) { gf.p[k] = proc(k, l); gc.p[k] = proc(k, l + 1); }
    const int tx0 = need[l].x0 / ATW, tx1 = need[l].x1 / ATW, ty0 = need[l].y0 / ATH, ty1 = need[l].y1 / ATH;
    const dim3 g(tx1 - tx0 + 1, ty1 - ty0 + 1);
    if (l == 0)
      TDK_LAUNCH("tdk_laplacian(assemble)", assemble_tiled_kernel<true>, g, dim3(256), 0, s, padded(l), output(l + 1), output(l), gf, gc, pw, ph, sigma, shadows,
                 highlights, clarity, tx0, ty0);
    else
      TDK_LAUNCH("tdk_laplacian(assemble)", assemble_tiled_kernel<false>, g, dim3(256), 0, s, padded(l), output(l + 1), output(l), gf, gc, pw, ph, sigma, shadows,
                 highlights, clarity, tx0, ty0);
  }

  TDK_LAUNCH("tdk_laplacian(write_back)", write_back_kernel, grid2(width, height), dim3(256), 0, s, output(0), lum_out, width, height, L.pad, L.bw);
  return TDK_OK;
}
