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
// MI355X design (first version): the reference's kernel sequence is kept, with
//  * the six gamma pointer tables passed by value as kernel arguments (the reference uploads
//    them to process-global __device__ symbols before every level, laplacian.cu:43-45,574-575:
//    not stream- or multi-instance safe);
//  * the six level-0 remap curves fused into one kernel that reads the padded input once and
//    writes the six processed images (the reference launches six full-resolution passes);
//  * the six per-gamma reduces of a level fused into one launch (blockIdx.z = gamma);
//  * workspace handed in by the caller (no allocation inside process()).
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

// laplacian.cu:266-290
__device__ __forceinline__ float curve(float x, float g, float sigma, float shadows, float highlights, float clarity) {
  const float c = x - g;
  float val;
  const float ssigma = c > 0.0f ? sigma : -sigma;
  const float shadhi = c > 0.0f ? shadows : highlights;
  if (fabsf(c) > 2 * sigma) {
    val = g + ssigma + shadhi * (c - ssigma);
  } else {
    const float t = clip01(c / (2.0f * ssigma));
    const float t2 = t * t;
    const float mt = 1.0f - t;
    val = g + ssigma * 2.0f * mt * t + t2 * (ssigma + ssigma * shadhi);
  }
  const float exp_arg = -c * c / (2.0f * sigma * sigma / 3.0f);
  val += clarity * c * expf(exp_arg);
  return val;
}

__global__ __launch_bounds__(256) void curves_kernel(const __half* __restrict__ padded, Ptr6 outs, int64_t n, float sigma, float shadows,
                                                     float highlights, float clarity) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = __half2float(padded[i]);
#pragma unroll
    for (int k = 0; k < NG; k++) outs.p[k][i] = __float2half_rn(curve(v, ((float)k + 0.5f) / (float)NG, sigma, shadows, highlights, clarity));
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

// laplacian.cu:221-252
__global__ __launch_bounds__(256) void assemble_kernel(const __half* __restrict__ input, const __half* __restrict__ out_coarse,
                                                       __half* __restrict__ out_fine, CPtr6 g_fine, CPtr6 g_coarse, int fw, int fh) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= fw || y >= fh) return;
  int qx = x, qy = y;  // clamp_boundary (laplacian.cu:53-65)
  if (fw & 1) { if (qx > fw - 2) qx = fw - 2; } else { if (qx > fw - 3) qx = fw - 3; }
  if (fh & 1) { if (qy > fh - 2) qy = fh - 2; } else { if (qy > fh - 3) qy = fh - 3; }
  if (qx <= 0) qx = 1;
  if (qy <= 0) qy = 1;
  const int cw = (fw - 1) / 2 + 1;
  float val = expand_gaussian(out_coarse, qx, qy, cw);
  const float v = hld(input, x, y, fw);
  int hi = 1;
  for (; hi < NG - 1 && ((float)hi + .5f) / (float)NG <= v; hi++) {}
  const int lo = hi - 1;
  const float a = fminf(fmaxf(v * NG - ((float)lo + .5f), 0.0f), 1.0f);
  // select the two bracketing gamma pyramids without dynamic indexing of the argument struct
  const __half *f0 = g_fine.p[0], *c0 = g_coarse.p[0], *f1 = g_fine.p[1], *c1 = g_coarse.p[1];
#pragma unroll
  for (int k = 1; k < NG - 1; k++)
    if (lo == k) { f0 = g_fine.p[k]; c0 = g_coarse.p[k]; f1 = g_fine.p[k + 1]; c1 = g_coarse.p[k + 1]; }
  const float l0 = hld(f0, x, y, fw) - expand_gaussian(c0, qx, qy, cw);
  const float l1 = hld(f1, x, y, fw) - expand_gaussian(c1, qx, qy, cw);
  val += l0 * (1.0f - a) + l1 * a;
  hst(out_fine, x, y, fw, val);
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

  {
    Ptr6 outs;
    for (int k = 0; k < NG; k++) outs.p[k] = proc(k, 0);
    const int64_t n = (int64_t)L.bw * L.bh;
    int64_t blocks = tdk_div_up64(n, 256);
    if (blocks > 4096) blocks = 4096;
    TDK_LAUNCH("tdk_laplacian(curves)", curves_kernel, dim3((unsigned)blocks), dim3(256), 0, s, padded(0), outs, n, sigma, shadows, highlights, clarity);
  }
  for (int l = 1; l < L.levels; l++) {
    const int cw = dl(L.bw, l), ch = dl(L.bh, l), fw = dl(L.bw, l - 1);
    CPtr6 f; Ptr6 c;
    for (int k = 0; k < NG; k++) { f.p[k] = proc(k, l - 1); c.p[k] = proc(k, l); }
    dim3 g = grid2(cw, ch);
    g.z = NG;
    TDK_LAUNCH("tdk_laplacian(reduce6)", reduce_kernel, g, dim3(256), 0, s, f, c, fw, cw, ch);
  }

  for (int l = L.levels - 2; l >= 0; l--) {
    const int pw = dl(L.bw, l), ph = dl(L.bh, l);
    CPtr6 gf, gc;
    for (int k = 0; k < NG; k++) { gf.p[k] = proc(k, l); gc.p[k] = proc(k, l + 1); }
    TDK_LAUNCH("tdk_laplacian(assemble)", assemble_kernel, grid2(pw, ph), dim3(256), 0, s, padded(l), output(l + 1), output(l), gf, gc, pw, ph);
  }

  TDK_LAUNCH("tdk_laplacian(write_back)", write_back_kernel, grid2(width, height), dim3(256), 0, s, output(0), lum_out, width, height, L.pad, L.bw);
  return TDK_OK;
}
