// bilinear.hip -- 13-tap "5x5 diamond" linear demosaic.
//
// Replaces reference csrc/debayer/bilinear.cu:64-148 (bilinear5x5_demosaic).  Semantics kept:
// taps enumerated column by column over |dx|+|dy| <= 2, edge-clamped reads, one integer weight
// set per site class and output channel (each summing to 16), accumulation in tap order, site
// class of each 2x2 position per pattern from csrc/debayer/bayer_device.h:19-22 (including
// that table's BGGR / GBRG green-site assignment).
//
// MI355X design: 256-thread workgroup, 128 x 16 pixel tile staged once through LDS with a 2-px
// clamped halo (coalesced row reads); every thread produces a 4 x 2 pixel block so each output
// row segment leaves as three 16-B stores (48 contiguous bytes per thread, 1.5 KiB per wave
// row).  The kernel is instantiated per pattern, so all 13 x 3 weights are immediates and the
// zero taps vanish.  Bit-exact against the oracle: same term order, no FMA, /16 == *0.0625.
#include "tdk_common.h"

namespace {

constexpr int NTAP = 13;
constexpr int TAP_DX[NTAP] = {-2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2};
constexpr int TAP_DY[NTAP] = {0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0};

// Filter rows in tap order (Malvar-He-Cutler, scaled by 16).
struct Taps { int w[NTAP]; };
constexpr Taps K_IDENT = {{0, 0, 0, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0}};
constexpr Taps K_G_AT_RB = {{-2, 0, 4, 0, -2, 4, 8, 4, -2, 0, 4, 0, -2}};
constexpr Taps K_RB_AT_BR = {{-3, 4, 0, 4, -3, 0, 12, 0, -3, 4, 0, 4, -3}};
constexpr Taps K_C_AT_G_H = {{-2, -2, 8, -2, 1, 0, 10, 0, 1, -2, 8, -2, -2}};
constexpr Taps K_C_AT_G_V = {{1, -2, 0, -2, -2, 8, 10, 8, -2, -2, 0, -2, 1}};

// site class -> (R, G, B) filters; classes 0 = R, 1 = G (red row), 2 = G (blue row), 3 = B
template <int CLS, int CH> constexpr Taps filter_of() {
  if constexpr (CLS == 0) return CH == 0 ? K_IDENT : (CH == 1 ? K_G_AT_RB : K_RB_AT_BR);
  else if constexpr (CLS == 1) return CH == 0 ? K_C_AT_G_H : (CH == 1 ? K_IDENT : K_C_AT_G_V);
  else if constexpr (CLS == 2) return CH == 0 ? K_C_AT_G_V : (CH == 1 ? K_IDENT : K_C_AT_G_H);
  else return CH == 0 ? K_RB_AT_BR : (CH == 1 ? K_G_AT_RB : K_IDENT);
}

// site class at 2x2 position (y&1)*2 + (x&1), per pattern index (0 RGGB, 1 BGGR, 2 GRBG, 3 GBRG)
template <int PAT, int POS> constexpr int site_class() {
  constexpr int tab[4][4] = {{0, 1, 2, 3}, {3, 1, 2, 0}, {1, 0, 3, 2}, {1, 3, 0, 2}};
  return tab[PAT][POS];
}

constexpr int TW = 128, TH = 16, HALO = 2;
constexpr int LW = TW + 2 * HALO;      // 132
constexpr int LH = TH + 2 * HALO;      // 20
constexpr int LSTRIDE = LW + 1;        // odd stride keeps the two wave halves on different banks

template <int CLS, int CH> __device__ __forceinline__ float apply(const float v[NTAP]) {
  constexpr Taps f = filter_of<CLS, CH>();
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < NTAP; k++)
    if (f.w[k] != 0) acc += (float)f.w[k] * v[k];
  return acc * 0.0625f;
}

template <int CLS> __device__ __forceinline__ void pixel(const float* t, int lx, int ly, float out[3]) {
  float v[NTAP];
#pragma unroll
  for (int k = 0; k < NTAP; k++) v[k] = t[(ly + TAP_DY[k]) * LSTRIDE + lx + TAP_DX[k]];
  out[0] = apply<CLS, 0>(v);
  out[1] = apply<CLS, 1>(v);
  out[2] = apply<CLS, 2>(v);
}

template <typename T, int PAT>
__global__ __launch_bounds__(256) void bilinear_kernel(const T* __restrict__ in, T* __restrict__ out, int width, int height,
                                                       int vec_ok) {
  __shared__ float tile[LH * LSTRIDE];
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  for (int i = threadIdx.x; i < LW * LH; i += 256) {
    const int r = i / LW, c = i - r * LW;
    const int gx = min(max(tx0 - HALO + c, 0), width - 1);
    const int gy = min(max(ty0 - HALO + r, 0), height - 1);
    tile[r * LSTRIDE + c] = ld(in, (size_t)gy * width + gx);
  }
  __syncthreads();

  const int lx = (threadIdx.x & 31) * 4, ly = (threadIdx.x >> 5) * 2;
  const int x = tx0 + lx, y = ty0 + ly;
  if (x >= width) return;
#pragma unroll
  for (int r = 0; r < 2; r++) {
    if (y + r >= height) break;
    float px[12];
    const float* t = tile + HALO * LSTRIDE + HALO;
    if (r == 0) {
      pixel<site_class<PAT, 0>()>(t, lx + 0, ly, px + 0);
      pixel<site_class<PAT, 1>()>(t, lx + 1, ly, px + 3);
      pixel<site_class<PAT, 0>()>(t, lx + 2, ly, px + 6);
      pixel<site_class<PAT, 1>()>(t, lx + 3, ly, px + 9);
    } else {
      pixel<site_class<PAT, 2>()>(t, lx + 0, ly + 1, px + 0);
      pixel<site_class<PAT, 3>()>(t, lx + 1, ly + 1, px + 3);
      pixel<site_class<PAT, 2>()>(t, lx + 2, ly + 1, px + 6);
      pixel<site_class<PAT, 3>()>(t, lx + 3, ly + 1, px + 9);
    }
    const size_t p = (size_t)(y + r) * width + x;
    if (vec_ok) {
      rgb4_io<T>::store(out, p >> 2, px);
    } else {
      for (int k = 0; k < 4 && x + k < width; k++) {
        st(out, (p + k) * 3 + 0, px[3 * k]);
        st(out, (p + k) * 3 + 1, px[3 * k + 1]);
        st(out, (p + k) * 3 + 2, px[3 * k + 2]);
      }
    }
  }
}

template <typename T>
int launch(const void* bayer, void* rgb, int width, int height, uint32_t pattern, hipStream_t s) {
  const dim3 grid(tdk_div_up(width, TW), tdk_div_up(height, TH)), block(256);
  const T* in = reinterpret_cast<const T*>(bayer);
  T* out = reinterpret_cast<T*>(rgb);
  const int vec_ok = (width % 4 == 0) && tdk_aligned(rgb, 16);
  switch (pattern) {
    case TDK_PATTERN_RGGB: TDK_LAUNCH("tdk_bilinear5x5", (bilinear_kernel<T, 0>), grid, block, 0, s, in, out, width, height, vec_ok); break;
    case TDK_PATTERN_BGGR: TDK_LAUNCH("tdk_bilinear5x5", (bilinear_kernel<T, 1>), grid, block, 0, s, in, out, width, height, vec_ok); break;
    case TDK_PATTERN_GRBG: TDK_LAUNCH("tdk_bilinear5x5", (bilinear_kernel<T, 2>), grid, block, 0, s, in, out, width, height, vec_ok); break;
    case TDK_PATTERN_GBRG: TDK_LAUNCH("tdk_bilinear5x5", (bilinear_kernel<T, 3>), grid, block, 0, s, in, out, width, height, vec_ok); break;
    default: tdk_set_error("tdk_bilinear5x5: invalid Bayer pattern 0x%08x", pattern); return TDK_ERR_INVALID_ARGUMENT;
  }
  return TDK_OK;
}

}  // namespace

TDK_EXPORT int tdk_bilinear5x5(const void* bayer, void* rgb, int width, int height, uint32_t pattern, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(bayer && rgb, "tdk_bilinear5x5: null pointer");
  TDK_REQUIRE(width > 0 && height > 0, "tdk_bilinear5x5: invalid size %dx%d", width, height);
  TDK_DISPATCH_DTYPE(dtype, T, return launch<T>(bayer, rgb, width, height, pattern, tdk_stream(stream)));
  return TDK_OK;
}
