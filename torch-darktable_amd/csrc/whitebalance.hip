// whitebalance.hip -- sample collection of estimate_white_balance.
//
// Replaces reference csrc/white_balance.cu:57-126 (collect_color_samples_kernel + collect_samples):
// one sample per cell of the (height / stride) x (width / stride) grid; a sample is the 2x2 CFA
// quad of the cell turned into RGB (bayer_device.h:35-43), chroma = (r, g) / (r + g + b),
// intensity = r + g + b, valid = max(quad) < 1.  The quantile / mean that follow stay torch
// ops on the device in the binding, as in the reference (white_balance.cu:149-161).
//
// Two reference slips are handled explicitly:
//  * the reference reads the quad at pos * 2 although the grid is sized by `stride` (:71), i.e. it
//    only looks at the top-left (2/stride)^2 of the frame: `literal_positions` != 0 reproduces that and is what the
//    binding passes by default (same gains as the reference); 0 reads cell (i, j) at (i * stride, j * stride), the
//    documented intent, as an opt-in correction;
//  * it leaves the skipped last row / column of cells uninitialised (torch::empty, :107-109):
//    here they are written as invalid (mask 0, values 0), so the result is a function of the input.
// Arithmetic is + * / max only: bit-exact against oracle/src/whitebalance.c.
#include "tdk_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void wb_collect_kernel(const T* __restrict__ bayer, int width, int sw, int sh, uint32_t pattern, int step,
                                                         float* __restrict__ chroma, float* __restrict__ intensity, uint8_t* __restrict__ mask) {
  for (int i = blockIdx.y; i < sh; i += gridDim.y)
    for (int j = blockIdx.x * 256 + threadIdx.x; j < sw; j += gridDim.x * 256) {
      const size_t n = (size_t)i * sw + j;
      float cr = 0.0f, cg = 0.0f, s = 0.0f;
      uint8_t ok = 0;
      if (j + 1 < sw && i + 1 < sh) {
        const size_t q = (size_t)(i * step) * width + j * step;
        const float p00 = ld<T>(bayer, q), p01 = ld<T>(bayer, q + 1), p10 = ld<T>(bayer, q + width), p11 = ld<T>(bayer, q + width + 1);
        float r, g, b;
        switch (pattern) {
          case TDK_PATTERN_RGGB: r = p00; g = (p01 + p10) * 0.5f; b = p11; break;
          case TDK_PATTERN_BGGR: r = p11; g = (p01 + p10) * 0.5f; b = p00; break;
          case TDK_PATTERN_GRBG: r = p01; g = (p00 + p11) * 0.5f; b = p10; break;
          default:               r = p10; g = (p00 + p11) * 0.5f; b = p01; break;
        }
        s = r + g + b;
        cr = r / s;
        cg = g / s;
        ok = fmaxf(fmaxf(p00, p01), fmaxf(p10, p11)) < 1.0f;
      }
      reinterpret_cast<float2*>(chroma)[n] = make_float2(cr, cg);
      intensity[n] = s;
      mask[n] = ok;
    }
}

}  // namespace

TDK_EXPORT int tdk_wb_collect_samples_ex(const void* bayer, int width, int height, uint32_t pattern, int stride, int literal_positions, float* chroma,
                                         float* intensity, uint8_t* mask, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(bayer && chroma && intensity && mask, "tdk_wb_collect_samples: null pointer");
  TDK_REQUIRE(stride >= 2, "tdk_wb_collect_samples: stride must be >= 2 (a sample is a 2x2 CFA quad), got %d", stride);
  TDK_REQUIRE(width >= stride && height >= stride, "tdk_wb_collect_samples: image %dx%d smaller than the stride %d", width, height, stride);
  TDK_REQUIRE(pattern == TDK_PATTERN_RGGB || pattern == TDK_PATTERN_BGGR || pattern == TDK_PATTERN_GRBG || pattern == TDK_PATTERN_GBRG,
              "tdk_wb_collect_samples: unknown Bayer pattern 0x%08x", pattern);
  TDK_REQUIRE(tdk_aligned(chroma, 8), "tdk_wb_collect_samples: chroma must be 8-byte aligned");
  const int sw = width / stride, sh = height / stride;
  const dim3 grid((unsigned)tdk_div_up(sw, 256), (unsigned)(sh < 32768 ? sh : 32768));
  TDK_DISPATCH_DTYPE(dtype, T, TDK_LAUNCH("tdk_wb_collect_samples", wb_collect_kernel<T>, grid, dim3(256), 0, tdk_stream(stream), reinterpret_cast<const T*>(bayer),
                                          width, sw, sh, pattern, literal_positions ? 2 : stride, chroma, intensity, mask));
  return TDK_OK;
}

TDK_EXPORT int tdk_wb_collect_samples(const float* bayer, int width, int height, uint32_t pattern, int stride, int literal_positions, float* chroma,
                                      float* intensity, uint8_t* mask, tdk_stream_t stream) {
  return tdk_wb_collect_samples_ex(bayer, width, height, pattern, stride, literal_positions, chroma, intensity, mask, TDK_F32, stream);
}
