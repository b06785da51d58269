/*
 * oracle/src/whitebalance.c -- per-CFA-site white-balance gain (CPU oracle, test
 * infrastructure only).  Follows reference csrc/white_balance.cu:10-42,164-183:
 * out = clamp(in * gain[channel], 0, 1) on a clone; CFA code 1 (and the dead code 3) -> green.
 */
#include "common.h"

TDK_API void oracle_apply_white_balance(const float* in, float* out, int width, int height, const float gains[3], uint32_t pattern) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      const int c = cfa_color(y, x, pattern);
      const float g = (c == 0) ? gains[0] : (c == 2 ? gains[2] : gains[1]);
      out[(size_t)y * width + x] = f_clamp(in[(size_t)y * width + x] * g, 0.0f, 1.0f);
    }
}

/*
 * Sample collection of estimate_white_balance (reference csrc/white_balance.cu:57-126): one
 * sample per cell (i, j) of the (height/stride) x (width/stride) grid, cells with
 * i + 1 >= height/stride or j + 1 >= width/stride are skipped (:69).  A sample is the 2x2 CFA
 * quad at (y0, x0) turned into RGB by bayer_2x2_to_rgb (bayer_device.h:35-43); chroma =
 * (r, g) / (r + g + b), intensity = r + g + b, valid = max(quad) < 1 (:73-81).
 *
 * Two reference slips, handled explicitly:
 *  - the quad is read at pos * 2 although the grid is sized by `stride` (:71), so the reference
 *    only ever looks at the top-left (2/stride)^2 of the frame.  literal_positions != 0
 *    reproduces that (the default of the numpy front-end and of the product, which returns the
 *    reference's gains); 0 reads cell (i, j) at (i * stride, j * stride) -- the documented intent
 *    ("Pixel sampling stride", white_balance.py:47), an opt-in correction in the product;
 *  - skipped cells leave chroma / intensity / mask uninitialised (torch::empty, :107-109), so
 *    the reference's result depends on stale memory.  Here (and in the product) skipped cells
 *    are invalid (mask = 0, values 0).
 * Outputs are full-grid arrays: chroma[n][2], intensity[n], mask[n] with n = sh * sw.
 */
TDK_API void oracle_wb_collect_samples(const float* bayer, int width, int height, uint32_t pattern, int stride, int literal_positions,
                                       float* chroma, float* intensity, uint8_t* mask) {
  const int sh = height / stride, sw = width / stride;
  const int step = literal_positions ? 2 : stride;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < sh; i++)
    for (int j = 0; j < sw; j++) {
      const size_t n = (size_t)i * sw + j;
      if (j + 1 >= sw || i + 1 >= sh) {
        chroma[2 * n] = chroma[2 * n + 1] = intensity[n] = 0.0f;
        mask[n] = 0;
        continue;
      }
      const int y0 = i * step, x0 = j * step;
      const float p00 = bayer[(size_t)y0 * width + x0], p01 = bayer[(size_t)y0 * width + x0 + 1];
      const float p10 = bayer[(size_t)(y0 + 1) * width + x0], p11 = bayer[(size_t)(y0 + 1) * width + x0 + 1];
      float r, g, b;
      switch (pattern) {
        case 0x94949494u: r = p00; g = (p01 + p10) * 0.5f; b = p11; break; /* RGGB */
        case 0x16161616u: r = p11; g = (p01 + p10) * 0.5f; b = p00; break; /* BGGR */
        case 0x61616161u: r = p01; g = (p00 + p11) * 0.5f; b = p10; break; /* GRBG */
        default:          r = p10; g = (p00 + p11) * 0.5f; b = p01; break; /* GBRG */
      }
      const float s = r + g + b;
      chroma[2 * n] = r / s;
      chroma[2 * n + 1] = g / s;
      intensity[n] = s;
      mask[n] = fmaxf(fmaxf(p00, p01), fmaxf(p10, p11)) < 1.0f;
    }
}
