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
