/*
 * oracle/src/bilinear.c -- 13-tap "5x5 diamond" linear demosaic (CPU oracle, test infrastructure only).
 *
 * Follows reference csrc/debayer/bilinear.cu:17-99:
 *  - taps enumerated column by column over the diamond |dx|+|dy| <= 2 (bilinear.cu:17-23,
 *    int2 is {x, y}), edge-clamped reads that ignore CFA parity (bilinear.cu:90);
 *  - per site class (R, G on a red row, G on a blue row, B) one integer weight per tap
 *    and output channel, every channel's weights summing to 16 (bilinear.cu:28-61);
 *  - out = (sum_k w_k * v_k) / (sum_k w_k), accumulated in tap order (bilinear.cu:86-97).
 * Site class of each 2x2 position per pattern: bayer_device.h:19-34.
 */
#include "common.h"

#define NTAP 13
static const int8_t TAP_DX[NTAP] = {-2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2};
static const int8_t TAP_DY[NTAP] = {0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0};

/* Building blocks, in tap order.  Malvar-He-Cutler filters scaled by 16. */
#define W_IDENT {0, 0, 0, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0}
/* green at a red/blue site */
#define W_G_AT_RB {-2, 0, 4, 0, -2, 4, 8, 4, -2, 0, 4, 0, -2}
/* red at blue site / blue at red site (diagonal neighbours) */
#define W_RB_AT_BR {-3, 4, 0, 4, -3, 0, 12, 0, -3, 4, 0, 4, -3}
/* chroma at a green site whose same-colour neighbours are left/right */
#define W_C_AT_G_H {-2, -2, 8, -2, 1, 0, 10, 0, 1, -2, 8, -2, -2}
/* chroma at a green site whose same-colour neighbours are above/below */
#define W_C_AT_G_V {1, -2, 0, -2, -2, 8, 10, 8, -2, -2, 0, -2, 1}

/* [site class][channel][tap]; classes: 0 = R, 1 = G (red row), 2 = G (blue row), 3 = B */
static const int8_t WEIGHTS[4][3][NTAP] = {
    {W_IDENT, W_G_AT_RB, W_RB_AT_BR},
    {W_C_AT_G_H, W_IDENT, W_C_AT_G_V},
    {W_C_AT_G_V, W_IDENT, W_C_AT_G_H},
    {W_RB_AT_BR, W_G_AT_RB, W_IDENT},
};

/* site class at 2x2 position C = (y&1)*2 + (x&1): bayer_device.h:19-22 */
static void site_classes(uint32_t pattern, int cls[4]) {
  static const int rggb[4] = {0, 1, 2, 3}, bggr[4] = {3, 1, 2, 0}, grbg[4] = {1, 0, 3, 2}, gbrg[4] = {1, 3, 0, 2};
  const int* src = rggb;
  if (pattern == TDK_BGGR) src = bggr;
  else if (pattern == TDK_GRBG) src = grbg;
  else if (pattern == TDK_GBRG) src = gbrg;
  for (int i = 0; i < 4; i++) cls[i] = src[i];
}

TDK_API void oracle_bilinear5x5(const float* in, float* out, int width, int height, uint32_t pattern) {
  int cls[4];
  site_classes(pattern, cls);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      const int t = cls[(y & 1) * 2 + (x & 1)];
      float acc[3] = {0.0f, 0.0f, 0.0f}, wsum[3] = {0.0f, 0.0f, 0.0f};
      for (int k = 0; k < NTAP; k++) {
        int cx = x + TAP_DX[k], cy = y + TAP_DY[k];
        cx = cx < 0 ? 0 : (cx > width - 1 ? width - 1 : cx);
        cy = cy < 0 ? 0 : (cy > height - 1 ? height - 1 : cy);
        const float v = in[(size_t)cy * width + cx];
        for (int c = 0; c < 3; c++) {
          const float w = (float)WEIGHTS[t][c][k];
          acc[c] += w * v;
          wsum[c] += w;
        }
      }
      float* o = out + ((size_t)y * width + x) * 3;
      for (int c = 0; c < 3; c++) o[c] = acc[c] / wsum[c];
    }
  }
}
