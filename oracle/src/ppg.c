/*
 * oracle/src/ppg.c -- Pattern Pixel Grouping demosaic (CPU oracle, test infrastructure only).
 *
 * Follows reference csrc/debayer/ppg.cu:
 *   border ring average  :342-389   (border_interpolate, also used by RCD)
 *   optional pre-median  :21-113    (threshold = median_threshold / 100, :445-449)
 *   green at R/B sites   :120-224
 *   red/blue fill        :230-337
 *   launch sequence      :413-464   (border -> [median] -> green -> red/blue)
 * Halo reads outside the image return 0 (the LDS fills at :57,:156,:267).
 */
#include "common.h"
#include "stencils.h"

/* ppg.cu:342-389 -- writes RGB for every pixel outside [border, dim-border) */
void tdk_border_interpolate(const float* in, float* rgb, int width, int height, uint32_t pattern, int border) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      if (x >= border && x < width - border && y >= border && y < height - border) continue;
      float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      int count[4] = {0, 0, 0, 0};
      for (int j = y - 1; j <= y + 1; j++)
        for (int i = x - 1; i <= x + 1; i++)
          if (j >= 0 && i >= 0 && j < height && i < width) {
            const int f = cfa_color(j, i, pattern);
            sum[f] += fmaxf(0.0f, in[(size_t)j * width + i]);
            count[f]++;
          }
      const float own = fmaxf(0.0f, in[(size_t)y * width + x]);
      float o[3];
      o[0] = count[0] > 0 ? sum[0] / (float)count[0] : own;
      o[1] = (count[1] + count[3]) > 0 ? (sum[1] + sum[3]) / (float)(count[1] + count[3]) : own;
      o[2] = count[2] > 0 ? sum[2] / (float)count[2] : own;
      const int f = cfa_color(y, x, pattern);
      if (f == 0) o[0] = own;
      else if (f == 2) o[2] = own;
      else o[1] = own;
      float* dst = rgb + ((size_t)y * width + x) * 3;
      dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    }
  }
}

static inline float px0(const float* img, int x, int y, int w, int h) {
  return (x >= 0 && y >= 0 && x < w && y < h) ? img[(size_t)y * w + x] : 0.0f;
}

/* ppg.cu:21-113 */
static void pre_median(const float* in, float* out, int width, int height, uint32_t pattern, float threshold) {
  static const int lim[5] = {0, 1, 2, 1, 0};
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      const float center = in[(size_t)y * width + x];
      float med[9];
      int cnt = 0, k = 0;
      for (int i = 0; i < 5; i++)
        for (int j = -lim[i]; j <= lim[i]; j += 2) {
          const float v = px0(in, x + j, y + i - 2, width, height);
          if (fabsf(v - center) < threshold) { med[k++] = v; cnt++; }
          else med[k++] = 64.0f + v;
        }
      for (int i = 0; i < 8; i++)
        for (int ii = i + 1; ii < 9; ii++)
          if (med[i] > med[ii]) { float t = med[i]; med[i] = med[ii]; med[ii] = t; }
      float color = center;
      if (cfa_color(y, x, pattern) & 1) {
        const float target = (cnt == 1) ? (med[4] - 64.0f) : med[(cnt - 1) / 2];
        const float delta = target - center;
        color = center + fminf(fmaxf(delta, -threshold), threshold);
      }
      out[(size_t)y * width + x] = fmaxf(color, 0.0f);
    }
  }
}

/* ppg.cu:120-224; `tmp` already holds the border ring, interior is written here */
static void green_pass(const float* in, float* tmp, int width, int height, uint32_t pattern) {
#pragma omp parallel for schedule(static)
  for (int y = 3; y < height - 3; y++) {
    for (int x = 3; x < width - 3; x++) {
      const int c = cfa_color(y, x, pattern);
      float col[3] = {0.0f, 0.0f, 0.0f};
      const float pc = in[(size_t)y * width + x];
      col[c == 0 ? 0 : (c == 2 ? 2 : 1)] = pc;
      if (c == 0 || c == 2) {
        float nb[2][7];
        for (int d = -3; d <= 3; d++) {
          nb[0][d + 3] = px0(in, x + d, y, width, height);
          nb[1][d + 3] = px0(in, x, y + d, width, height);
        }
        col[1] = tdk_ppg_green(nb[0], nb[1]);
      }
      float* dst = tmp + ((size_t)y * width + x) * 3;
      for (int k = 0; k < 3; k++) dst[k] = fmaxf(col[k], 0.0f);
    }
  }
}

/* ppg.cu:230-337 */
static void redblue_pass(const float* tmp, float* out, int width, int height, uint32_t pattern) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      float nb[3][3][3];
      for (int j = -1; j <= 1; j++)
        for (int i = -1; i <= 1; i++) {
          const int xx = x + i, yy = y + j;
          const int ok = xx >= 0 && yy >= 0 && xx < width && yy < height;
          for (int k = 0; k < 3; k++) nb[j + 1][i + 1][k] = ok ? tmp[((size_t)yy * width + xx) * 3 + k] : 0.0f;
        }
      float col[3] = {nb[1][1][0], nb[1][1][1], nb[1][1][2]};
      if (!(x == 0 || y == 0 || x == width - 1 || y == height - 1))
        tdk_ppg_redblue(nb, cfa_color(y, x, pattern), cfa_color(y, x + 1, pattern) == 0, col);
      float* dst = out + ((size_t)y * width + x) * 3;
      for (int k = 0; k < 3; k++) dst[k] = fmaxf(col[k], 0.0f);
    }
  }
}

TDK_API void oracle_ppg(const float* in, float* out, int width, int height, uint32_t pattern, float median_threshold) {
  const size_t n = (size_t)width * height;
  float* tmp = (float*)calloc(n * 3, sizeof(float));
  float* med = NULL;
  tdk_border_interpolate(in, tmp, width, height, pattern, 3);
  const float* src = in;
  if (median_threshold > 0.0f) {
    med = (float*)calloc(n, sizeof(float));
    pre_median(in, med, width, height, pattern, median_threshold / 100.0f);
    src = med;
  }
  green_pass(src, tmp, width, height, pattern);
  redblue_pass(tmp, out, width, height, pattern);
  free(tmp);
  free(med);
}

TDK_API void oracle_border_interpolate(const float* in, float* rgb, int width, int height, uint32_t pattern, int border) {
  tdk_border_interpolate(in, rgb, width, height, pattern, border);
}
