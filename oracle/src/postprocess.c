/*
 * oracle/src/postprocess.c -- demosaic post-processing (CPU oracle, test infrastructure only).
 *
 * Follows reference csrc/debayer/postprocess.cu:
 *   colour smoothing     :24-78   3x3 median of (R-G) and (B-G), zero halo outside the image,
 *                                 19-compare-swap network from csrc/reduction.h:85-116
 *   global green eq      :175-255 + host code :350-376 (ratio = sum(G2)/sum(G1))
 *   local green eq       :84-169  threshold = green_eq_threshold / 100 (:383)
 *   order                :341-387 smoothing passes -> global -> local
 */
#include "common.h"

static inline void cswap(float* a, float* b) {
  const float x = *a;
  const int c = *a > *b;
  *a = c ? *b : *a;
  *b = c ? x : *b;
}

/* reduction.h:93-116: returns s4 of the network */
static float median9(float s[9]) {
  cswap(&s[1], &s[2]); cswap(&s[4], &s[5]); cswap(&s[7], &s[8]);
  cswap(&s[0], &s[1]); cswap(&s[3], &s[4]); cswap(&s[6], &s[7]);
  cswap(&s[1], &s[2]); cswap(&s[4], &s[5]); cswap(&s[7], &s[8]);
  cswap(&s[0], &s[3]); cswap(&s[5], &s[8]); cswap(&s[4], &s[7]);
  cswap(&s[3], &s[6]); cswap(&s[1], &s[4]); cswap(&s[2], &s[5]);
  cswap(&s[4], &s[7]); cswap(&s[4], &s[2]); cswap(&s[6], &s[4]);
  cswap(&s[4], &s[2]);
  return s[4];
}

static void color_smoothing(const float* in, float* out, int w, int h) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float dr[9], db[9];
      int k = 0;
      for (int j = -1; j <= 1; j++)
        for (int i = -1; i <= 1; i++, k++) {
          const int xx = x + i, yy = y + j;
          if (xx >= 0 && yy >= 0 && xx < w && yy < h) {
            const float* p = in + ((size_t)yy * w + xx) * 3;
            dr[k] = p[0] - p[1];
            db[k] = p[2] - p[1];
          } else {
            dr[k] = 0.0f - 0.0f;
            db[k] = 0.0f - 0.0f;
          }
        }
      const float* c = in + ((size_t)y * w + x) * 3;
      float* o = out + ((size_t)y * w + x) * 3;
      const float r = fmaxf(median9(dr) + c[1], 0.0f);
      const float b = fmaxf(median9(db) + c[1], 0.0f);
      o[0] = fmaxf(r, 0.0f);
      o[1] = fmaxf(c[1], 0.0f);
      o[2] = fmaxf(b, 0.0f);
    }
}

/* postprocess.cu:175-226: per 16x16 block, pairwise tree over the 256 lanes (l < offset
 * adds lane l + offset), then the per-block partials are summed.  The reference sums the
 * partials with torch's CUDA sum (order unspecified); the oracle adds them in block order
 * in fp32 and reports the same in fp64 so a test can bound the difference. */
TDK_API void oracle_green_eq_sums(const float* in, int w, int h, uint32_t pattern, float sums_f32[2], double sums_f64[2]) {
  const int gx = (w + 15) / 16, gy = (h + 15) / 16;
  float s1 = 0.0f, s2 = 0.0f;
  double d1 = 0.0, d2 = 0.0;
  for (int by = 0; by < gy; by++)
    for (int bx = 0; bx < gx; bx++) {
      float b1[256], b2[256];
      for (int ly = 0; ly < 16; ly++)
        for (int lx = 0; lx < 16; lx++) {
          const int x = bx * 16 + lx, y = by * 16 + ly;
          const int c = cfa_color(y, x, pattern);
          const int inimg = (x < 2 * (w / 2) && y < 2 * (h / 2));
          const float g = (x < w && y < h) ? in[((size_t)y * w + x) * 3 + 1] : 0.0f;
          b1[ly * 16 + lx] = (inimg && c == 1 && !(y & 1)) ? g : 0.0f;
          b2[ly * 16 + lx] = (inimg && c == 1 && (y & 1)) ? g : 0.0f;
        }
      for (int off = 128; off > 0; off /= 2)
        for (int l = 0; l < off; l++) { b1[l] += b1[l + off]; b2[l] += b2[l + off]; }
      s1 += b1[0]; s2 += b2[0];
      d1 += (double)b1[0]; d2 += (double)b2[0];
    }
  sums_f32[0] = s1; sums_f32[1] = s2;
  sums_f64[0] = d1; sums_f64[1] = d2;
}

static void green_eq_global_apply(const float* in, float* out, int w, int h, uint32_t pattern, float ratio) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const float* p = in + ((size_t)y * w + x) * 3;
      float* o = out + ((size_t)y * w + x) * 3;
      const int g1 = (cfa_color(y, x, pattern) == 1 && !(y & 1));
      const float g = p[1] * (g1 ? ratio : 1.0f);
      o[0] = fmaxf(p[0], 0.0f); o[1] = fmaxf(g, 0.0f); o[2] = fmaxf(p[2], 0.0f);
    }
}

static inline float g0(const float* img, int x, int y, int w, int h) {
  return (x >= 0 && y >= 0 && x < w && y < h) ? img[((size_t)y * w + x) * 3 + 1] : 0.0f;
}

static void green_eq_local(const float* in, float* out, int w, int h, uint32_t pattern, float threshold) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const float* p = in + ((size_t)y * w + x) * 3;
      float* dst = out + ((size_t)y * w + x) * 3;
      const float maximum = 1.0f;
      float o = p[1];
      if (cfa_color(y, x, pattern) == 1 && (y & 1)) {
        const float o1_1 = g0(in, x - 1, y - 1, w, h), o1_2 = g0(in, x + 1, y - 1, w, h);
        const float o1_3 = g0(in, x - 1, y + 1, w, h), o1_4 = g0(in, x + 1, y + 1, w, h);
        const float o2_1 = g0(in, x, y - 2, w, h), o2_2 = g0(in, x, y + 2, w, h);
        const float o2_3 = g0(in, x - 2, y, w, h), o2_4 = g0(in, x + 2, y, w, h);
        const float m1 = (o1_1 + o1_2 + o1_3 + o1_4) / 4.0f;
        const float m2 = (o2_1 + o2_2 + o2_3 + o2_4) / 4.0f;
        if ((m2 > 0.0f) && (m1 > 0.0f) && (m1 / m2 < maximum * 2.0f)) {
          const float c1 = (fabsf(o1_1 - o1_2) + fabsf(o1_1 - o1_3) + fabsf(o1_1 - o1_4) + fabsf(o1_2 - o1_3) + fabsf(o1_3 - o1_4) + fabsf(o1_2 - o1_4)) / 6.0f;
          const float c2 = (fabsf(o2_1 - o2_2) + fabsf(o2_1 - o2_3) + fabsf(o2_1 - o2_4) + fabsf(o2_2 - o2_3) + fabsf(o2_3 - o2_4) + fabsf(o2_2 - o2_4)) / 6.0f;
          if ((o < maximum * 0.95f) && (c1 < maximum * threshold) && (c2 < maximum * threshold)) o *= m1 / m2;
        }
      }
      dst[0] = p[0]; dst[1] = fmaxf(o, 0.0f); dst[2] = p[2];
    }
}

/* ratio_override < 0: compute the ratio from the fp32 block-order sums (see above);
 * otherwise use the given ratio (lets a test feed the device-computed ratio). */
TDK_API void oracle_postprocess(const float* in, float* out, int w, int h, uint32_t pattern, int smoothing_passes,
                                int eq_local, int eq_global, float eq_threshold, float ratio_override) {
  const size_t n3 = (size_t)w * h * 3;
  float* a = (float*)malloc(n3 * sizeof(float));
  float* b = (float*)malloc(n3 * sizeof(float));
  memcpy(a, in, n3 * sizeof(float));
  for (int p = 0; p < smoothing_passes; p++) {
    color_smoothing(a, b, w, h);
    float* t = a; a = b; b = t;
  }
  if (eq_global) {
    float ratio = ratio_override;
    if (ratio_override < 0.0f) {
      float s[2];
      double d[2];
      oracle_green_eq_sums(a, w, h, pattern, s, d);
      ratio = (s[0] > 0.0f && s[1] > 0.0f) ? s[1] / s[0] : 1.0f;
    }
    green_eq_global_apply(a, b, w, h, pattern, ratio);
    float* t = a; a = b; b = t;
  }
  if (eq_local) {
    /* postprocess.cu:383: green_eq_threshold_ / 100. is a double expression narrowed to
     * the kernel's float parameter */
    green_eq_local(a, b, w, h, pattern, (float)((double)eq_threshold / 100.0));
    float* t = a; a = b; b = t;
  }
  memcpy(out, a, n3 * sizeof(float));
  free(a);
  free(b);
}
