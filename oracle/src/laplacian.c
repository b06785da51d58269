/*
 * oracle/src/laplacian.c -- local Laplacian filter, 6 gamma levels, fp16 pyramid storage
 * (CPU oracle, test infrastructure only).
 *
 * Follows reference csrc/local_contrast/laplacian.cu:
 *   sizes          :50,:415-418  dl(x,l) = (x + 2^l - 1) >> l; levels = min(30, floor(log2(min(W,H))));
 *                                pad = 2^(levels-1) replicate pixels per side
 *   pad            :90-109       fp32 -> fp16
 *   gauss reduce   :177-207      5x5 binomial at 2c, c clamped to [1, size-2], fp16 -> fp16
 *   curve          :266-290      remap around g_k = (k + .5)/6
 *   assemble       :221-252      expand(coarse output) + lerp of the two bracketing Laplacians
 *   expand         :111-141      parity-dependent 2x2 / 2x3 / 3x3 taps, x4
 *   boundary clamp :53-65
 *   sequencing     :482-592      top gaussian level lives in the output pyramid (:526)
 * Every stored value is rounded to binary16 (write_imagef_half), math is fp32.
 */
#include "common.h"

#define NG 6

typedef struct { uint16_t* d; int w, h; } himg;

static inline int dl(int x, int level) { return (x + (1 << level) - 1) >> level; }
static inline float hget(const himg* im, int x, int y) { return f16_bits_to_f32(im->d[(size_t)y * im->w + x]); }
static inline void hset(himg* im, int x, int y, float v) { im->d[(size_t)y * im->w + x] = f32_to_f16_bits(v); }

static const float BW[5] = {1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f};

static void gauss_reduce(const himg* fine, himg* coarse) {
#pragma omp parallel for schedule(static)
  for (int py = 0; py < coarse->h; py++)
    for (int px = 0; px < coarse->w; px++) {
      int cx = px, cy = py;
      if (px >= coarse->w - 1) cx = coarse->w - 2;
      if (py >= coarse->h - 1) cy = coarse->h - 2;
      if (cx <= 0) cx = 1;
      if (cy <= 0) cy = 1;
      float acc = 0.0f;
      for (int j = -2; j <= 2; j++)
        for (int i = -2; i <= 2; i++) acc += hget(fine, 2 * cx + i, 2 * cy + j) * BW[i + 2] * BW[j + 2];
      hset(coarse, px, py, acc);
    }
}

static float expand_gaussian(const himg* coarse, int x, int y) {
  const int cx = x / 2, cy = y / 2;
  const int x_odd = x & 1, y_odd = y & 1;
  const int i0 = x_odd ? 0 : -1, j0 = y_odd ? 0 : -1;
  float c = 0.0f;
  for (int i = i0; i <= 1; i++)
    for (int j = j0; j <= 1; j++) {
      const float p = hget(coarse, cx + i, cy + j);
      const int wi = x_odd ? (2 * i + 1) : (2 * i + 2);
      const int wj = y_odd ? (2 * j + 1) : (2 * j + 2);
      c += p * BW[wi] * BW[wj];
    }
  return 4.0f * c;
}

static void clamp_boundary(int* x, int* y, int w, int h) {
  if (w & 1) { if (*x > w - 2) *x = w - 2; } else { if (*x > w - 3) *x = w - 3; }
  if (h & 1) { if (*y > h - 2) *y = h - 2; } else { if (*y > h - 3) *y = h - 3; }
  if (*x <= 0) *x = 1;
  if (*y <= 0) *y = 1;
}

static float curve(float x, float g, float sigma, float shadows, float highlights, float clarity) {
  const float c = x - g;
  float val;
  const float ssigma = c > 0.0f ? sigma : -sigma;
  const float shadhi = c > 0.0f ? shadows : highlights;
  if (fabsf(c) > 2 * sigma) {
    val = g + ssigma + shadhi * (c - ssigma);
  } else {
    const float t = f_clip01(c / (2.0f * ssigma));
    const float t2 = t * t;
    const float mt = 1.0f - t;
    val = g + ssigma * 2.0f * mt * t + t2 * (ssigma + ssigma * shadhi);
  }
  const float exp_arg = -c * c / (2.0f * sigma * sigma / 3.0f);
  val += clarity * c * expf(exp_arg);
  return val;
}

TDK_API int oracle_laplacian_levels(int width, int height) {
  const int m = width < height ? width : height;
  int lg = 0;
  while ((1 << (lg + 1)) <= m) lg++;
  return lg < 30 ? lg : 30;
}

TDK_API int oracle_laplacian(const float* in, float* out, int width, int height, float sigma, float shadows, float highlights, float clarity) {
  const int L = oracle_laplacian_levels(width, height);
  if (L < 2) return 1;
  const int pad = 1 << (L - 1);
  const int bw = width + 2 * pad, bh = height + 2 * pad;

  himg* padded = (himg*)calloc(L, sizeof(himg));
  himg* output = (himg*)calloc(L, sizeof(himg));
  himg* proc = (himg*)calloc((size_t)L * NG, sizeof(himg));
  for (int l = 0; l < L; l++) {
    const int lw = dl(bw, l), lh = dl(bh, l);
    padded[l].w = output[l].w = lw;
    padded[l].h = output[l].h = lh;
    padded[l].d = (uint16_t*)calloc((size_t)lw * lh, 2);
    output[l].d = (uint16_t*)calloc((size_t)lw * lh, 2);
    for (int k = 0; k < NG; k++) {
      proc[k * L + l].w = lw;
      proc[k * L + l].h = lh;
      proc[k * L + l].d = (uint16_t*)calloc((size_t)lw * lh, 2);
    }
  }

  /* pad_input_half */
#pragma omp parallel for schedule(static)
  for (int y = 0; y < bh; y++)
    for (int x = 0; x < bw; x++) {
      int cx = x - pad, cy = y - pad;
      if (cx >= width) cx = width - 1;
      if (cy >= height) cy = height - 1;
      if (cx < 0) cx = 0;
      if (cy < 0) cy = 0;
      hset(&padded[0], x, y, in[(size_t)cy * width + cx]);
    }

  /* gaussian pyramid of the input; the coarsest level is stored in output[L-1] */
  for (int l = 1; l < L; l++) gauss_reduce(&padded[l - 1], (l == L - 1) ? &output[l] : &padded[l]);

  /* processed pyramids */
  for (int k = 0; k < NG; k++) {
    const float g = ((float)k + 0.5f) / (float)NG;
    himg* p0 = &proc[k * L];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < bh; y++)
      for (int x = 0; x < bw; x++) hset(p0, x, y, curve(hget(&padded[0], x, y), g, sigma, shadows, highlights, clarity));
    for (int l = 1; l < L; l++) gauss_reduce(&proc[k * L + l - 1], &proc[k * L + l]);
  }

  /* assemble from coarse to fine */
  for (int l = L - 2; l >= 0; l--) {
    const int pw = dl(bw, l), ph = dl(bh, l);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ph; y++)
      for (int x = 0; x < pw; x++) {
        int qx = x, qy = y;
        clamp_boundary(&qx, &qy, pw, ph);
        float val = expand_gaussian(&output[l + 1], qx, qy);
        const float v = hget(&padded[l], x, y);
        int hi = 1;
        for (; hi < NG - 1 && ((float)hi + .5f) / (float)NG <= v; hi++) {}
        const int lo = hi - 1;
        const float a = fminf(fmaxf(v * NG - ((float)lo + .5f), 0.0f), 1.0f);
        const float l0 = hget(&proc[lo * L + l], x, y) - expand_gaussian(&proc[lo * L + l + 1], qx, qy);
        const float l1 = hget(&proc[(lo + 1) * L + l], x, y) - expand_gaussian(&proc[(lo + 1) * L + l + 1], qx, qy);
        val += l0 * (1.0f - a) + l1 * a;
        hset(&output[l], x, y, val);
      }
  }

  /* write_back_half */
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) out[(size_t)y * width + x] = hget(&output[0], x + pad, y + pad);

  for (int l = 0; l < L; l++) {
    free(padded[l].d);
    free(output[l].d);
    for (int k = 0; k < NG; k++) free(proc[k * L + l].d);
  }
  free(padded); free(output); free(proc);
  return 0;
}
