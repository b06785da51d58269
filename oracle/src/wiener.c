/*
 * oracle/src/wiener.c -- tiled-FFT Wiener shrinkage denoiser (CPU oracle, test
 * infrastructure only).
 *
 * Follows reference csrc/denoise/denoise.cu:84-242,266-331, csrc/denoise/fft.h and
 * csrc/denoise/window.h:18-42:
 *   tile g in [0, grid), origin (g - ov) * s with s = K / ov, grid = ceil((L + K) / s) + ov
 *   load with reflect(x) = x < 0 ? -x : (x >= L ? 2L - x - 1 : x)        (denoise.cu:118-148)
 *   tile mean over K*K samples; v = (x - mean) * wf[tx] * wf[ty]          (:84-101,:204-205)
 *   2-D FFT = radix-2 DIT along x, transpose, along y (fft.h:133-230); inverse: along y
 *   first, then x, each 1-D inverse pass scaled by 1/K
 *   gain = max(|X|^2 + 1e-15 - sigma^2, 0) / (|X|^2 + 1e-15)                (:181-185)
 *   accumulate (y + mean * wf2d) * wi2d and wf2d * wi2d into the (H+2K, W+2K) buffers at
 *   origin + t + K, bounds-checked on the high side only                    (:151-178)
 *   out = acc / (mask + 1e-15), cropped by K                                (:223-242)
 * Windows: w[i] = exp(-r_i^2 / (0.3 (K/2)^2)) / ||.||_2, r_i = -K/2 + 0.5 + i (window.h:22-35);
 * the reference evaluates them with torch fp32 ops on the GPU, the oracle in fp64 rounded
 * once to fp32 (within an ulp of either).
 * The reference accumulates tiles with float atomics in nondeterministic order and reduces
 * the tile mean the same way; the oracle uses a fixed order (mean in fp64).
 */
#include "common.h"

typedef struct { float re, im; } cpx;

static inline cpx c_add(cpx a, cpx b) { cpx r = {a.re + b.re, a.im + b.im}; return r; }
static inline cpx c_sub(cpx a, cpx b) { cpx r = {a.re - b.re, a.im - b.im}; return r; }
static inline cpx c_mul(cpx a, cpx b) { cpx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }

/* fft.h:59-118 writes the twiddles as 10-significant-decimal literals; reproduce that rounding */
static float lit10(double v) { return (float)(round(v * 1e10) / 1e10); }

static void make_twiddles(int n, int inverse, cpx* tw) {
  for (int k = 0; k < n / 2; k++) {
    const double a = 2.0 * M_PI * k / n;
    tw[k].re = lit10(cos(a));
    tw[k].im = lit10(inverse ? sin(a) : -sin(a));
  }
}

/* fft.h:133-167 */
static void fft1d(cpx* d, int n, int stages, const cpx* tw, float norm) {
  cpx tmp[32];
  for (int t = 0; t < n; t++) {
    int r = 0;
    for (int b = 0; b < stages; b++) r |= ((t >> b) & 1) << (stages - 1 - b);
    tmp[t] = d[r];
  }
  for (int s = 0; s < stages; s++) {
    const int step = 1 << s;
    cpx nxt[32];
    for (int t = 0; t < n; t++) {
      const int partner = t ^ step;
      if ((t & step) == 0) nxt[t] = c_add(tmp[t], c_mul(tmp[partner], tw[(t & (step - 1)) * ((n / 2) >> s)]));
      else nxt[t] = c_sub(tmp[partner], c_mul(tmp[t], tw[(partner & (step - 1)) * ((n / 2) >> s)]));
    }
    memcpy(tmp, nxt, sizeof(cpx) * n);
  }
  for (int t = 0; t < n; t++) { d[t].re = tmp[t].re * norm; d[t].im = tmp[t].im * norm; }
}

TDK_API void oracle_wiener_window(int K, float weight, float* w) {
  const double half = K / 2.0;
  const double scale = (double)weight * half * half;
  double v[32], nrm = 0.0;
  for (int i = 0; i < K; i++) {
    const double r = -half + 0.5 + i;
    v[i] = exp(-(r * r) / scale);
    nrm += v[i] * v[i];
  }
  nrm = sqrt(nrm);
  for (int i = 0; i < K; i++) w[i] = (float)(v[i] / nrm);
}

static inline int reflect(int x, int limit) {
  if (x < 0) x = -x;
  if (x >= limit) x = 2 * limit - x - 1;
  return x;
}

/* Process one tile for one channel: tile[K*K] in -> reconstructed samples out (before the
 * interpolation window), returns nothing; all per-tile math in fp32. */
static void tile_channel(float* v, int K, int stages, const cpx* twf, const cpx* twi, float sigma) {
  cpx X[32 * 32];
  cpx line[32];
  for (int y = 0; y < K; y++) { /* forward along x */
    for (int x = 0; x < K; x++) { line[x].re = v[y * K + x]; line[x].im = 0.0f; }
    fft1d(line, K, stages, twf, 1.0f);
    for (int x = 0; x < K; x++) X[y * K + x] = line[x];
  }
  for (int kx = 0; kx < K; kx++) { /* forward along y */
    for (int y = 0; y < K; y++) line[y] = X[y * K + kx];
    fft1d(line, K, stages, twf, 1.0f);
    for (int y = 0; y < K; y++) X[y * K + kx] = line[y];
  }
  for (int i = 0; i < K * K; i++) { /* denoise.cu:181-185 */
    const float power = (X[i].re * X[i].re + X[i].im * X[i].im) + 1e-15f;
    const float gain = fmaxf(power - sigma * sigma, 0.0f) / power;
    X[i].re = gain * X[i].re;
    X[i].im = gain * X[i].im;
  }
  const float inv = 1.0f / (float)K;
  for (int kx = 0; kx < K; kx++) { /* inverse along y */
    for (int y = 0; y < K; y++) line[y] = X[y * K + kx];
    fft1d(line, K, stages, twi, inv);
    for (int y = 0; y < K; y++) X[y * K + kx] = line[y];
  }
  for (int y = 0; y < K; y++) { /* inverse along x */
    for (int x = 0; x < K; x++) line[x] = X[y * K + x];
    fft1d(line, K, stages, twi, inv);
    for (int x = 0; x < K; x++) v[y * K + x] = line[x].re;
  }
}

TDK_API int oracle_wiener(const float* in, float* out, int W, int H, int C, int K, int ov, const float* sigmas) {
  if (!(K == 16 || K == 32) || !(C == 1 || C == 3) || !(ov == 2 || ov == 4 || ov == 8)) return 1;
  if (W < K || H < K) return 2; /* reflect() would index out of the image */
  const int stages = (K == 16) ? 4 : 5;
  const int s = K / ov;
  const int Hp = H + 2 * K, Wp = W + 2 * K;
  const int grid_h = (H + K + s - 1) / s + ov, grid_w = (W + K + s - 1) / s + ov;
  cpx twf[16], twi[16];
  make_twiddles(K, 0, twf);
  make_twiddles(K, 1, twi);
  float wf[32], wi[32];
  oracle_wiener_window(K, 0.3f, wf);
  oracle_wiener_window(K, 0.3f, wi);

  float* acc = (float*)calloc((size_t)Hp * Wp * C, sizeof(float));
  float* mask = (float*)calloc((size_t)Hp * Wp, sizeof(float));

  /* tile rows whose index differs by >= ov never touch the same output rows */
  for (int phase = 0; phase < ov; phase++) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int gy = phase; gy < grid_h; gy += ov) {
      float tile[3][32 * 32];
      const int oy = (gy - ov) * s;
      if (oy + K <= 0 || oy >= H) continue; /* contributes only to cropped-away rows */
      for (int gx = 0; gx < grid_w; gx++) {
        const int ox = (gx - ov) * s;
        if (ox + K <= 0 || ox >= W) continue;
        float mean[3];
        for (int c = 0; c < C; c++) {
          double sum = 0.0;
          for (int ty = 0; ty < K; ty++)
            for (int tx = 0; tx < K; tx++) {
              const float v = in[((size_t)reflect(oy + ty, H) * W + reflect(ox + tx, W)) * C + c];
              tile[c][ty * K + tx] = v;
              sum += (double)v;
            }
          mean[c] = (float)(sum / (double)(K * K));
          for (int ty = 0; ty < K; ty++)
            for (int tx = 0; tx < K; tx++) tile[c][ty * K + tx] = (tile[c][ty * K + tx] - mean[c]) * (wf[tx] * wf[ty]);
          tile_channel(tile[c], K, stages, twf, twi, sigmas[c]);
        }
        for (int ty = 0; ty < K; ty++) {
          const int py = oy + ty + K;
          if (py >= Hp) continue;
          for (int tx = 0; tx < K; tx++) {
            const int px = ox + tx + K;
            if (px >= Wp) continue;
            const float fw = wf[tx] * wf[ty], iw = wi[tx] * wi[ty];
            const size_t o = (size_t)py * Wp + px;
            for (int c = 0; c < C; c++) acc[o * C + c] += (tile[c][ty * K + tx] + mean[c] * fw) * iw;
            mask[o] += fw * iw;
          }
        }
      }
    }
  }

#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const size_t p = (size_t)(y + K) * Wp + (x + K);
      for (int c = 0; c < C; c++) out[((size_t)y * W + x) * C + c] = acc[p * C + c] / (mask[p] + 1e-15f);
    }
  free(acc);
  free(mask);
  return 0;
}
