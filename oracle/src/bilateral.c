/*
 * oracle/src/bilateral.c -- bilateral-grid local contrast (CPU oracle, test infrastructure only).
 *
 * Follows reference csrc/local_contrast/bilateral.cu:
 *   grid size       :273-299  (round(W / sigma_s) clamped [4,3000], z clamped [4,50], then
 *                              ceil(W / effective_sigma) + 1)
 *   sample coords   :71-86    clamp(x / sigma_s, 0, sx - 1) with the RAW sigma; base
 *                              min(int(g), size - 2); frac may be exactly 1
 *   splat           :99-112   trilinear weights * 1 / sigma_s^2  (atomics in the reference;
 *                              pixel order here)
 *   blur x, y       :132-168  [1 4 6 4 1] / 16 with zero extension, x (grid -> tmp) then
 *                              y (tmp -> grid)                                   (:301-308)
 *   z derivative    :171-204  [-2 -4 0 4 2] / 16 with one-sided ends (grid -> tmp)
 *   slice           :208-228  max(0, L - detail * sigma_r * 4 * trilerp(tmp))
 * Grid layout x + sx * (y + sy * z) (device_math.h:446-452).
 */
#include "common.h"

TDK_API void oracle_bilateral_grid_size(int width, int height, float sigma_s, float sigma_r, int size[3]) {
  float ss = sigma_s;
  if (ss < 0.5f) ss = 0.5f;
  const float L_range = 1.0f;
  const float gx = f_clamp(roundf((float)width / ss), 4.0f, 3000.0f);
  const float gy = f_clamp(roundf((float)height / ss), 4.0f, 3000.0f);
  const float gz = f_clamp(roundf(L_range / sigma_r), 4.0f, 50.0f);
  const float s_s = fmaxf((float)height / gy, (float)width / gx);
  const float s_r = L_range / gz;
  size[0] = (int)ceilf((float)width / s_s) + 1;
  size[1] = (int)ceilf((float)height / s_s) + 1;
  size[2] = (int)ceilf(L_range / s_r) + 1;
}

typedef struct { int gi; int ox, oy, oz; float fx, fy, fz; } gsample;

static inline gsample make_sample(int x, int y, float L, const int size[3], float sigma_s, float sigma_r) {
  const float gx = f_clamp((float)x / sigma_s, 0.0f, (float)(size[0] - 1));
  const float gy = f_clamp((float)y / sigma_s, 0.0f, (float)(size[1] - 1));
  const float gz = f_clamp(L / sigma_r, 0.0f, (float)(size[2] - 1));
  int ix = (int)gx, iy = (int)gy, iz = (int)gz;
  if (ix > size[0] - 2) ix = size[0] - 2;
  if (iy > size[1] - 2) iy = size[1] - 2;
  if (iz > size[2] - 2) iz = size[2] - 2;
  gsample s;
  s.gi = ix + size[0] * (iy + size[1] * iz);
  s.ox = 1; s.oy = size[0]; s.oz = size[0] * size[1];
  s.fx = gx - (float)ix; s.fy = gy - (float)iy; s.fz = gz - (float)iz;
  return s;
}

/* one line of the 5-tap blur; stride in floats (bilateral.cu:132-168) */
static void blur_line(const float* ib, float* ob, int n, ptrdiff_t st) {
  const float w0 = 6.0f / 16.0f, w1 = 4.0f / 16.0f, w2 = 1.0f / 16.0f;
  if (n < 4) {
    /* The reference's line walker reads and writes past a line shorter than 4 cells (undefined
     * behaviour; reachable only for images a few pixels wide with an aspect ratio >= 4, where
     * compute_grid_size yields 3 cells).  Defined here -- and in the HIP kernels -- as the same
     * stencil with zero extension, which is what the literal code computes for every n >= 4. */
    for (int i = 0; i < n; i++) {
      const float m2 = i >= 2 ? ib[(i - 2) * st] : 0.0f, m1 = i >= 1 ? ib[(i - 1) * st] : 0.0f;
      const float p1 = i + 1 < n ? ib[(i + 1) * st] : 0.0f, p2 = i + 2 < n ? ib[(i + 2) * st] : 0.0f;
      ob[i * st] = ib[i * st] * w0 + w1 * (p1 + m1) + w2 * (p2 + m2);
    }
    return;
  }
  ptrdiff_t i0 = 0;
  float tmp1 = ib[i0];
  ob[i0] = ib[i0] * w0 + w1 * ib[i0 + st] + w2 * ib[i0 + 2 * st];
  i0 += st;
  float tmp2 = ib[i0];
  ob[i0] = ib[i0] * w0 + w1 * (ib[i0 + st] + tmp1) + w2 * ib[i0 + 2 * st];
  i0 += st;
  for (int i = 2; i < n - 2; i++) {
    const float tmp3 = ib[i0];
    ob[i0] = ib[i0] * w0 + w1 * (ib[i0 + st] + tmp2) + w2 * (ib[i0 + 2 * st] + tmp1);
    i0 += st;
    tmp1 = tmp2;
    tmp2 = tmp3;
  }
  const float tmp3 = ib[i0];
  ob[i0] = ib[i0] * w0 + w1 * (ib[i0 + st] + tmp2) + w2 * tmp1;
  i0 += st;
  ob[i0] = ib[i0] * w0 + w1 * tmp3 + w2 * tmp2;
}

/* bilateral.cu:171-204 */
static void blur_line_z(const float* ib, float* ob, int n, ptrdiff_t st) {
  const float w1 = 4.0f / 16.0f, w2 = 2.0f / 16.0f;
  ptrdiff_t i0 = 0;
  float tmp1 = ib[i0];
  ob[i0] = w1 * ib[i0 + st] + w2 * ib[i0 + 2 * st];
  i0 += st;
  float tmp2 = ib[i0];
  ob[i0] = w1 * (ib[i0 + st] - tmp1) + w2 * ib[i0 + 2 * st];
  i0 += st;
  for (int i = 2; i < n - 2; i++) {
    const float tmp3 = ib[i0];
    ob[i0] = +w1 * (ib[i0 + st] - tmp2) + w2 * (ib[i0 + 2 * st] - tmp1);
    i0 += st;
    tmp1 = tmp2;
    tmp2 = tmp3;
  }
  const float tmp3 = ib[i0];
  ob[i0] = w1 * (ib[i0 + st] - tmp2) - w2 * tmp1;
  i0 += st;
  ob[i0] = -w1 * tmp3 - w2 * tmp2;
}

TDK_API void oracle_bilateral(const float* in, float* out, int width, int height, float sigma_s, float sigma_r, float detail) {
  int size[3];
  oracle_bilateral_grid_size(width, height, sigma_s, sigma_r, size);
  const size_t ncell = (size_t)size[0] * size[1] * size[2];
  float* grid = (float*)calloc(ncell, sizeof(float));
  float* tmp = (float*)calloc(ncell, sizeof(float));
  const ptrdiff_t sx = size[0], sxy = (ptrdiff_t)size[0] * size[1];

  const float contrib = 1.0f / (sigma_s * sigma_s);
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      const gsample s = make_sample(x, y, in[(size_t)y * width + x], size, sigma_s, sigma_r);
      const float ax = 1.0f - s.fx, ay = 1.0f - s.fy, az = 1.0f - s.fz;
      const float bx = s.fx, by = s.fy, bz = s.fz;
      float* g = grid + s.gi;
      g[0] += ax * ay * az * contrib;
      g[s.ox] += bx * ay * az * contrib;
      g[s.oy] += ax * by * az * contrib;
      g[s.oy + s.ox] += bx * by * az * contrib;
      g[s.oz] += ax * ay * bz * contrib;
      g[s.oz + s.ox] += bx * ay * bz * contrib;
      g[s.oz + s.oy] += ax * by * bz * contrib;
      g[s.oz + s.oy + s.ox] += bx * by * bz * contrib;
    }

#pragma omp parallel for schedule(static)
  for (int z = 0; z < size[2]; z++)
    for (int y = 0; y < size[1]; y++) blur_line(grid + z * sxy + y * sx, tmp + z * sxy + y * sx, size[0], 1);
#pragma omp parallel for schedule(static)
  for (int z = 0; z < size[2]; z++)
    for (int x = 0; x < size[0]; x++) blur_line(tmp + z * sxy + x, grid + z * sxy + x, size[1], sx);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < size[1]; y++)
    for (int x = 0; x < size[0]; x++) blur_line_z(grid + y * sx + x, tmp + y * sx + x, size[2], sxy);

  const float norm = -detail * sigma_r * 4.0f;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      const float L = in[(size_t)y * width + x];
      const gsample s = make_sample(x, y, L, size, sigma_s, sigma_r);
      const float ax = 1.0f - s.fx, ay = 1.0f - s.fy, az = 1.0f - s.fz;
      const float bx = s.fx, by = s.fy, bz = s.fz;
      const float* g = tmp + s.gi;
      const float Ldiff = g[0] * ax * ay * az + g[s.ox] * bx * ay * az + g[s.oy] * ax * by * az + g[s.oy + s.ox] * bx * by * az +
                          g[s.oz] * ax * ay * bz + g[s.oz + s.ox] * bx * ay * bz + g[s.oz + s.oy] * ax * by * bz +
                          g[s.oz + s.oy + s.ox] * bx * by * bz;
      out[(size_t)y * width + x] = fmaxf(0.0f, L + norm * Ldiff);
    }
  free(grid);
  free(tmp);
}
