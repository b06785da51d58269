/*
 * oracle/src/color.h -- the reference's TWO colour-math headers, restated side by side.
 * CPU oracle, test infrastructure only.
 *
 *   cA_* : csrc/device_conversions.h        (used by color_conversions.cu: public colour
 *          ops, luminance extract / replace)
 *   cB_* : csrc/device_color_conversions.h  (used by the tonemap kernels: vibrance)
 *
 * They share function names in the reference but differ in lab_f / lab_f_inv, clamping
 * and HSL handling, so both are kept.
 */
#ifndef TDK_ORACLE_COLOR_H
#define TDK_ORACLE_COLOR_H

#include "common.h"

typedef struct { float x, y, z; } vec3;

static inline vec3 v3(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
static inline vec3 v3_clip(vec3 a) { return v3(f_clip01(a.x), f_clip01(a.y), f_clip01(a.z)); }

/* row-major 3x3 times vector, device_math.h:108-114 */
static inline vec3 mat3_mul(const float m[9], vec3 v) {
  return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

static const float M_RGB2XYZ[9] = {0.4124564f, 0.3575761f, 0.1804375f, 0.2126729f, 0.7151522f, 0.0721750f, 0.0193339f, 0.1191920f, 0.9503041f};
static const float M_XYZ2RGB[9] = {3.2404542f, -1.5371385f, -0.4985314f, -0.9692660f, 1.8760108f, 0.0415560f, 0.0556434f, -0.2040259f, 1.0572252f};
static const float D65[3] = {0.95047f, 1.0f, 1.08883f};

/* ------------------------------------------------------------------ header A */
/* device_conversions.h:12-23: pow((rgb + a) / (1 + a), 2.4) if rgb > 0.04045 else rgb * (1/12.92) */
static inline float cA_srgb_to_linear1(float c) {
  const float a = 0.055f;
  const float lin = c * (1.0f / 12.92f);
  return (c > 0.04045f) ? powf((c + a) / (1.0f + a), 2.4f) : lin;
}
static inline vec3 cA_srgb_to_linear(vec3 c) { return v3(cA_srgb_to_linear1(c.x), cA_srgb_to_linear1(c.y), cA_srgb_to_linear1(c.z)); }

/* device_conversions.h:25-36 */
static inline float cA_linear_to_srgb1(float c) {
  const float a = 0.055f;
  return (c > 0.0031308f) ? ((1.0f + a) * powf(c, 1.0f / 2.4f) - a) : c * 12.92f;
}
static inline vec3 cA_linear_to_srgb(vec3 c) { return v3(cA_linear_to_srgb1(c.x), cA_linear_to_srgb1(c.y), cA_linear_to_srgb1(c.z)); }

/* device_conversions.h:39-53 */
static inline float cA_lab_f(float t) { return (t > 0.008856f) ? powf(t, 1.0f / 3.0f) : (t * 7.787f + 16.0f / 116.0f); }
/* device_conversions.h:60-71 (vector form; the scalar form at :55-58 is equivalent) */
static inline float cA_lab_f_inv(float t) {
  const float t3 = t * t * t;
  return (t3 > 0.008856f) ? t3 : (t - 16.0f / 116.0f) / 7.787f;
}

static inline vec3 cA_rgb_to_xyz(vec3 rgb) { return mat3_mul(M_RGB2XYZ, cA_srgb_to_linear(rgb)); }

/* device_conversions.h:87-99 */
static inline vec3 cA_xyz_to_lab(vec3 xyz) {
  const float fx = cA_lab_f(xyz.x / D65[0]), fy = cA_lab_f(xyz.y / D65[1]), fz = cA_lab_f(xyz.z / D65[2]);
  return v3((116.0f / 100.0f) * fy - (16.0f / 100.0f), (500.0f / 128.0f) * (fx - fy), (200.0f / 128.0f) * (fy - fz));
}

/* device_conversions.h:101-114 */
static inline vec3 cA_lab_to_xyz(vec3 lab) {
  const float fy = lab.x * (100.0f / 116.0f) + (16.0f / 116.0f);
  const float fx = lab.y * (128.0f / 500.0f) + fy;
  const float fz = fy - lab.z * (128.0f / 200.0f);
  return v3(cA_lab_f_inv(fx) * D65[0], cA_lab_f_inv(fy) * D65[1], cA_lab_f_inv(fz) * D65[2]);
}

static inline vec3 cA_xyz_to_rgb(vec3 xyz) { return cA_linear_to_srgb(mat3_mul(M_XYZ2RGB, xyz)); }
static inline vec3 cA_rgb_to_lab(vec3 rgb) { return cA_xyz_to_lab(cA_rgb_to_xyz(rgb)); }
static inline vec3 cA_lab_to_rgb(vec3 lab) { return cA_xyz_to_rgb(cA_lab_to_xyz(lab)); }

/* device_conversions.h:197-207 */
static inline float cA_rgb_to_lab_l(vec3 rgb) {
  const vec3 lin = cA_srgb_to_linear(rgb);
  const float y = 0.2126729f * lin.x + 0.7151522f * lin.y + 0.0721750f * lin.z;
  return fmaxf(0.0f, (116.0f / 100.0f) * cA_lab_f(y) - (16.0f / 100.0f));
}

/* device_conversions.h:213-225 (eps is ignored on the modify side) */
static inline vec3 cA_modify_luminance(vec3 rgb, float lum) {
  const vec3 lab = cA_rgb_to_lab(rgb);
  return v3_clip(cA_lab_to_rgb(v3(fmaxf(0.0f, fminf(1.0f, lum)), lab.y, lab.z)));
}
static inline vec3 cA_modify_log_luminance(vec3 rgb, float log_lum) {
  const vec3 lab = cA_rgb_to_lab(rgb);
  return v3_clip(cA_lab_to_rgb(v3(fmaxf(0.0f, fminf(1.0f, expf(log_lum))), lab.y, lab.z)));
}

/* device_conversions.h:146-195 */
static inline vec3 cA_rgb_to_hsl(vec3 c) {
  const float mx = fmaxf(fmaxf(c.x, c.y), c.z), mn = fminf(fminf(c.x, c.y), c.z);
  const float delta = mx - mn;
  float h = 0.0f, s = 0.0f;
  const float l = (mx + mn) * 0.5f;
  if (delta > 1e-6f) {
    s = (l < 0.5f) ? delta / (mx + mn) : delta / (2.0f - mx - mn);
    if (mx == c.x) h = (c.y - c.z) / delta + (c.y < c.z ? 6.0f : 0.0f);
    else if (mx == c.y) h = (c.z - c.x) / delta + 2.0f;
    else h = (c.x - c.y) / delta + 4.0f;
    h /= 6.0f;
  }
  return v3(h, s, l);
}
static inline float hsl_hue(float p, float q, float t) {
  if (t < 0.0f) t += 1.0f;
  if (t > 1.0f) t -= 1.0f;
  if (t < 1.0f / 6.0f) return p + (q - p) * 6.0f * t;
  if (t < 1.0f / 2.0f) return q;
  if (t < 2.0f / 3.0f) return p + (q - p) * (2.0f / 3.0f - t) * 6.0f;
  return p;
}
static inline vec3 cA_hsl_to_rgb(vec3 hsl) {
  const float h = hsl.x, s = hsl.y, l = hsl.z;
  if (s < 1e-6f) return v3(l, l, l);
  const float q = (l < 0.5f) ? l * (1.0f + s) : l + s - l * s;
  const float p = 2.0f * l - q;
  return v3(hsl_hue(p, q, h + 1.0f / 3.0f), hsl_hue(p, q, h), hsl_hue(p, q, h - 1.0f / 3.0f));
}
/* device_conversions.h:227-239 */
static inline vec3 cA_modify_hsl(vec3 rgb, float hue, float sat, float lum) {
  const vec3 hsl = cA_rgb_to_hsl(rgb);
  float nh = hsl.x + hue;
  if (nh < 0.0f) nh += 1.0f;
  if (nh > 1.0f) nh -= 1.0f;
  const float ns = powf(hsl.y, 1.0f / (1.0f + sat));
  const float nl = powf(hsl.z, 1.0f / (1.0f + lum));
  return v3_clip(cA_hsl_to_rgb(v3(nh, ns, nl)));
}
/* device_conversions.h:242-261 */
static inline vec3 cA_vibrance(vec3 rgb, float amount) {
  const vec3 lab = cA_rgb_to_lab(rgb);
  const float chroma = sqrtf(lab.y * lab.y + lab.z * lab.z);
  const float ls = 1.0f - amount * chroma * 0.25f;
  const float ss = 1.0f + amount * chroma;
  return v3_clip(cA_lab_to_rgb(v3(lab.x * ls, lab.y * ss, lab.z * ss)));
}

/* ------------------------------------------------------------------ header B */
/* device_color_conversions.h:7-21 */
static inline float cB_linear_to_srgb1(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f; }
static inline float cB_srgb_to_linear1(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
/* device_color_conversions.h:35-51 */
static inline float cB_lab_f(float t) {
  const float delta = 6.0f / 29.0f;
  const float delta_cubed = delta * delta * delta;
  const float factor = 1.0f / (3.0f * delta * delta);
  const float offset = 4.0f / 29.0f;
  return (t > delta_cubed) ? cbrtf(t) : factor * t + offset;
}
static inline float cB_lab_f_inv(float t) {
  const float delta = 6.0f / 29.0f;
  const float factor = 3.0f * delta * delta;
  const float offset = 4.0f / 29.0f;
  return (t > delta) ? (t * t * t) : factor * (t - offset);
}
/* device_color_conversions.h:23-33, 53-66 */
static inline vec3 cB_rgb_to_lab(vec3 rgb) {
  const vec3 lin = v3(cB_srgb_to_linear1(rgb.x), cB_srgb_to_linear1(rgb.y), cB_srgb_to_linear1(rgb.z));
  const vec3 xyz = mat3_mul(M_RGB2XYZ, lin);
  const float fx = cB_lab_f(xyz.x / D65[0]), fy = cB_lab_f(xyz.y / D65[1]), fz = cB_lab_f(xyz.z / D65[2]);
  const float L = 116.0f * fy - 16.0f, a = 500.0f * (fx - fy), b = 200.0f * (fy - fz);
  return v3(L / 100.0f, a / 128.0f, b / 128.0f);
}
/* device_color_conversions.h:68-95 */
static inline vec3 cB_lab_to_rgb(vec3 lab) {
  const float L = lab.x * 100.0f, a = lab.y * 128.0f, b = lab.z * 128.0f;
  const float fy = (L + 16.0f) / 116.0f;
  const float fx = a / 500.0f + fy;
  const float fz = fy - b / 200.0f;
  const vec3 xyz = v3(cB_lab_f_inv(fx) * D65[0], cB_lab_f_inv(fy) * D65[1], cB_lab_f_inv(fz) * D65[2]);
  const vec3 lin = mat3_mul(M_XYZ2RGB, xyz);
  return v3(cB_linear_to_srgb1(lin.x), cB_linear_to_srgb1(lin.y), cB_linear_to_srgb1(lin.z));
}
/* device_color_conversions.h:199-213 */
static inline vec3 cB_vibrance(vec3 rgb, float amount) {
  const vec3 lab = cB_rgb_to_lab(rgb);
  const float chroma = sqrtf(lab.y * lab.y + lab.z * lab.z);
  const float ls = 1.0f - amount * chroma * 0.25f;
  const float ss = 1.0f + amount * chroma;
  return v3_clip(cB_lab_to_rgb(v3(lab.x * ls, lab.y * ss, lab.z * ss)));
}

#endif
