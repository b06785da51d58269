// tdk_color.h -- device colour math.  The reference carries two divergent headers that
// define the same names; both behaviours are needed, so they live in two namespaces:
//   cA : csrc/device_conversions.h        -> public colour ops, luminance extract / replace
//   cB : csrc/device_color_conversions.h  -> Lab vibrance inside the tonemap kernels
//
// Transcendentals: the reference is built with nvcc --use_fast_math (setup.py:36), i.e. its
// powf / expf / logf are the hardware exp2/log2 approximations, not correctly rounded libm
// calls.  The same class of arithmetic is used here: v_log_f32 / v_exp_f32 (about 1 ulp each),
// pow(x, y) = exp2(y * log2(x)).  Against the libm-based CPU oracle this costs a few 1e-7
// relative (tests use 2e-5 absolute); the full-accuracy device-library calls are ~20x more
// instructions and made every colour kernel ALU-bound (0.4 ms instead of ~0.05 ms per 12 MP
// pass).
#pragma once

#include "tdk_common.h"

// x > 0: exp2(y * log2 x); x == 0: log2 -> -inf -> 0 for y > 0; x < 0 -> NaN (as powf for
// non-integer y).  pow(x, 0) == 1 including x == 0 is kept by the y == 0 test.
__device__ __forceinline__ float tdk_pow(float x, float y) {
  const float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
  return (y == 0.0f) ? 1.0f : r;
}
__device__ __forceinline__ float tdk_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float tdk_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
__device__ __forceinline__ float tdk_cbrt(float x) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 3.0f)); }
// a / b as a * rcp(b) (v_rcp_f32, 1 ulp) -- the reference's fast-math division class; the IEEE
// sequence is ~10 instructions and the colour kernels carry ~18 divisions per pixel.
__device__ __forceinline__ float tdk_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float tdk_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ f3 clip3(f3 a) { return mk3(clip01(a.x), clip01(a.y), clip01(a.z)); }

// row-major 3x3 times vector (reference csrc/device_math.h:108-114)
__device__ __forceinline__ f3 mat3_mul(const float m[9], f3 v) {
  return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

__device__ __forceinline__ f3 rgb_to_xyz_lin(f3 v) {
  return mk3(0.4124564f * v.x + 0.3575761f * v.y + 0.1804375f * v.z, 0.2126729f * v.x + 0.7151522f * v.y + 0.0721750f * v.z,
             0.0193339f * v.x + 0.1191920f * v.y + 0.9503041f * v.z);
}
__device__ __forceinline__ f3 xyz_to_rgb_lin(f3 v) {
  return mk3(3.2404542f * v.x + -1.5371385f * v.y + -0.4985314f * v.z, -0.9692660f * v.x + 1.8760108f * v.y + 0.0415560f * v.z,
             0.0556434f * v.x + -0.2040259f * v.y + 1.0572252f * v.z);
}

#define TDK_D65_X 0.95047f
#define TDK_D65_Y 1.0f
#define TDK_D65_Z 1.08883f

// The Lab / sRGB / HSL math below may be contracted to FMAs: its transcendentals are hardware approximations
// already (parity is by tolerance, 2e-5 against the libm oracle), an FMA only removes one rounding, and the
// 3x3 matrices and a * b + c chains are ~20 % of the colour kernels' instructions.  mat3_mul above (the
// bit-exact color_transform_3x3) stays uncontracted.
#pragma clang fp contract(fast)
namespace cA {  // device_conversions.h

__device__ __forceinline__ float srgb_to_linear(float c) {
  const float a = 0.055f;
  const float lin = c * (1.0f / 12.92f);
  return (c > 0.04045f) ? tdk_pow((c + a) * (1.0f / (1.0f + a)), 2.4f) : lin;
}
__device__ __forceinline__ float linear_to_srgb(float c) {
  const float a = 0.055f;
  return (c > 0.0031308f) ? ((1.0f + a) * tdk_pow(c, 1.0f / 2.4f) - a) : c * 12.92f;
}
__device__ __forceinline__ float lab_f(float t) { return (t > 0.008856f) ? tdk_pow(t, 1.0f / 3.0f) : (t * 7.787f + 16.0f / 116.0f); }
__device__ __forceinline__ float lab_f_inv(float t) {
  const float t3 = t * t * t;
  return (t3 > 0.008856f) ? t3 : (t - 16.0f / 116.0f) * (1.0f / 7.787f);
}
__device__ __forceinline__ f3 rgb_to_xyz(f3 c) { return rgb_to_xyz_lin(mk3(srgb_to_linear(c.x), srgb_to_linear(c.y), srgb_to_linear(c.z))); }
__device__ __forceinline__ f3 xyz_to_lab(f3 xyz) {
  const float fx = lab_f(xyz.x * (1.0f / TDK_D65_X)), fy = lab_f(xyz.y), fz = lab_f(xyz.z * (1.0f / TDK_D65_Z));
  return mk3((116.0f / 100.0f) * fy - (16.0f / 100.0f), (500.0f / 128.0f) * (fx - fy), (200.0f / 128.0f) * (fy - fz));
}
__device__ __forceinline__ f3 lab_to_xyz(f3 lab) {
  const float fy = lab.x * (100.0f / 116.0f) + (16.0f / 116.0f);
  const float fx = lab.y * (128.0f / 500.0f) + fy;
  const float fz = fy - lab.z * (128.0f / 200.0f);
  return mk3(lab_f_inv(fx) * TDK_D65_X, lab_f_inv(fy) * TDK_D65_Y, lab_f_inv(fz) * TDK_D65_Z);
}
__device__ __forceinline__ f3 xyz_to_rgb(f3 xyz) {
  const f3 l = xyz_to_rgb_lin(xyz);
  return mk3(linear_to_srgb(l.x), linear_to_srgb(l.y), linear_to_srgb(l.z));
}
__device__ __forceinline__ f3 rgb_to_lab(f3 c) { return xyz_to_lab(rgb_to_xyz(c)); }
__device__ __forceinline__ f3 lab_to_rgb(f3 c) { return xyz_to_rgb(lab_to_xyz(c)); }

// device_conversions.h:197-207
__device__ __forceinline__ float rgb_to_lab_l(f3 c) {
  const float lx = srgb_to_linear(c.x), ly = srgb_to_linear(c.y), lz = srgb_to_linear(c.z);
  const float y = 0.2126729f * lx + 0.7151522f * ly + 0.0721750f * lz;
  return fmaxf(0.0f, (116.0f / 100.0f) * lab_f(y) - (16.0f / 100.0f));
}
// device_conversions.h:213-225 (the log variant ignores eps)
__device__ __forceinline__ f3 modify_luminance(f3 rgb, float lum) {
  const f3 lab = rgb_to_lab(rgb);
  return clip3(lab_to_rgb(mk3(fmaxf(0.0f, fminf(1.0f, lum)), lab.y, lab.z)));
}
__device__ __forceinline__ f3 modify_log_luminance(f3 rgb, float log_lum) {
  const f3 lab = rgb_to_lab(rgb);
  return clip3(lab_to_rgb(mk3(fmaxf(0.0f, fminf(1.0f, tdk_exp(log_lum))), lab.y, lab.z)));
}

__device__ __forceinline__ f3 rgb_to_hsl(f3 c) {
  const float mx = fmaxf(fmaxf(c.x, c.y), c.z), mn = fminf(fminf(c.x, c.y), c.z);
  const float delta = mx - mn;
  float h = 0.0f, s = 0.0f;
  const float l = (mx + mn) * 0.5f;
  if (delta > 1e-6f) {
    s = (l < 0.5f) ? tdk_div(delta, mx + mn) : tdk_div(delta, 2.0f - mx - mn);
    if (mx == c.x) h = tdk_div(c.y - c.z, delta) + (c.y < c.z ? 6.0f : 0.0f);
    else if (mx == c.y) h = tdk_div(c.z - c.x, delta) + 2.0f;
    else h = tdk_div(c.x - c.y, delta) + 4.0f;
    h *= (1.0f / 6.0f);
  }
  return mk3(h, s, l);
}
__device__ __forceinline__ float hsl_hue(float p, float q, float t) {
  if (t < 0.0f) t += 1.0f;
  if (t > 1.0f) t -= 1.0f;
  if (t < 1.0f / 6.0f) return p + (q - p) * 6.0f * t;
  if (t < 1.0f / 2.0f) return q;
  if (t < 2.0f / 3.0f) return p + (q - p) * (2.0f / 3.0f - t) * 6.0f;
  return p;
}
__device__ __forceinline__ f3 hsl_to_rgb(f3 hsl) {
  const float h = hsl.x, s = hsl.y, l = hsl.z;
  if (s < 1e-6f) return mk3(l, l, l);
  const float q = (l < 0.5f) ? l * (1.0f + s) : l + s - l * s;
  const float p = 2.0f * l - q;
  return mk3(hsl_hue(p, q, h + 1.0f / 3.0f), hsl_hue(p, q, h), hsl_hue(p, q, h - 1.0f / 3.0f));
}
__device__ __forceinline__ f3 modify_hsl(f3 rgb, float hue, float sat, float lum) {
  const f3 hsl = rgb_to_hsl(rgb);
  float nh = hsl.x + hue;
  if (nh < 0.0f) nh += 1.0f;
  if (nh > 1.0f) nh -= 1.0f;
  const float ns = tdk_pow(hsl.y, tdk_div(1.0f, 1.0f + sat));
  const float nl = tdk_pow(hsl.z, tdk_div(1.0f, 1.0f + lum));
  return clip3(hsl_to_rgb(mk3(nh, ns, nl)));
}
__device__ __forceinline__ f3 vibrance(f3 rgb, float amount) {
  const f3 lab = rgb_to_lab(rgb);
  const float chroma = tdk_sqrt(lab.y * lab.y + lab.z * lab.z);
  const float ls = 1.0f - amount * chroma * 0.25f;
  const float ss = 1.0f + amount * chroma;
  return clip3(lab_to_rgb(mk3(lab.x * ls, lab.y * ss, lab.z * ss)));
}

}  // namespace cA

namespace cB {  // device_color_conversions.h

__device__ __forceinline__ float linear_to_srgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * tdk_pow(c, 1.0f / 2.4f) - 0.055f; }
__device__ __forceinline__ float srgb_to_linear(float c) { return c <= 0.04045f ? c * (1.0f / 12.92f) : tdk_pow((c + 0.055f) * (1.0f / 1.055f), 2.4f); }
__device__ __forceinline__ float lab_f(float t) {
  const float delta = 6.0f / 29.0f;
  const float delta_cubed = delta * delta * delta;
  const float factor = 1.0f / (3.0f * delta * delta);
  const float offset = 4.0f / 29.0f;
  return (t > delta_cubed) ? tdk_cbrt(t) : factor * t + offset;
}
__device__ __forceinline__ float lab_f_inv(float t) {
  const float delta = 6.0f / 29.0f;
  const float factor = 3.0f * delta * delta;
  const float offset = 4.0f / 29.0f;
  return (t > delta) ? (t * t * t) : factor * (t - offset);
}
__device__ __forceinline__ f3 rgb_to_lab(f3 c) {
  const f3 xyz = rgb_to_xyz_lin(mk3(srgb_to_linear(c.x), srgb_to_linear(c.y), srgb_to_linear(c.z)));
  const float fx = lab_f(xyz.x * (1.0f / TDK_D65_X)), fy = lab_f(xyz.y), fz = lab_f(xyz.z * (1.0f / TDK_D65_Z));
  const float L = 116.0f * fy - 16.0f, a = 500.0f * (fx - fy), b = 200.0f * (fy - fz);
  return mk3(L * (1.0f / 100.0f), a * (1.0f / 128.0f), b * (1.0f / 128.0f));
}
__device__ __forceinline__ f3 lab_to_rgb(f3 lab) {
  const float L = lab.x * 100.0f, a = lab.y * 128.0f, b = lab.z * 128.0f;
  const float fy = (L + 16.0f) * (1.0f / 116.0f);
  const float fx = a * (1.0f / 500.0f) + fy;
  const float fz = fy - b * (1.0f / 200.0f);
  const f3 lin = xyz_to_rgb_lin(mk3(lab_f_inv(fx) * TDK_D65_X, lab_f_inv(fy) * TDK_D65_Y, lab_f_inv(fz) * TDK_D65_Z));
  return mk3(linear_to_srgb(lin.x), linear_to_srgb(lin.y), linear_to_srgb(lin.z));
}
__device__ __forceinline__ f3 vibrance(f3 rgb, float amount) {
  const f3 lab = rgb_to_lab(rgb);
  const float chroma = tdk_sqrt(lab.y * lab.y + lab.z * lab.z);
  const float ls = 1.0f - amount * chroma * 0.25f;
  const float ss = 1.0f + amount * chroma;
  return clip3(lab_to_rgb(mk3(lab.x * ls, lab.y * ss, lab.z * ss)));
}

}  // namespace cB
#pragma clang fp contract(off)
