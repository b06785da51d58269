/*
 * oracle/src/color_ops.c -- per-pixel colour operators, tonemaps and image statistics
 * (CPU oracle, test infrastructure only).
 *
 * Colour ops     : reference csrc/color_conversions.cu:12-314 (math: device_conversions.h)
 * Tonemaps       : csrc/tonemap/{reinhard.cu:17-45, aces.cu:13-89, linear.cu:13-40},
 *                  adaptation csrc/tonemap/color_adaption.h:17-76,
 *                  vibrance / Lab from device_color_conversions.h, u8 store
 *                  device_math.h:347-349,393-397
 * Bounds/metrics : csrc/tonemap/color_adaption.cu:12-166
 */
#include "color.h"

enum { OP_RGB2XYZ = 0, OP_XYZ2LAB, OP_LAB2XYZ, OP_XYZ2RGB, OP_RGB2LAB, OP_LAB2RGB, OP_HSL, OP_VIBRANCE, OP_MAT3 };

/* params: HSL -> {hue, sat, lum}; VIBRANCE -> {amount}; MAT3 -> 9 row-major entries */
TDK_API void oracle_color_op(const float* in, float* out, int64_t npix, int op, const float* params) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < npix; i++) {
    const vec3 p = v3(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
    vec3 r = p;
    switch (op) {
      case OP_RGB2XYZ: r = cA_rgb_to_xyz(p); break;
      case OP_XYZ2LAB: r = cA_xyz_to_lab(p); break;
      case OP_LAB2XYZ: r = cA_lab_to_xyz(p); break;
      case OP_XYZ2RGB: r = cA_xyz_to_rgb(p); break;
      case OP_RGB2LAB: r = cA_rgb_to_lab(p); break;
      case OP_LAB2RGB: r = cA_lab_to_rgb(p); break;
      case OP_HSL: r = cA_modify_hsl(p, params[0], params[1], params[2]); break;
      case OP_VIBRANCE: r = cA_vibrance(p, params[0]); break;
      case OP_MAT3: r = v3_clip(mat3_mul(params, p)); break; /* device_conversions.h:209-211 */
    }
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
  }
}

/* color_conversions.cu:168-183 */
TDK_API void oracle_compute_luminance(const float* in, float* out, int64_t npix, int log_mode, float eps) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < npix; i++) {
    const float lum = cA_rgb_to_lab_l(v3_clip(v3(in[3 * i], in[3 * i + 1], in[3 * i + 2])));
    out[i] = log_mode ? logf(fmaxf(eps, lum)) : lum;
  }
}

/* color_conversions.cu:240-254 */
TDK_API void oracle_modify_luminance(const float* in, const float* lum, float* out, int64_t npix, int log_mode) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < npix; i++) {
    const vec3 p = v3(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
    const vec3 r = log_mode ? cA_modify_log_luminance(p, lum[i]) : cA_modify_luminance(p, lum[i]);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
  }
}

/* ------------------------------------------------------------------ tonemaps */
enum { TM_REINHARD = 0, TM_ACES, TM_ACES_ADAPTIVE, TM_LINEAR };

/* device_math.h:347-349 */
static inline uint8_t to_u8(float x) { return (uint8_t)fminf(roundf(x * 255.0f), 255.0f); }

/* color_adaption.h:17-29 */
static inline float map_key_of(float log_mean) {
  const float normalized = fmaxf(0.0f, fminf(1.0f, (-log_mean) / 9.21034f));
  return 0.3f + 0.7f * powf(normalized, 1.4f);
}

/* aces.cu:13-34; the double literals there narrow to float at the float3 operators */
static inline float rrt_odt(float v) {
  const float a = v * (v + 0.0245786f) - 0.000090537f;
  const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
  return a / b;
}
static inline vec3 aces_fit(vec3 rgb) {
  static const float in_m[9] = {0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f};
  static const float out_m[9] = {1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f, -0.00327f, -0.07276f, 1.07602f};
  const vec3 a = mat3_mul(in_m, rgb);
  return mat3_mul(out_m, v3(rrt_odt(a.x), rrt_odt(a.y), rrt_odt(a.z)));
}

/* float variant exposes the pre-quantisation value so tests can bound the u8 +-1 LSB cases */
static inline vec3 tonemap_pixel(vec3 rgb, int mode, const float* metrics, float gamma, float intensity, float light_adapt, float vibrance) {
  vec3 tm;
  if (mode == TM_ACES) {
    const float s = powf(2.0f, intensity);
    tm = aces_fit(v3(rgb.x * s, rgb.y * s, rgb.z * s));
  } else {
    const float key = map_key_of(metrics[0]);
    const float exposure = expf(intensity);
    const vec3 mean = v3(f_lerp(light_adapt, metrics[2], rgb.x), f_lerp(light_adapt, metrics[3], rgb.y), f_lerp(light_adapt, metrics[4], rgb.z));
    const vec3 adapt = v3(powf(mean.x / exposure, key), powf(mean.y / exposure, key), powf(mean.z / exposure, key));
    if (mode == TM_REINHARD) tm = v3(rgb.x / (adapt.x + rgb.x), rgb.y / (adapt.y + rgb.y), rgb.z / (adapt.z + rgb.z));
    else if (mode == TM_LINEAR) tm = v3(rgb.x / adapt.x, rgb.y / adapt.y, rgb.z / adapt.z);
    else tm = aces_fit(v3(rgb.x / adapt.x, rgb.y / adapt.y, rgb.z / adapt.z));
  }
  const float ig = 1.0f / gamma;
  const vec3 g = v3(powf(fmaxf(tm.x, 0.0f), ig), powf(fmaxf(tm.y, 0.0f), ig), powf(fmaxf(tm.z, 0.0f), ig));
  vec3 o = cB_vibrance(g, vibrance);
  if (mode == TM_LINEAR) o = v3_clip(o);
  return o;
}

TDK_API void oracle_tonemap(const float* in, uint8_t* out_u8, float* out_f32, int64_t npix, int mode, const float* metrics,
                            float gamma, float intensity, float light_adapt, float vibrance) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < npix; i++) {
    const vec3 o = tonemap_pixel(v3(in[3 * i], in[3 * i + 1], in[3 * i + 2]), mode, metrics, gamma, intensity, light_adapt, vibrance);
    if (out_u8) { out_u8[3 * i] = to_u8(o.x); out_u8[3 * i + 1] = to_u8(o.y); out_u8[3 * i + 2] = to_u8(o.z); }
    if (out_f32) { out_f32[3 * i] = o.x; out_f32[3 * i + 1] = o.y; out_f32[3 * i + 2] = o.z; }
  }
}

/* ------------------------------------------------------------------ statistics */
/* color_adaption.cu:12-36,90-120: sample grid x = i*stride, y = j*stride; bounds start at
 * (FLT_MAX, -FLT_MAX) and accumulate over successive images. */
TDK_API void oracle_image_bounds(const float* img, int w, int h, int stride, float bounds[2]) {
  float lo = bounds[0], hi = bounds[1];
  for (int y = 0; y < h; y += stride)
    for (int x = 0; x < w; x += stride) {
      const float* p = img + ((size_t)y * w + x) * 3;
      lo = fminf(lo, fminf(fminf(p[0], p[1]), p[2]));
      hi = fmaxf(hi, fmaxf(fmaxf(p[0], p[1]), p[2]));
    }
  bounds[0] = lo; bounds[1] = hi;
}

/* color_adaption.cu:39-84: accumulates the six sums in fp64 (the reference's atomic order
 * is nondeterministic; fp64 is the order-free yardstick).  acc = {log, gray, r, g, b, valid} */
TDK_API void oracle_image_metrics_accumulate(const float* img, int w, int h, int stride, float min_gray, const float bounds[2], double acc[6]) {
  const float eps = 1e-6f;
  const float range = bounds[1] - bounds[0] + eps;
  for (int y = 0; y < h; y += stride)
    for (int x = 0; x < w; x += stride) {
      const float* p = img + ((size_t)y * w + x) * 3;
      const float sx = (p[0] - bounds[0]) / range, sy = (p[1] - bounds[0]) / range, sz = (p[2] - bounds[0]) / range;
      const int saturated = (sx >= 0.99f || sy >= 0.99f || sz >= 0.99f);
      const float mask = saturated ? 0.0f : 1.0f;
      const float gray = sx * 0.299f + sy * 0.587f + sz * 0.114f;
      const float log_gray = logf(fmaxf(gray, min_gray));
      acc[0] += (double)(log_gray * mask);
      acc[1] += (double)(gray * mask);
      acc[2] += (double)(sx * mask);
      acc[3] += (double)(sy * mask);
      acc[4] += (double)(sz * mask);
      acc[5] += (double)mask;
    }
}

/* color_adaption.cu:161-165 */
TDK_API void oracle_image_metrics_finish(const double acc[6], float metrics[5]) {
  const float valid = (float)acc[5];
  const float norm = 1.0f / fmaxf(valid, 1.0f);
  for (int k = 0; k < 5; k++) metrics[k] = (float)acc[k] * norm;
}
