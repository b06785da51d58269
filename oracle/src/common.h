/*
 * oracle/src/common.h -- shared helpers for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This directory restates, in plain strict-IEEE C,
 * the arithmetic of the reference's hot-path kernels (uc-vision/torch-darktable
 * @ 2025-11-14, csrc/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (torch-darktable_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" for every kernel op -- the reference ships
 * no golden vectors or known-answer tests (its only test is a pydantic
 * round-trip) and its CUDA build cannot be compiled or run here.  The oracle is
 * pinned instead by source-derived identities (tests/test_oracle_kat.py) and,
 * for the two pure-torch helpers that are importable (rgb_to_bayer,
 * estimate_channel_noise), by fixtures in tests/golden/.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  No
 * FMA contraction, IEEE division, libm transcendentals.
 */
#ifndef TDK_ORACLE_COMMON_H
#define TDK_ORACLE_COMMON_H

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TDK_API __attribute__((visibility("default")))

/* Bayer pattern words: reference csrc/debayer/demosaic.h:7-12 */
#define TDK_RGGB 0x94949494u
#define TDK_BGGR 0x16161616u
#define TDK_GRBG 0x61616161u
#define TDK_GBRG 0x49494949u

/* CFA colour of (row, col): 0=R 1=G 2=B.  reference csrc/debayer/bayer_device.h:9-11 */
static inline int cfa_color(int row, int col, uint32_t pattern) {
  return (int)((pattern >> ((((row << 1) & 14) + (col & 1)) << 1)) & 3u);
}

static inline float f_clamp(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline float f_clip01(float v) { return f_clamp(v, 0.0f, 1.0f); }
static inline float f_sq(float v) { return v * v; }
/* reference csrc/device_math.h:80-82 */
static inline float f_mix(float a, float b, float t) { return (1.0f - t) * a + t * b; }
/* reference csrc/device_math.h:407-409: lerp(t, a, b) */
static inline float f_lerp(float t, float a, float b) { return a + t * (b - a); }

/* IEEE binary16 <-> binary32, round-to-nearest-even (matches at::Half / __float2half). */
static inline uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t mant = x & 0x007fffffu;
  int32_t exp = (int32_t)((x >> 23) & 0xff);
  if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0u));
  int32_t e = exp - 127 + 15;
  if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    mant |= 0x00800000u;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t half_m = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m++;
    return (uint16_t)(sign | half_m);
  }
  uint32_t half_m = mant >> 13;
  uint32_t rem = mant & 0x1fffu;
  uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | half_m);
  if (rem > 0x1000u || (rem == 0x1000u && (half_m & 1u))) h++;
  return h;
}

static inline float f16_bits_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t mant = h & 0x3ffu;
  uint32_t x;
  if (exp == 0) {
    if (mant == 0) {
      x = sign;
    } else {
      int e = -1;
      do { e++; mant <<= 1; } while (!(mant & 0x400u));
      mant &= 0x3ffu;
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | (mant << 13);
    }
  } else if (exp == 0x1f) {
    x = sign | 0x7f800000u | (mant << 13);
  } else {
    x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

static inline float round_to_f16(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

#endif
