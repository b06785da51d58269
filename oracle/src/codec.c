/*
 * oracle/src/codec.c -- 12-bit packed raw codec (CPU oracle, test infrastructure only).
 *
 * Follows reference csrc/packed.cu:8-31 (bit layouts) and :34-155 (kernels):
 *   standard: b0 = p0 & 0xff; b1 = (p1 & 0xf) << 4 | p0 >> 8; b2 = p1 >> 4
 *   IDS     : b0 = p0 >> 4;   b1 = p1 >> 4;   b2 = (p0 & 0xf) << 4 | (p1 & 0xf)
 * float encode: p = min(uint16(roundf(f * scale)), 4095), scale = 4095 when `scaled`
 * float decode: f = float(p) * scale, scale = 1/4095 when `scaled` (packed.cu:196,218)
 */
#include "common.h"

static inline void pack_pair(uint16_t p0, uint16_t p1, int ids, uint8_t* o) {
  if (ids) {
    o[0] = (uint8_t)(p0 >> 4);
    o[1] = (uint8_t)(p1 >> 4);
    o[2] = (uint8_t)(((p0 & 0xf) << 4) | (p1 & 0xf));
  } else {
    o[0] = (uint8_t)(p0 & 0xff);
    o[1] = (uint8_t)(((p1 & 0xf) << 4) | (p0 >> 8));
    o[2] = (uint8_t)(p1 >> 4);
  }
}

/* Decode is restated literally from packed.cu:14-17 (standard) and :27-31 (IDS).
 * NB the reference's IDS decoder takes p0's low nibble from the LOW nibble of byte 2
 * and p1's from the HIGH nibble, while its IDS encoder (packed.cu:20-24) stores p0's
 * low nibble in the HIGH nibble: in the reference, IDS decode(encode(x)) swaps the
 * low nibbles of each pair.  Both directions are kept as the reference has them. */
static inline void unpack_any(const uint8_t* i, int ids, uint16_t* p0, uint16_t* p1) {
  if (ids) {
    *p0 = (uint16_t)(((uint16_t)i[0] << 4) | ((uint16_t)i[2] & 0xf));
    *p1 = (uint16_t)(((uint16_t)i[1] << 4) | ((uint16_t)i[2] >> 4));
  } else {
    *p0 = (uint16_t)((((uint16_t)i[1] & 0xf) << 8) | (uint16_t)i[0]);
    *p1 = (uint16_t)(((uint16_t)i[2] << 4) | ((uint16_t)i[1] >> 4));
  }
}

static inline uint16_t min_u16(uint16_t a, uint16_t b) { return a < b ? a : b; }

TDK_API void oracle_encode12_u16(const uint16_t* in, uint8_t* out, int64_t num_pairs, int ids) {
  for (int64_t k = 0; k < num_pairs; k++) {
    uint16_t p0 = min_u16(in[2 * k], 4095), p1 = min_u16(in[2 * k + 1], 4095);
    pack_pair(p0, p1, ids, out + 3 * k);
  }
}

TDK_API void oracle_encode12_f32(const float* in, uint8_t* out, int64_t num_pairs, int ids, int scaled) {
  const float scale = scaled ? 4095.0f : 1.0f;
  for (int64_t k = 0; k < num_pairs; k++) {
    /* packed.cu:72-77: min(uint16_t(roundf(f)), 4095).  The CUDA float->u16 conversion
     * saturates (negatives and NaN -> 0, large -> 65535), so the composite is a clamp of
     * the rounded value to [0, 4095]; restated as such to stay defined in C. */
    float f0 = in[2 * k] * scale, f1 = in[2 * k + 1] * scale;
    uint16_t p0 = (uint16_t)fminf(fmaxf(roundf(f0), 0.0f), 4095.0f);
    uint16_t p1 = (uint16_t)fminf(fmaxf(roundf(f1), 0.0f), 4095.0f);
    pack_pair(p0, p1, ids, out + 3 * k);
  }
}

TDK_API void oracle_decode12_f32(const uint8_t* in, float* out, int64_t num_pairs, int ids, int scaled) {
  const float scale = scaled ? (1.0f / 4095.0f) : 1.0f;
  for (int64_t k = 0; k < num_pairs; k++) {
    uint16_t p0, p1;
    unpack_any(in + 3 * k, ids, &p0, &p1);
    out[2 * k] = (float)p0 * scale;
    out[2 * k + 1] = (float)p1 * scale;
  }
}

TDK_API void oracle_decode12_f16(const uint8_t* in, uint16_t* out_bits, int64_t num_pairs, int ids, int scaled) {
  const float scale = scaled ? (1.0f / 4095.0f) : 1.0f;
  for (int64_t k = 0; k < num_pairs; k++) {
    uint16_t p0, p1;
    unpack_any(in + 3 * k, ids, &p0, &p1);
    out_bits[2 * k] = f32_to_f16_bits((float)p0 * scale);
    out_bits[2 * k + 1] = f32_to_f16_bits((float)p1 * scale);
  }
}

TDK_API void oracle_decode12_u16(const uint8_t* in, uint16_t* out, int64_t num_pairs, int ids) {
  for (int64_t k = 0; k < num_pairs; k++) unpack_any(in + 3 * k, ids, out + 2 * k, out + 2 * k + 1);
}
