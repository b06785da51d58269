// codec.hip -- 12-bit packed raw <-> u16 / f32 / f16.
//
// Replaces reference csrc/packed.cu:34-280.  Bit layouts (packed.cu:8-31):
//   standard: b0 = p0 & 0xff, b1 = (p1 & 0xf) << 4 | p0 >> 8, b2 = p1 >> 4
//   IDS encode: b0 = p0 >> 4, b1 = p1 >> 4, b2 = (p0 & 0xf) << 4 | (p1 & 0xf)
//   IDS decode: p0 = b0 << 4 | (b2 & 0xf), p1 = b1 << 4 | b2 >> 4   (as the reference has it)
//
// MI355X design: pure HBM streaming.  One thread owns 8 pixels = 12 packed bytes = three
// dwords, so a wave reads 768 contiguous bytes and writes 2 KiB (f32) with 16-B stores; the
// reference's one-pair-per-thread byte accesses are replaced by dword traffic.  The bulk
// kernel needs 4-B aligned packed data and 16-B aligned pixels; anything else (and the
// tail) goes through the per-pair kernel.
#include "tdk_common.h"

namespace {

__device__ __forceinline__ void unpack(uint32_t b0, uint32_t b1, uint32_t b2, bool ids, uint32_t& p0, uint32_t& p1) {
  if (ids) {
    p0 = (b0 << 4) | (b2 & 0xfu);
    p1 = (b1 << 4) | (b2 >> 4);
  } else {
    p0 = ((b1 & 0xfu) << 8) | b0;
    p1 = (b2 << 4) | (b1 >> 4);
  }
}

__device__ __forceinline__ void pack(uint32_t p0, uint32_t p1, bool ids, uint32_t& b0, uint32_t& b1, uint32_t& b2) {
  if (ids) {
    b0 = p0 >> 4;
    b1 = p1 >> 4;
    b2 = ((p0 & 0xfu) << 4) | (p1 & 0xfu);
  } else {
    b0 = p0 & 0xffu;
    b1 = ((p1 & 0xfu) << 4) | (p0 >> 8);
    b2 = p1 >> 4;
  }
}

// float -> 12-bit code: min(u16(roundf(f)), 4095) with CUDA's saturating conversion
// (packed.cu:72-77) == clamp(roundf(f), 0, 4095); NaN -> 0.
__device__ __forceinline__ uint32_t quant12(float f) { return (uint32_t)fminf(fmaxf(roundf(f), 0.0f), 4095.0f); }

template <typename OUT> __device__ __forceinline__ OUT cvt_out(uint32_t p, float scale);
template <> __device__ __forceinline__ float cvt_out<float>(uint32_t p, float scale) { return (float)p * scale; }
template <> __device__ __forceinline__ __half cvt_out<__half>(uint32_t p, float scale) { return __float2half_rn((float)p * scale); }
template <> __device__ __forceinline__ uint16_t cvt_out<uint16_t>(uint32_t p, float) { return (uint16_t)p; }

template <typename IN> __device__ __forceinline__ uint32_t cvt_in(IN v, float scale);
template <> __device__ __forceinline__ uint32_t cvt_in<float>(float v, float scale) { return quant12(v * scale); }
template <> __device__ __forceinline__ uint32_t cvt_in<uint16_t>(uint16_t v, float) { return v < 4095 ? v : 4095u; }

// ---- bulk: 4 pairs (8 pixels, 12 bytes) per thread
template <typename OUT>
__global__ __launch_bounds__(256) void decode12_bulk(const uint32_t* __restrict__ in, OUT* __restrict__ out, int64_t ngroups,
                                                      bool ids, float scale) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t w0 = in[3 * g], w1 = in[3 * g + 1], w2 = in[3 * g + 2];
    uint32_t by[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      by[k] = (w0 >> (8 * k)) & 0xffu;
      by[4 + k] = (w1 >> (8 * k)) & 0xffu;
      by[8 + k] = (w2 >> (8 * k)) & 0xffu;
    }
    OUT px[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t p0, p1;
      unpack(by[3 * k], by[3 * k + 1], by[3 * k + 2], ids, p0, p1);
      px[2 * k] = cvt_out<OUT>(p0, scale);
      px[2 * k + 1] = cvt_out<OUT>(p1, scale);
    }
    if constexpr (sizeof(OUT) == 4) {
      float4* o = reinterpret_cast<float4*>(out) + 2 * g;
      o[0] = *reinterpret_cast<float4*>(&px[0]);
      o[1] = *reinterpret_cast<float4*>(&px[4]);
    } else {
      reinterpret_cast<uint4*>(out)[g] = *reinterpret_cast<uint4*>(&px[0]);
    }
  }
}

template <typename IN>
__global__ __launch_bounds__(256) void encode12_bulk(const IN* __restrict__ in, uint32_t* __restrict__ out, int64_t ngroups,
                                                      bool ids, float scale) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
    IN px[8];
    if constexpr (sizeof(IN) == 4) {
      const float4* i4 = reinterpret_cast<const float4*>(in) + 2 * g;
      *reinterpret_cast<float4*>(&px[0]) = i4[0];
      *reinterpret_cast<float4*>(&px[4]) = i4[1];
    } else {
      *reinterpret_cast<uint4*>(&px[0]) = reinterpret_cast<const uint4*>(in)[g];
    }
    uint32_t by[12];
#pragma unroll
    for (int k = 0; k < 4; k++) pack(cvt_in<IN>(px[2 * k], scale), cvt_in<IN>(px[2 * k + 1], scale), ids, by[3 * k], by[3 * k + 1], by[3 * k + 2]);
    uint32_t w[3];
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = by[4 * k] | (by[4 * k + 1] << 8) | (by[4 * k + 2] << 16) | (by[4 * k + 3] << 24);
    out[3 * g] = w[0];
    out[3 * g + 1] = w[1];
    out[3 * g + 2] = w[2];
  }
}

// ---- per-pair fallback / tail
template <typename OUT>
__global__ __launch_bounds__(256) void decode12_pairs(const uint8_t* __restrict__ in, OUT* __restrict__ out, int64_t first,
                                                       int64_t num_pairs, bool ids, float scale) {
  const int64_t k = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_pairs) return;
  uint32_t p0, p1;
  unpack(in[3 * k], in[3 * k + 1], in[3 * k + 2], ids, p0, p1);
  out[2 * k] = cvt_out<OUT>(p0, scale);
  out[2 * k + 1] = cvt_out<OUT>(p1, scale);
}

template <typename IN>
__global__ __launch_bounds__(256) void encode12_pairs(const IN* __restrict__ in, uint8_t* __restrict__ out, int64_t first,
                                                       int64_t num_pairs, bool ids, float scale) {
  const int64_t k = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_pairs) return;
  uint32_t b0, b1, b2;
  pack(cvt_in<IN>(in[2 * k], scale), cvt_in<IN>(in[2 * k + 1], scale), ids, b0, b1, b2);
  out[3 * k] = (uint8_t)b0;
  out[3 * k + 1] = (uint8_t)b1;
  out[3 * k + 2] = (uint8_t)b2;
}

// ---- decode12_float + apply_white_balance in one pass (the head of the pipeline): 8 pixels per thread, the gain of
// each CFA site picked by its colour, clamp to [0, 1] (white_balance.cu:10-42).  Same arithmetic as the two kernels.
__global__ __launch_bounds__(256) void decode12_wb_bulk(const uint32_t* __restrict__ in, float* __restrict__ out, int64_t ngroups, bool ids, int width,
                                                         uint32_t pattern, const float* __restrict__ gains) {
  const bool wb = gains != nullptr;
  const float gr = wb ? gains[0] : 1.0f, gg = wb ? gains[1] : 1.0f, gb = wb ? gains[2] : 1.0f;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t w0 = in[3 * g], w1 = in[3 * g + 1], w2 = in[3 * g + 2];
    uint32_t by[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      by[k] = (w0 >> (8 * k)) & 0xffu;
      by[4 + k] = (w1 >> (8 * k)) & 0xffu;
      by[8 + k] = (w2 >> (8 * k)) & 0xffu;
    }
    float px[8];
    const int64_t i0 = 8 * g;
    int row = (int)(i0 / width), col = (int)(i0 - (int64_t)row * width);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t p[2];
      unpack(by[3 * k], by[3 * k + 1], by[3 * k + 2], ids, p[0], p[1]);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        float v = (float)p[j] * (1.0f / 4095.0f);
        if (wb) {
          const int c = cfa_color(row, col, pattern);
          v = clampf(v * (c == 0 ? gr : (c == 2 ? gb : gg)), 0.0f, 1.0f);
        }
        px[2 * k + j] = v;
        if (++col == width) { col = 0; row++; }
      }
    }
    float4* o = reinterpret_cast<float4*>(out) + 2 * g;
    o[0] = make_float4(px[0], px[1], px[2], px[3]);
    o[1] = make_float4(px[4], px[5], px[6], px[7]);
  }
}
__global__ __launch_bounds__(256) void decode12_wb_pairs(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t first, int64_t num_pairs, bool ids,
                                                          int width, uint32_t pattern, const float* __restrict__ gains) {
  const int64_t k = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_pairs) return;
  uint32_t p[2];
  unpack(in[3 * k], in[3 * k + 1], in[3 * k + 2], ids, p[0], p[1]);
  for (int j = 0; j < 2; j++) {
    const int64_t i = 2 * k + j;
    const int row = (int)(i / width), col = (int)(i - (int64_t)row * width);
    float v = (float)p[j] * (1.0f / 4095.0f);
    if (gains) {
      const int c = cfa_color(row, col, pattern);
      v = clampf(v * gains[c == 0 ? 0 : (c == 2 ? 2 : 1)], 0.0f, 1.0f);
    }
    out[i] = v;
  }
}

inline int bulk_grid(int64_t ngroups) {
  int64_t b = tdk_div_up64(ngroups, 256);
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

template <typename OUT>
int run_decode(const uint8_t* in, OUT* out, int64_t num_pairs, bool ids, float scale, hipStream_t s, const char* name) {
  TDK_REQUIRE(num_pairs >= 0, "%s: negative length", name);
  if (num_pairs == 0) return TDK_OK;
  TDK_REQUIRE(in && out, "%s: null pointer", name);
  int64_t done = 0;
  if (tdk_aligned(in, 4) && tdk_aligned(out, 16)) {
    const int64_t ngroups = num_pairs / 4;
    if (ngroups > 0) {
      TDK_LAUNCH(name, decode12_bulk<OUT>, dim3(bulk_grid(ngroups)), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(in), out,
                         ngroups, ids, scale);
      done = ngroups * 4;
    }
  }
  if (done < num_pairs) {
    TDK_LAUNCH(name, decode12_pairs<OUT>, dim3((unsigned)tdk_div_up64(num_pairs - done, 256)), dim3(256), 0, s, in, out, done,
                       num_pairs, ids, scale);
  }
  return TDK_OK;
}

template <typename IN>
int run_encode(const IN* in, uint8_t* out, int64_t num_pairs, bool ids, float scale, hipStream_t s, const char* name) {
  TDK_REQUIRE(num_pairs >= 0, "%s: negative length", name);
  if (num_pairs == 0) return TDK_OK;
  TDK_REQUIRE(in && out, "%s: null pointer", name);
  int64_t done = 0;
  if (tdk_aligned(out, 4) && tdk_aligned(in, 16)) {
    const int64_t ngroups = num_pairs / 4;
    if (ngroups > 0) {
      TDK_LAUNCH(name, encode12_bulk<IN>, dim3(bulk_grid(ngroups)), dim3(256), 0, s, in, reinterpret_cast<uint32_t*>(out), ngroups,
                         ids, scale);
      done = ngroups * 4;
    }
  }
  if (done < num_pairs) {
    TDK_LAUNCH(name, encode12_pairs<IN>, dim3((unsigned)tdk_div_up64(num_pairs - done, 256)), dim3(256), 0, s, in, out, done,
                       num_pairs, ids, scale);
  }
  return TDK_OK;
}

}  // namespace

// used by tdk_decode12_wb_rcd (rcd.hip)
int tdk_decode12_wb_plane(const uint8_t* packed, float* mosaic, const float* gains, int width, int height, uint32_t pattern, int ids_format, hipStream_t s) {
  const int64_t num_pairs = (int64_t)width * height / 2;
  int64_t done = 0;
  if (tdk_aligned(packed, 4) && tdk_aligned(mosaic, 16)) {
    const int64_t ngroups = num_pairs / 4;
    if (ngroups > 0) {
      TDK_LAUNCH("tdk_decode12_wb", decode12_wb_bulk, dim3(bulk_grid(ngroups)), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(packed), mosaic, ngroups,
                 ids_format != 0, width, pattern, gains);
      done = ngroups * 4;
    }
  }
  if (done < num_pairs)
    TDK_LAUNCH("tdk_decode12_wb", decode12_wb_pairs, dim3((unsigned)tdk_div_up64(num_pairs - done, 256)), dim3(256), 0, s, packed, mosaic, done, num_pairs,
               ids_format != 0, width, pattern, gains);
  return TDK_OK;
}

TDK_EXPORT int tdk_encode12_u16(const uint16_t* in, uint8_t* out, int64_t num_pairs, int ids_format, tdk_stream_t stream) {
  return run_encode<uint16_t>(in, out, num_pairs, ids_format != 0, 1.0f, tdk_stream(stream), "tdk_encode12_u16");
}
TDK_EXPORT int tdk_encode12_f32(const float* in, uint8_t* out, int64_t num_pairs, int ids_format, int scaled, tdk_stream_t stream) {
  return run_encode<float>(in, out, num_pairs, ids_format != 0, scaled ? 4095.0f : 1.0f, tdk_stream(stream), "tdk_encode12_f32");
}
TDK_EXPORT int tdk_decode12_f32(const uint8_t* in, float* out, int64_t num_pairs, int ids_format, int scaled, tdk_stream_t stream) {
  return run_decode<float>(in, out, num_pairs, ids_format != 0, scaled ? (1.0f / 4095.0f) : 1.0f, tdk_stream(stream), "tdk_decode12_f32");
}
TDK_EXPORT int tdk_decode12_f16(const uint8_t* in, void* out_half, int64_t num_pairs, int ids_format, int scaled, tdk_stream_t stream) {
  return run_decode<__half>(in, reinterpret_cast<__half*>(out_half), num_pairs, ids_format != 0, scaled ? (1.0f / 4095.0f) : 1.0f,
                            tdk_stream(stream), "tdk_decode12_f16");
}
TDK_EXPORT int tdk_decode12_u16(const uint8_t* in, uint16_t* out, int64_t num_pairs, int ids_format, tdk_stream_t stream) {
  return run_decode<uint16_t>(in, out, num_pairs, ids_format != 0, 1.0f, tdk_stream(stream), "tdk_decode12_u16");
}
