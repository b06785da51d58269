// tdk_common.h -- shared host/device helpers for the gfx950 RAW-ISP kernels.
//
// Numerics contract: every translation unit is compiled with -ffp-contract=off and the
// default correctly-rounded fp32 divide/sqrt, so kernels built only from + - * / min max
// abs compare reproduce the strict-IEEE CPU oracle bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/tdk_hip.h"

#define TDK_EXPORT extern "C" __attribute__((visibility("default")))

// Experiment scaffolding (per-phase clocks, truncated kernels, getenv launch knobs: profiles/*_exp.py) is compiled only
// into -DTDK_EXPERIMENTS builds; the default library has none of it, so a debug path can never be the one that is timed.
#ifndef TDK_EXPERIMENTS
#undef TDK_RCD_TIMING
#undef TDK_RCD_STOP
#undef TDK_BIL_TIMING
#endif

// Wave priority of the short streaming kernels (luminance extraction, Wiener finish, metrics, tone map).  With several frames in
// flight their waves are always the YOUNGEST on a SIMD, and the issue arbiter serves the oldest wave first: next to the long-lived
// waves of the RCD / Wiener / bilateral tile kernels a tone-map launch takes 3.7 x its stand-alone time
// (profiles/r05/bench_isp_plain.json: kernel_ms_per_frame_streams).  s_setprio raises the wave's own priority (0 .. 3).
#ifndef TDK_STREAMING_PRIO
#define TDK_STREAMING_PRIO 0
#endif
// Rotating wave priority inside the long-running tile kernels.  All their workgroups are resident at once (2 or 3 per CU), the
// issue arbiter serves the OLDEST wave of a SIMD first, and so the first workgroup of every CU runs at full speed while the later
// ones take what is left and then finish alone, one wave per SIMD, at half the issue rate (profiles/r05/experiments/
// wg_lifetimes.txt: Wiener strips end at 115 / 165 us, RCD strips at 73 / 97 / 115 us).  tdk_rotate_prio(step) makes wave slot s
// the favoured one in every N-th step, so that the waves of a SIMD advance together and keep each other's stalls covered to the end.
#ifndef TDK_FAIR_PRIO
#define TDK_FAIR_PRIO 0
#endif
__device__ __forceinline__ int tdk_wave_slot() {  // HW_REG_HW_ID (4), WAVE_ID = bits 3:0: the wave's slot on its SIMD
  return (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
}
template <int N> __device__ __forceinline__ void tdk_rotate_prio(int step, int slot) {
  if (TDK_FAIR_PRIO) {
    if ((step + slot) % N == 0) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
}

#define TDK_STREAMING_KERNEL_PROLOGUE() do { if (TDK_STREAMING_PRIO > 0) __builtin_amdgcn_s_setprio(TDK_STREAMING_PRIO); } while (0)

// ---------------------------------------------------------------- host side: status + launch checks
void tdk_set_error(const char* fmt, ...);
int tdk_device_cus();  // compute units of the current device (cached)

#define TDK_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      tdk_set_error(__VA_ARGS__);              \
      return TDK_ERR_INVALID_ARGUMENT;         \
    }                                          \
  } while (0)

#define TDK_CHECK_LAUNCH(what)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      tdk_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));  \
      return TDK_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

#define TDK_HIP_CALL(expr, what)                                            \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess) {                                                 \
      tdk_set_error("%s: %s", what, hipGetErrorString(e_));                 \
      return TDK_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

// Every kernel launch goes through TDK_LAUNCH: optional per-kernel event timing (the
// tdk_profile_* entry points; the reference's counterpart is its CudaTimer,
// csrc/cuda_utils.h:40-85) plus the launch-error check.
bool tdk_timer_begin(const char* name, hipStream_t s);  // false: filtered out, not recorded
void tdk_timer_end(hipStream_t s);
extern bool g_tdk_profile_on;

#define TDK_LAUNCH(name, kernel, grid, block, lds, stream, ...)                       \
  do {                                                                                \
    const bool tdk_timed_ = g_tdk_profile_on && tdk_timer_begin(name, stream);        \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                \
    if (tdk_timed_) tdk_timer_end(stream);                                            \
    TDK_CHECK_LAUNCH(name);                                                           \
  } while (0)

// Raise a kernel's dynamic-LDS limit (a property of the function on a device): done once per (device, function), under a
// lock (runtime.hip), so hipFuncSetAttribute stays off the per-call path and a second GPU or thread is still served.
int tdk_raise_lds_limit(const void* func, int bytes, const char* what);
#define TDK_MAX_LDS_ONCE(kernel, what)                                                                   \
  do {                                                                                                    \
    const int tdk_rc_ = tdk_raise_lds_limit(reinterpret_cast<const void*>(&kernel), 160 * 1024, what);    \
    if (tdk_rc_ != TDK_OK) return tdk_rc_;                                                                \
  } while (0)

static inline int tdk_div_up(int a, int b) { return (a + b - 1) / b; }
static inline int64_t tdk_div_up64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t tdk_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline bool tdk_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline hipStream_t tdk_stream(tdk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------- device side
// CFA colour of (row, col): 0 = R, 1 = G, 2 = B (reference csrc/debayer/bayer_device.h:9-11)
__host__ __device__ __forceinline__ int cfa_color(int row, int col, uint32_t pattern) {
  return (int)((pattern >> ((((row << 1) & 14) + (col & 1)) << 1)) & 3u);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float sqf(float v) { return v * v; }
// reference csrc/device_math.h:80-82 and :407-409
__device__ __forceinline__ float mixf(float a, float b, float t) { return (1.0f - t) * a + t * b; }
__device__ __forceinline__ float lerpf(float t, float a, float b) { return a + t * (b - a); }

// Storage-type load/store: images live in HBM as float or __half, arithmetic is fp32.
template <typename T> __device__ __forceinline__ float ld(const T* p, size_t i);
template <> __device__ __forceinline__ float ld<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p, size_t i) { return __half2float(p[i]); }
template <typename T> __device__ __forceinline__ void st(T* p, size_t i, float v);
template <> __device__ __forceinline__ void st<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<__half>(__half* p, size_t i, float v) { p[i] = __float2half_rn(v); }

// the value a later kernel will read back after `v` has been stored as T
template <typename T> __device__ __forceinline__ float as_stored(float v);
template <> __device__ __forceinline__ float as_stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float as_stored<__half>(float v) { return __half2float(__float2half_rn(v)); }

struct f3 {
  float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }

// Four consecutive interleaved RGB pixels (12 values): 48 B as 3 x 16-B accesses for fp32,
// 24 B as 3 x 8-B accesses for fp16.  `i4` indexes groups of 4 pixels; the caller guarantees
// the base pointer is 16-B (fp32) / 8-B (fp16) aligned.
template <typename T> struct rgb4_io;
template <> struct rgb4_io<float> {
  static __device__ __forceinline__ void load(const float* p, size_t i4, float v[12]) {
    const float4* q = reinterpret_cast<const float4*>(p) + i4 * 3;
    float4 a = q[0], b = q[1], c = q[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
  }
  static __device__ __forceinline__ void store(float* p, size_t i4, const float v[12]) {
    float4* q = reinterpret_cast<float4*>(p) + i4 * 3;
    q[0] = make_float4(v[0], v[1], v[2], v[3]);
    q[1] = make_float4(v[4], v[5], v[6], v[7]);
    q[2] = make_float4(v[8], v[9], v[10], v[11]);
  }
};
template <> struct rgb4_io<__half> {
  static __device__ __forceinline__ void load(const __half* p, size_t i4, float v[12]) {
    const uint2* q = reinterpret_cast<const uint2*>(p) + i4 * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      uint2 u = q[k];
      __half2 lo = *reinterpret_cast<__half2*>(&u.x), hi = *reinterpret_cast<__half2*>(&u.y);
      float2 a = __half22float2(lo), b = __half22float2(hi);
      v[4 * k] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = b.x; v[4 * k + 3] = b.y;
    }
  }
  static __device__ __forceinline__ void store(__half* p, size_t i4, const float v[12]) {
    uint2* q = reinterpret_cast<uint2*>(p) + i4 * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      __half2 lo = __floats2half2_rn(v[4 * k], v[4 * k + 1]), hi = __floats2half2_rn(v[4 * k + 2], v[4 * k + 3]);
      uint2 u;
      u.x = *reinterpret_cast<unsigned int*>(&lo);
      u.y = *reinterpret_cast<unsigned int*>(&hi);
      q[k] = u;
    }
  }
};

// Four consecutive scalars of a plane.
template <typename T> struct s4_io;
template <> struct s4_io<float> {
  static __device__ __forceinline__ void load(const float* p, size_t i4, float v[4]) {
    float4 a = reinterpret_cast<const float4*>(p)[i4];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  static __device__ __forceinline__ void store(float* p, size_t i4, const float v[4]) {
    reinterpret_cast<float4*>(p)[i4] = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct s4_io<__half> {
  static __device__ __forceinline__ void load(const __half* p, size_t i4, float v[4]) {
    uint2 u = reinterpret_cast<const uint2*>(p)[i4];
    float2 a = __half22float2(*reinterpret_cast<__half2*>(&u.x)), b = __half22float2(*reinterpret_cast<__half2*>(&u.y));
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  }
  static __device__ __forceinline__ void store(__half* p, size_t i4, const float v[4]) {
    __half2 lo = __floats2half2_rn(v[0], v[1]), hi = __floats2half2_rn(v[2], v[3]);
    uint2 u;
    u.x = *reinterpret_cast<unsigned int*>(&lo);
    u.y = *reinterpret_cast<unsigned int*>(&hi);
    reinterpret_cast<uint2*>(p)[i4] = u;
  }
};

// wave64 reductions by cross-lane shuffles (all 64 lanes must be active)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Dispatch a templated launcher on the storage dtype tag.
#define TDK_DISPATCH_DTYPE(dtype, T, ...)                                  \
  do {                                                                     \
    if ((dtype) == TDK_F32) { using T = float; __VA_ARGS__; }              \
    else if ((dtype) == TDK_F16) { using T = __half; __VA_ARGS__; }        \
    else { tdk_set_error("unsupported dtype tag %d", (int)(dtype)); return TDK_ERR_INVALID_ARGUMENT; } \
  } while (0)
