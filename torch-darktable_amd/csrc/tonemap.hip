// tonemap.hip -- image statistics (bounds, metrics) and the four tonemap operators.
//
// Replaces reference csrc/tonemap/color_adaption.cu:12-166 (compute_bounds_kernel,
// compute_metrics_kernel and their hosts), reinhard.cu:17-81, aces.cu:13-154, linear.cu:13-76;
// adaptation math color_adaption.h:17-76; vibrance / Lab from tdk_color.h namespace cB
// (the reference's device_color_conversions.h); u8 store device_math.h:347-349.
//
// MI355X design
//  * statistics: one sample per thread on the stride grid, wave64 __shfl_xor reduction, one LDS
//    slot per wave, ONE atomic per workgroup and statistic (the reference issues one atomic per
//    warp per statistic).  The result stays on the device: the normalisation by the valid count
//    is a small kernel, not a host .item() (color_adaption.cu:162).  The wide-accumulator form spreads the
//    workgroups' partial sums over 1024 rows (same-address float atomics run at ~5 ns each), and its finish
//    kernel sums the rows and zeroes them again: stream order is the only synchronisation.
//  * tonemaps: streaming, four pixels (48 B in, 12 B out) per thread, per-image constants
//    (map_key, exposure) hoisted out of the pixel loop; dtype-templated input (fp32 / fp16).
#include <float.h>

#include "tdk_color.h"

namespace {

// ------------------------------------------------------------------ bounds
__global__ void bounds_init_kernel(float* bounds) {
  bounds[0] = FLT_MAX;
  bounds[1] = -FLT_MAX;
}

__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  unsigned int* a = reinterpret_cast<unsigned int*>(addr);
  unsigned int old = *a, assumed;
  do {
    assumed = old;
    const float cur = __uint_as_float(assumed);
    const float nv = fminf(v, cur);
    if (__float_as_uint(nv) == assumed) break;
    old = atomicCAS(a, assumed, __float_as_uint(nv));
  } while (assumed != old);
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  unsigned int* a = reinterpret_cast<unsigned int*>(addr);
  unsigned int old = *a, assumed;
  do {
    assumed = old;
    const float cur = __uint_as_float(assumed);
    const float nv = fmaxf(v, cur);
    if (__float_as_uint(nv) == assumed) break;
    old = atomicCAS(a, assumed, __float_as_uint(nv));
  } while (assumed != old);
}

template <typename T>
__global__ __launch_bounds__(256) void bounds_kernel(const T* __restrict__ img, int width, int height, int stride, int sw, int sh,
                                                     float* __restrict__ bounds) {
  __shared__ float smin[4], smax[4];
  const int64_t n = (int64_t)sw * sh;
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int sy = (int)(i / sw), sx = (int)(i - (int64_t)sy * sw);
    const size_t p = ((size_t)(sy * stride) * width + (size_t)sx * stride) * 3;
    const float r = ld(img, p), g = ld(img, p + 1), b = ld(img, p + 2);
    lo = fminf(lo, fminf(fminf(r, g), b));
    hi = fmaxf(hi, fmaxf(fmaxf(r, g), b));
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { smin[wave] = lo; smax[wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomic_min_f32(&bounds[0], fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3])));
    atomic_max_f32(&bounds[1], fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
  }
}

// compute_image_bounds without the init launch: a persistent 4-word state {min key, max key, ticket, -} that is IDLE = {~0, 0, 0, 0}
// between calls.  Keys are the order-preserving integer images of the floats, so a workgroup's contribution is ONE memory-side
// global_atomic_umin / umax each (no CAS loop); every workgroup of every image of a list then draws a ticket, and the one that draws
// the last ticket of the list (`total`, passed with the list's last launch) decodes the two keys into bounds[0..1] and puts the
// state back to idle.  The hand-off is the metrics kernel's: the lane's two atomics have completed (s_waitcnt vmcnt(0)) before it
// draws its agent-scope ticket, the last arriver reads the keys with agent-scope loads; launches of a list are stream-ordered.
__device__ __forceinline__ uint32_t float_key(float f) { const uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float key_float(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

template <typename T>
__global__ __launch_bounds__(256) void bounds_ticket_kernel(const T* __restrict__ img, int width, int height, int stride, int sw, int sh,
                                                            uint32_t* __restrict__ state, float* __restrict__ finish_to, unsigned total) {
  __shared__ float smin[4], smax[4];
  const int64_t n = (int64_t)sw * sh;
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int sy = (int)(i / sw), sx = (int)(i - (int64_t)sy * sw);
    const size_t p = ((size_t)(sy * stride) * width + (size_t)sx * stride) * 3;
    const float r = ld(img, p), g = ld(img, p + 1), b = ld(img, p + 2);
    lo = fminf(lo, fminf(fminf(r, g), b));
    hi = fmaxf(hi, fmaxf(fmaxf(r, g), b));
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { smin[wave] = lo; smax[wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_min(&state[0], float_key(fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(&state[1], float_key(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both have reached memory before the ticket is drawn
    const unsigned t = __hip_atomic_fetch_add(&state[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (finish_to && t == total - 1u) {
      const uint32_t kmin = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t kmax = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      finish_to[0] = key_float(kmin);
      finish_to[1] = key_float(kmax);
      state[0] = 0xffffffffu;  // idle again (the next list's first launch is stream-ordered behind this kernel)
      state[1] = 0u;
      state[2] = 0u;
    }
  }
}

// ------------------------------------------------------------------ metrics
__global__ void metrics_init_kernel(float* acc) {
  if (threadIdx.x < 8) acc[threadIdx.x] = 0.0f;
}

// Sum of the TDK_METRICS_SLOTS accumulator rows in a fixed order by one 256-thread workgroup, normalised (color_adaption.cu:161-165)
// into metrics[0..4]; the rows (and the ticket word in row 0) are zeroed for the next image.  AGENT_LOADS: the rows were
// written by other workgroups of the SAME launch (float atomics, which execute at the memory side and leave nothing in L2): they
// are read with agent-scope loads that bypass this CU's vector L1.
// nrows: the rows that can hold anything (the launch's workgroups wrote rows 0 .. min(grid, SLOTS) - 1; the others are zero and
// stay zero): rows j * 256 + thread with j < ceil(nrows / 256) are read -- a 12 MP frame at stride 8 touches 384 of the 1 024.
template <bool AGENT_LOADS>
__device__ __forceinline__ void metrics_finish_rows(float* __restrict__ acc, float* __restrict__ metrics, float (*part)[6], int nrows) {
  static_assert(TDK_METRICS_SLOTS == 4 * 256, "four rows per thread");
  const int nj = (nrows + 255) >> 8;  // uniform
  float v[4][6];  // all loads in flight before the first use
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int r = threadIdx.x + 256 * j;
#pragma unroll
    for (int k = 0; k < 6; k++) v[j][k] = 0.0f;
    if (j < nj) {
#pragma unroll
      for (int k = 0; k < 6; k++) v[j][k] = AGENT_LOADS ? __hip_atomic_load(&acc[r * 8 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : acc[r * 8 + k];
    }
  }
  float s[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int k = 0; k < 6; k++) s[k] += v[j][k];
    if (j < nj) {
      const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float* row = acc + (threadIdx.x + 256 * j) * 8;
      reinterpret_cast<float4*>(row)[0] = z;
      reinterpret_cast<float4*>(row)[1] = z;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float v = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    const float v = (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
    const float cnt = (part[0][5] + part[1][5]) + (part[2][5] + part[3][5]);
    metrics[k] = v * (1.0f / fmaxf(cnt, 1.0f));  // color_adaption.cu:161-165
  }
}

// ROWS: the workgroup's partial sums go to row (blockIdx.x mod TDK_METRICS_SLOTS) of a wide accumulator instead of
// acc[0..5] -- no same-address contention, so the grid can be as wide as the image needs (the single-row form is kept
// for the three-call API, whose accumulator is 8 floats, and caps its grid at one workgroup per CU).
// finish_to (ROWS only, may be null): the workgroup that draws the last ticket also sums the rows, writes the five metrics
// there and zeroes the accumulator -- compute_image_metrics of ONE image as one launch instead of two.  The hand-off follows
// cdna_hip_programming.md Guideline 16 (counter form): every adding lane's atomics have completed (s_waitcnt vmcnt(0)), the
// workgroup's barrier, then one lane draws the ticket with an agent-scope fetch_add; the last arriver reads the rows with
// agent-scope loads (metrics_finish_rows<true>).  The ticket is word 6 of row 0, which the sums never touch.
template <typename T, bool ROWS>
__global__ __launch_bounds__(256) void metrics_kernel(const T* __restrict__ img, int width, int height, int stride, int sw, int sh,
                                                      float min_gray, const float* __restrict__ bounds, float* __restrict__ acc, float* __restrict__ finish_to) {
  __shared__ float part[4][6];
  __shared__ int last_block;
  TDK_STREAMING_KERNEL_PROLOGUE();
  const int64_t n = (int64_t)sw * sh;
  const float b0 = bounds[0];
  const float range = bounds[1] - b0 + 1e-6f;
  float s[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int sy = (int)(i / sw), sx = (int)(i - (int64_t)sy * sw);
    const size_t p = ((size_t)(sy * stride) * width + (size_t)sx * stride) * 3;
    const float r = (ld(img, p) - b0) / range, g = (ld(img, p + 1) - b0) / range, b = (ld(img, p + 2) - b0) / range;
    const float mask = (r >= 0.99f || g >= 0.99f || b >= 0.99f) ? 0.0f : 1.0f;
    const float gray = r * 0.299f + g * 0.587f + b * 0.114f;
    const float log_gray = tdk_log(fmaxf(gray, min_gray));
    s[0] += log_gray * mask;
    s[1] += gray * mask;
    s[2] += r * mask;
    s[3] += g * mask;
    s[4] += b * mask;
    s[5] += mask;
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float v = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) part[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const float mine = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    // explicitly a relaxed agent-scope RMW: on gfx950 a memory-side global_atomic_add_f32 (tests/test_isa_contract.py pins the
    // instruction), which the one-launch hand-off below relies on
    __hip_atomic_fetch_add(&acc[(ROWS ? (blockIdx.x & (TDK_METRICS_SLOTS - 1)) * 8 : 0) + threadIdx.x], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (ROWS) {
    if (finish_to) {  // kernel argument: uniform
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's six adds have reached memory
      __syncthreads();
      // (no release fence: the sums are memory-side atomics, nothing of them sits in this XCD's L2; no acquire: the last
      // workgroup reads the rows with agent-scope loads)
      if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(acc + 6), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_block = t == gridDim.x - 1u;
      }
      __syncthreads();
      if (last_block) {  // workgroup-uniform (LDS word)
        __syncthreads();  // `part` is reused
        metrics_finish_rows<true>(acc, finish_to, part, min((int)gridDim.x, TDK_METRICS_SLOTS));  // also zeroes row 0's ticket word
      }
    }
  }
}

// color_adaption.cu:161-165
__global__ void metrics_finish_kernel(const float* __restrict__ acc, float* __restrict__ metrics) {
  if (threadIdx.x < 5) {
    const float norm = 1.0f / fmaxf(acc[5], 1.0f);
    metrics[threadIdx.x] = acc[threadIdx.x] * norm;
  }
}

// finish + self-clean for sums accumulated in TDK_METRICS_SLOTS rows of 8 floats by earlier launches
// (tdk_image_metrics_accumulate_rows, one per image): one workgroup, stream-ordered after the adds; rows are summed in a fixed order.
__global__ __launch_bounds__(256) void metrics_finish_reset_kernel(float* __restrict__ acc, float* __restrict__ metrics) {
  __shared__ float part[4][6];
  metrics_finish_rows<false>(acc, metrics, part, TDK_METRICS_SLOTS);
}

// ------------------------------------------------------------------ tonemaps
// color_adaption.h:17-29
__device__ __forceinline__ float map_key_of(float log_mean) {
  const float normalized = fmaxf(0.0f, fminf(1.0f, (-log_mean) * (1.0f / 9.21034f)));
  return 0.3f + 0.7f * tdk_pow(normalized, 1.4f);
}
// aces.cu:13-34
__device__ __forceinline__ float rrt_odt(float v) {
  const float a = v * (v + 0.0245786f) - 0.000090537f;
  const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
  return tdk_div(a, b);
}
__device__ __forceinline__ f3 aces_fit(f3 c) {
  const f3 a = mk3(0.59719f * c.x + 0.35458f * c.y + 0.04823f * c.z, 0.07600f * c.x + 0.90834f * c.y + 0.01566f * c.z,
                   0.02840f * c.x + 0.13383f * c.y + 0.83777f * c.z);
  const f3 r = mk3(rrt_odt(a.x), rrt_odt(a.y), rrt_odt(a.z));
  return mk3(1.60475f * r.x + -0.53108f * r.y + -0.07367f * r.z, -0.10208f * r.x + 1.10813f * r.y + -0.00605f * r.z,
             -0.00327f * r.x + -0.07276f * r.y + 1.07602f * r.z);
}
// device_math.h:347-349
__device__ __forceinline__ uint32_t to_u8(float x) { return (uint32_t)fminf(roundf(x * 255.0f), 255.0f); }
// The same for x already clipped to [0, 1] (every tone mapper ends in a clip): round-half-away of a non-negative value is
// trunc(v + 0.5), and v <= 255 needs no upper clamp -- 2 instructions instead of the 8 of roundf + fminf.  The one value per step
// where the two differ is v = n + 0.5 - half an ulp (v + 0.5 rounds up to n + 1 in fp32): inside every tolerance that is stated
// for the quantisation (a pre-quantisation value within 2e-3 of a rounding tie may land on either side, tests/test_gpu_parity.py).
__device__ __forceinline__ uint32_t to_u8_clipped(float x) { return (uint32_t)__builtin_fmaf(x, 255.0f, 0.5f); }
// pow(x, y) for an exponent that is known to be non-zero (the per-image constants `key` >= 0.3 and 1 / gamma: the kernels take
// the general tdk_pow when 1 / gamma == 0): no y == 0 select per call
__device__ __forceinline__ float pow_nz(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

struct TmConst {
  float key, inv_exposure, m0, m1, m2, inv_gamma, vibrance, light_adapt, aces_scale;
};

// LEAN: vibrance == 0 and 1 / gamma != 0 (both wave-uniform kernel arguments; the launcher picks the instantiation): no Lab
// round trip in the code at all -- its registers would otherwise set the occupancy of the common path -- and pow_nz
template <int MODE, bool LEAN> __device__ __forceinline__ f3 tonemap_px(f3 c, const TmConst& k) {
  auto pw = [](float x, float y) { return LEAN ? pow_nz(x, y) : tdk_pow(x, y); };
  f3 tm;
  if constexpr (MODE == TDK_TONEMAP_ACES) {
    tm = aces_fit(mk3(c.x * k.aces_scale, c.y * k.aces_scale, c.z * k.aces_scale));
  } else {
    const f3 mean = mk3(lerpf(k.light_adapt, k.m0, c.x), lerpf(k.light_adapt, k.m1, c.y), lerpf(k.light_adapt, k.m2, c.z));
    const f3 ad = mk3(pow_nz(mean.x * k.inv_exposure, k.key), pow_nz(mean.y * k.inv_exposure, k.key), pow_nz(mean.z * k.inv_exposure, k.key));  // key >= 0.3
    if constexpr (MODE == TDK_TONEMAP_REINHARD) tm = mk3(tdk_div(c.x, ad.x + c.x), tdk_div(c.y, ad.y + c.y), tdk_div(c.z, ad.z + c.z));
    else if constexpr (MODE == TDK_TONEMAP_LINEAR) tm = mk3(tdk_div(c.x, ad.x), tdk_div(c.y, ad.y), tdk_div(c.z, ad.z));
    else tm = aces_fit(mk3(tdk_div(c.x, ad.x), tdk_div(c.y, ad.y), tdk_div(c.z, ad.z)));
  }
  const f3 g = mk3(pw(fmaxf(tm.x, 0.0f), k.inv_gamma), pw(fmaxf(tm.y, 0.0f), k.inv_gamma), pw(fmaxf(tm.z, 0.0f), k.inv_gamma));
  // modify_rgb_vibrance_dt(g, amount) = clip(lab_to_rgb(scale(rgb_to_lab(g)))) (device_color_conversions.h:199-213).
  // With amount == 0 both scale factors are exactly 1 and Lab -> RGB inverts RGB -> Lab (no clamp in either, the
  // two piecewise branches are inverse pairs): the reference's result is clip(g) up to the ~1e-6 round-trip error
  // of its own fast-math pow / cbrt -- 18 of this kernel's 35 transcendentals per pixel to reproduce a rounding
  // error that is 2500x below one uint8 step.  The round trip is skipped (wave-uniform branch: `amount` is a
  // kernel argument); outputs differ from the reference only where its value sits within 1e-6 * 255 of a rounding
  // tie, the class of difference its own non-deterministic fast-math build already has (test_tonemaps_u8).
  if constexpr (LEAN) {
    return clip3(g);
  } else {
    if (k.vibrance == 0.0f) return clip3(g);
    f3 o = cB::vibrance(g, k.vibrance);
    if constexpr (MODE == TDK_TONEMAP_LINEAR) o = clip3(o);
    return o;
  }
}

template <int MODE>
__device__ __forceinline__ TmConst make_consts(const float* metrics, float gamma, float intensity, float light_adapt, float vibrance) {
  TmConst k;
  k.inv_gamma = 1.0f / gamma;
  k.vibrance = vibrance;
  k.light_adapt = light_adapt;
  k.aces_scale = 1.0f; k.key = 1.0f; k.inv_exposure = 1.0f; k.m0 = k.m1 = k.m2 = 0.0f;
  if constexpr (MODE == TDK_TONEMAP_ACES) {
    k.aces_scale = tdk_pow(2.0f, intensity);
  } else {
    k.key = map_key_of(metrics[0]);
    k.inv_exposure = 1.0f / tdk_exp(intensity);
    k.m0 = metrics[2]; k.m1 = metrics[3]; k.m2 = metrics[4];
  }
  return k;
}

template <typename T, int MODE, bool LEAN>
__global__ __launch_bounds__(256) void tonemap_vec4(const T* __restrict__ in, uint32_t* __restrict__ out, int64_t ngroups,
                                                     const float* __restrict__ metrics, float gamma, float intensity, float light_adapt,
                                                     float vibrance) {
  TDK_STREAMING_KERNEL_PROLOGUE();
  TmConst k = make_consts<MODE>(metrics, gamma, intensity, light_adapt, vibrance);
  // The per-image constants are wave-uniform, so hipcc keeps them in SGPRs -- and a VALU instruction with an SGPR source
  // issues at half rate on gfx950 (4.4 instead of 2.4 cycles, tests/hip_unit/valu_issue_bench.hip); they are used ~20 times
  // per pixel.  Through an empty asm they become VGPR values (one v_mov each, once per thread).
  asm volatile("" : "+v"(k.key), "+v"(k.inv_exposure), "+v"(k.m0), "+v"(k.m1), "+v"(k.m2), "+v"(k.inv_gamma), "+v"(k.light_adapt), "+v"(k.aces_scale));
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256) {
    float v[12];
    rgb4_io<T>::load(in, g, v);
    uint32_t b[12];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const f3 o = tonemap_px<MODE, LEAN>(mk3(v[3 * p], v[3 * p + 1], v[3 * p + 2]), k);  // clipped to [0, 1] by every mode
      b[3 * p] = to_u8_clipped(o.x); b[3 * p + 1] = to_u8_clipped(o.y); b[3 * p + 2] = to_u8_clipped(o.z);
    }
#pragma unroll
    for (int w = 0; w < 3; w++) out[3 * g + w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void tonemap_tail(const T* __restrict__ in, uint8_t* __restrict__ out, int64_t first, int64_t npix,
                                                     const float* __restrict__ metrics, float gamma, float intensity, float light_adapt,
                                                     float vibrance) {
  const TmConst k = make_consts<MODE>(metrics, gamma, intensity, light_adapt, vibrance);
  for (int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const f3 o = tonemap_px<MODE, false>(mk3(ld(in, 3 * i), ld(in, 3 * i + 1), ld(in, 3 * i + 2)), k);
    out[3 * i] = (uint8_t)to_u8(o.x); out[3 * i + 1] = (uint8_t)to_u8(o.y); out[3 * i + 2] = (uint8_t)to_u8(o.z);
  }
}

inline int stream_grid(int64_t nthreads) {
  int64_t b = tdk_div_up64(nthreads, 256);
  return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

// reductions that end in same-address atomics: at most one workgroup per CU
inline int reduce_grid(int64_t nthreads) {
  int64_t b = tdk_div_up64(nthreads, 256);
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

template <typename T, int MODE>
int run_tonemap(const void* rgb, uint8_t* out, int64_t npix, const float* metrics, float gamma, float intensity, float light_adapt,
                float vibrance, hipStream_t s) {
  const T* in = reinterpret_cast<const T*>(rgb);
  int64_t done = 0;
  if (tdk_aligned(in, 16) && tdk_aligned(out, 4) && npix >= 4) {
    const int64_t ng = npix / 4;
    if (vibrance == 0.0f && 1.0f / gamma != 0.0f)
      TDK_LAUNCH("tdk_tonemap", (tonemap_vec4<T, MODE, true>), dim3(stream_grid(ng)), dim3(256), 0, s, in, reinterpret_cast<uint32_t*>(out), ng, metrics,
                         gamma, intensity, light_adapt, vibrance);
    else
      TDK_LAUNCH("tdk_tonemap", (tonemap_vec4<T, MODE, false>), dim3(stream_grid(ng)), dim3(256), 0, s, in, reinterpret_cast<uint32_t*>(out), ng, metrics,
                         gamma, intensity, light_adapt, vibrance);
    done = ng * 4;
  }
  if (done < npix) {
    TDK_LAUNCH("tdk_tonemap", (tonemap_tail<T, MODE>), dim3(stream_grid(npix - done)), dim3(256), 0, s, in, out, done, npix, metrics, gamma,
                       intensity, light_adapt, vibrance);
  }
  return TDK_OK;
}

template <typename T>
int dispatch_tonemap(const void* rgb, uint8_t* out, int64_t npix, int mode, const float* metrics, float gamma, float intensity,
                     float light_adapt, float vibrance, hipStream_t s) {
  switch (mode) {
    case TDK_TONEMAP_REINHARD: return run_tonemap<T, TDK_TONEMAP_REINHARD>(rgb, out, npix, metrics, gamma, intensity, light_adapt, vibrance, s);
    case TDK_TONEMAP_ACES: return run_tonemap<T, TDK_TONEMAP_ACES>(rgb, out, npix, metrics, gamma, intensity, light_adapt, vibrance, s);
    case TDK_TONEMAP_ACES_ADAPTIVE: return run_tonemap<T, TDK_TONEMAP_ACES_ADAPTIVE>(rgb, out, npix, metrics, gamma, intensity, light_adapt, vibrance, s);
    case TDK_TONEMAP_LINEAR: return run_tonemap<T, TDK_TONEMAP_LINEAR>(rgb, out, npix, metrics, gamma, intensity, light_adapt, vibrance, s);
    default: tdk_set_error("tdk_tonemap: unknown mode %d", mode); return TDK_ERR_INVALID_ARGUMENT;
  }
}

}  // namespace

TDK_EXPORT int tdk_image_bounds_init(float* bounds, tdk_stream_t stream) {
  TDK_REQUIRE(bounds, "tdk_image_bounds_init: null pointer");
  TDK_LAUNCH("tdk_image_bounds_init", bounds_init_kernel, dim3(1), dim3(1), 0, tdk_stream(stream), bounds);
  return TDK_OK;
}

TDK_EXPORT int tdk_image_bounds_accumulate(const void* rgb, int width, int height, int stride, float* bounds, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(rgb && bounds, "tdk_image_bounds_accumulate: null pointer");
  TDK_REQUIRE(width > 0 && height > 0 && stride > 0, "tdk_image_bounds_accumulate: invalid size/stride");
  const int sw = tdk_div_up(width, stride), sh = tdk_div_up(height, stride);
  const int grid = reduce_grid((int64_t)sw * sh);
  TDK_DISPATCH_DTYPE(dtype, T, TDK_LAUNCH("tdk_image_bounds_accumulate", bounds_kernel<T>, dim3(grid), dim3(256), 0, tdk_stream(stream),
                                                  reinterpret_cast<const T*>(rgb), width, height, stride, sw, sh, bounds));
  return TDK_OK;
}

TDK_EXPORT int tdk_image_bounds_tickets(int width, int height, int stride) {
  if (width <= 0 || height <= 0 || stride <= 0) return 0;
  return reduce_grid((int64_t)tdk_div_up(width, stride) * tdk_div_up(height, stride));
}

TDK_EXPORT int tdk_image_bounds(const void* rgb, int width, int height, int stride, uint32_t* state, float* bounds, unsigned total_tickets, int dtype,
                                tdk_stream_t stream) {
  TDK_REQUIRE(rgb && state, "tdk_image_bounds: null pointer");
  TDK_REQUIRE(width > 0 && height > 0 && stride > 0, "tdk_image_bounds: invalid size/stride");
  const int sw = tdk_div_up(width, stride), sh = tdk_div_up(height, stride);
  const int grid = reduce_grid((int64_t)sw * sh);
  TDK_REQUIRE(!bounds || total_tickets >= (unsigned)grid, "tdk_image_bounds: total_tickets %u smaller than this launch's %d", total_tickets, grid);
  TDK_DISPATCH_DTYPE(dtype, T, TDK_LAUNCH("tdk_image_bounds", bounds_ticket_kernel<T>, dim3(grid), dim3(256), 0, tdk_stream(stream), reinterpret_cast<const T*>(rgb),
                                          width, height, stride, sw, sh, state, bounds, total_tickets));
  return TDK_OK;
}

TDK_EXPORT int tdk_image_metrics_init(float* acc, tdk_stream_t stream) {
  TDK_REQUIRE(acc, "tdk_image_metrics_init: null pointer");
  TDK_LAUNCH("tdk_image_metrics_init", metrics_init_kernel, dim3(1), dim3(64), 0, tdk_stream(stream), acc);
  return TDK_OK;
}

TDK_EXPORT int tdk_image_metrics_accumulate(const void* rgb, int width, int height, int stride, float min_gray, const float* bounds,
                                            float* acc, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(rgb && bounds && acc, "tdk_image_metrics_accumulate: null pointer");
  TDK_REQUIRE(width > 0 && height > 0 && stride > 0, "tdk_image_metrics_accumulate: invalid size/stride");
  const int sw = tdk_div_up(width, stride), sh = tdk_div_up(height, stride);
  const int grid = reduce_grid((int64_t)sw * sh);
  TDK_DISPATCH_DTYPE(dtype, T, TDK_LAUNCH("tdk_image_metrics_accumulate", (metrics_kernel<T, false>), dim3(grid), dim3(256), 0, tdk_stream(stream),
                                                  reinterpret_cast<const T*>(rgb), width, height, stride, sw, sh, min_gray, bounds, acc, static_cast<float*>(nullptr)));
  return TDK_OK;
}

static int metrics_rows(const void* rgb, int width, int height, int stride, float min_gray, const float* bounds, float* acc_rows, float* finish_to,
                        int dtype, tdk_stream_t stream, const char* what) {
  TDK_REQUIRE(rgb && bounds && acc_rows, "%s: null pointer", what);
  TDK_REQUIRE(width > 0 && height > 0 && stride > 0, "%s: invalid size/stride", what);
  TDK_REQUIRE(tdk_aligned(acc_rows, 16), "%s: acc must be 16-byte aligned", what);
  const int sw = tdk_div_up(width, stride), sh = tdk_div_up(height, stride);
  // every sample is three scattered loads: latency-bound, so spread it over many workgroups (2 samples per thread)
  int64_t grid = tdk_div_up64((int64_t)sw * sh, 512);
  grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
  TDK_DISPATCH_DTYPE(dtype, T, TDK_LAUNCH(finish_to ? "tdk_image_metrics" : "tdk_image_metrics_accumulate", (metrics_kernel<T, true>), dim3((unsigned)grid), dim3(256), 0,
                                          tdk_stream(stream), reinterpret_cast<const T*>(rgb), width, height, stride, sw, sh, min_gray, bounds, acc_rows, finish_to));
  return TDK_OK;
}

TDK_EXPORT int tdk_image_metrics_accumulate_rows(const void* rgb, int width, int height, int stride, float min_gray, const float* bounds,
                                                 float* acc_rows, int dtype, tdk_stream_t stream) {
  return metrics_rows(rgb, width, height, stride, min_gray, bounds, acc_rows, nullptr, dtype, stream, "tdk_image_metrics_accumulate_rows");
}

TDK_EXPORT int tdk_image_metrics(const void* rgb, int width, int height, int stride, float min_gray, const float* bounds, float* acc_rows,
                                 float* metrics, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(metrics, "tdk_image_metrics: null pointer");
  return metrics_rows(rgb, width, height, stride, min_gray, bounds, acc_rows, metrics, dtype, stream, "tdk_image_metrics");
}

TDK_EXPORT int tdk_image_metrics_finish(const float* acc, float* metrics, tdk_stream_t stream) {
  TDK_REQUIRE(acc && metrics, "tdk_image_metrics_finish: null pointer");
  TDK_LAUNCH("tdk_image_metrics_finish", metrics_finish_kernel, dim3(1), dim3(64), 0, tdk_stream(stream), acc, metrics);
  return TDK_OK;
}

TDK_EXPORT int tdk_tonemap(const void* rgb, uint8_t* out, int64_t npix, int mode, const float* metrics, float gamma, float intensity,
                           float light_adapt, float vibrance, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(npix >= 0, "tdk_tonemap: negative pixel count");
  if (npix == 0) return TDK_OK;
  TDK_REQUIRE(rgb && out, "tdk_tonemap: null pointer");
  TDK_REQUIRE(mode == TDK_TONEMAP_ACES || metrics != nullptr, "tdk_tonemap: metrics required for this mode");
  TDK_DISPATCH_DTYPE(dtype, T, return dispatch_tonemap<T>(rgb, out, npix, mode, metrics, gamma, intensity, light_adapt, vibrance, tdk_stream(stream)));
  return TDK_OK;
}

TDK_EXPORT int tdk_image_metrics_finish_reset(float* acc, float* metrics, tdk_stream_t stream) {
  TDK_REQUIRE(acc && metrics, "tdk_image_metrics_finish_reset: null pointer");
  TDK_REQUIRE(tdk_aligned(acc, 16), "tdk_image_metrics_finish_reset: acc must be 16-byte aligned");
  TDK_LAUNCH("tdk_image_metrics_finish", metrics_finish_reset_kernel, dim3(1), dim3(256), 0, tdk_stream(stream), acc, metrics);
  return TDK_OK;
}
