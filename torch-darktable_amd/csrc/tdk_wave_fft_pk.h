// tdk_wave_fft_pk.h -- the in-register 32-point complex FFT of tdk_wave_fft.h on PACKED fp32 instructions.
//
// Why: on gfx950 a wave issues a plain VOP2 instruction every 4-8 cycles (tests/hip_unit/valu_issue_bench.hip: 1 wave per
// SIMD 4.0-8.1 cycles per instruction), so a kernel that register pressure holds at 2 waves per SIMD runs the SIMD at about
// half its rate.  A packed instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two fp32 operations per lane) issues
// every 4.3 cycles from ONE wave (tests/hip_unit/pk_issue_bench.hip) -- the SIMD's full fp32 rate -- and its op_sel / neg
// modifiers and an SGPR-pair operand are free.  A complex value kept as a {re, im} register pair makes every FFT butterfly
// packed: a + b and a - b are one instruction each, a * (-i) is an operand swizzle, and a general butterfly
// a +- w b with w = c (1 -+ i t) is three (tdk_wave_fft.h: six).  The compiler matches only some of the swizzles, so the
// primitives are inline asm; they are not volatile, the scheduler may reorder them.
//
// Constants travel as {c, t} pairs (c = cos or sin of the twiddle angle, t = tan or cot) in SGPR pairs.
#pragma once

#include "tdk_wave_fft.h"

namespace tdk_fft {

typedef float v2f __attribute__((ext_vector_type(2)));

// d = a + b, a - b
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { return a + b; }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) { return a - b; }
// d = a + (-i) b = {a.x + b.y, a.y - b.x};  d = a + i b = {a.x - b.y, a.y + b.x}
__device__ __forceinline__ v2f pk_add_mi(v2f a, v2f b) {
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ v2f pk_add_pi(v2f a, v2f b) {
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// b (1 - i t) = {b.x + t b.y, b.y - t b.x} (forward), b (1 + i t) (inverse); t = ct.y
template <bool INV> __device__ __forceinline__ v2f pk_tw_tan(v2f ct, v2f b) {
  v2f d;
  if constexpr (!INV) asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(d) : "s"(ct), "v"(b));
  else asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "s"(ct), "v"(b));
  return d;
}
// b (t - i) = {t b.x + b.y, t b.y - b.x} (forward), b (t + i) = {t b.x - b.y, t b.y + b.x} (inverse); t = ct.y
template <bool INV> __device__ __forceinline__ v2f pk_tw_cot(v2f ct, v2f b) {
  v2f d;
  if constexpr (!INV) asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[1,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]" : "=v"(d) : "s"(ct), "v"(b));
  else asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[1,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1]" : "=v"(d) : "s"(ct), "v"(b));
  return d;
}
// a + c p, a - c p; c = ct.x
__device__ __forceinline__ v2f pk_axpy_lo(v2f ct, v2f p, v2f a) {
  v2f d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(ct), "v"(p), "v"(a));
  return d;
}
__device__ __forceinline__ v2f pk_axmy_lo(v2f ct, v2f p, v2f a) {
  v2f d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "s"(ct), "v"(p), "v"(a));
  return d;
}

// Scalar broadcast of one half of a constant pair: x * w, fma(w, x, y) with w = wp.x (H = 0) or wp.y (H = 1)
template <int H> __device__ __forceinline__ v2f pk_scale(v2f wp, v2f x) {
  v2f d;
  if constexpr (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "s"(wp), "v"(x));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "s"(wp), "v"(x));
  return d;
}
template <int H> __device__ __forceinline__ v2f pk_fma_s(v2f wp, v2f x, v2f y) {
  v2f d;
  if constexpr (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(wp), "v"(x), "v"(y));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(wp), "v"(x), "v"(y));
  return d;
}

// The same radix-2 decimation-in-time network as fft_inreg (tdk_wave_fft.h), unscaled in both directions.
template <int N, bool INV> __device__ __forceinline__ void fft_inreg_pk(v2f (&z)[N]) {
  constexpr int STAGES = ilog2(N);
#pragma unroll
  for (int t = 0; t < N; t++) {
    const int r = bitrev(t, STAGES);
    if (t < r) {
      const v2f a = z[t];
      z[t] = z[r];
      z[r] = a;
    }
  }
#pragma unroll
  for (int s = 0; s < STAGES; s++) {
    const int step = 1 << s;
#pragma unroll
    for (int t = 0; t < N; t++) {
      if ((t & step) == 0) {
        const int p = t | step;
        const int k = (t & (step - 1)) * ((N / 2) >> s) * (32 / N);  // index into the 32-point table
        const v2f a = z[t], b = z[p];
        if (k == 0) {            // w = 1
          z[t] = pk_add(a, b);
          z[p] = pk_sub(a, b);
        } else if (k == 8) {     // w = -i (forward) / +i (inverse)
          z[t] = INV ? pk_add_pi(a, b) : pk_add_mi(a, b);
          z[p] = INV ? pk_add_mi(a, b) : pk_add_pi(a, b);
        } else {
          const double cd = TW_COS_D[k], sd = TW_SIN_D[k];
          if ((cd < 0 ? -cd : cd) >= sd) {   // w = c (1 -+ i tan)
            const v2f ct = {(float)cd, (float)(sd / cd)};
            const v2f q = pk_tw_tan<INV>(ct, b);
            z[t] = pk_axpy_lo(ct, q, a);
            z[p] = pk_axmy_lo(ct, q, a);
          } else {                           // w = s (cot -+ i)
            const v2f ct = {(float)sd, (float)(cd / sd)};
            const v2f q = pk_tw_cot<INV>(ct, b);
            z[t] = pk_axpy_lo(ct, q, a);
            z[p] = pk_axmy_lo(ct, q, a);
          }
        }
      }
    }
  }
}

}  // namespace tdk_fft
