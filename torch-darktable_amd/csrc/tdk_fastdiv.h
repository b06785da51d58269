// tdk_fastdiv.h -- the correctly rounded fp32 quotient without the range-scaling wrapper.
//
// hipcc expands an IEEE `a / b` into  v_div_scale x2, v_rcp, 4 x v_fma, v_mul, v_div_fmas, v_div_fixup.
// Measured on gfx950 (tests/hip_unit/div_bench.hip, profiles/r02/microbench.txt) that costs 50-60 SIMD
// cycles per wave64 division: v_div_fmas reads VCC (~23 cycles, like every VCC-reading VOP2/VOP3) and the
// two v_div_scale write an SGPR pair.  The arithmetic CORE of that expansion is
//     y = rcp(b); e = fma(-b, y, 1); y = fma(e, y, y);
//     q = a * y;  r = fma(-b, q, a); q = fma(r, y, q); r = fma(-b, q, a); q = fma(r, y, q)
// and the wrapper only (1) pre-scales operands by 2^+-64 and (2) patches 0 / inf / nan results.  Where the
// wrapper is the identity, running the core alone gives the SAME BITS as `a / b` -- the same instructions on
// the same values.  From the ISA definition of V_DIV_SCALE_F32 / V_DIV_FIXUP_F32 the wrapper is the identity
// when all of these hold (fp32 denormals are enabled in these kernels):
//     b normal, 1/b normal          (2^-126 <= |b| < 2^126)
//     a == +0, or |a| >= 2^-102     (biased exponent(a) > 23: the residuals a - b*q stay exactly representable)
//     a / b normal, exponent(a) - exponent(b) < 96
// (a == +0, b > 0: the core gives q = +0 = the fixed-up result; a == -0 is NOT covered: the core returns +0.)
// The callers establish these preconditions -- see rcd.hip (tile-level range check of the CFA samples for the
// non-negative numerators, a per-wave check of the signed ones).  tests/hip_unit/fastdiv_test.hip compares the
// core with `/` on the device over the guarded range (random sweeps + all-ones / power-of-two mantissas).
#pragma once

#include <hip/hip_runtime.h>

namespace tdk {

__device__ __forceinline__ float div_core(float a, float b) {
  float y = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y, 1.0f);
  y = __builtin_fmaf(e, y, y);
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}

// smallest numerator magnitude the callers let through to div_core (a margin above 2^-102: with
// |b| <= 2^44 the quotient stays normal as well)
constexpr float DIV_CORE_MIN_NUM = 0x1p-80f;

}  // namespace tdk
