// dpp_transpose.h -- in-register K x K transposition with cross-lane VALU moves (experiment record).
//
// Built for the Wiener tile kernel to free its LDS transposition buffer; verified element by element
// (wave_fft_test.hip) but NOT used by the product: on gfx950 it is slower than the LDS round trip
// (wave_fft_bench.hip, dpp_bench*.hip).  Kept so the measurement can be repeated.
#pragma once

#include "tdk_wave_fft.h"

namespace tdk_fft {

// ---------------------------------------------------------------- in-register K x K transposition
// Data: M[lane][reg], lane = position inside the slot (K consecutive lanes), reg = 0..K-1.  Stage b swaps
// bit b of the lane index with bit b of the register index: for every register pair (A = v[r], B = v[r | m]),
// m = 1 << b, the lanes with bit b SET take B of lane ^ m into A, the lanes with bit b CLEAR take A of
// lane ^ m into B.  After all log2(K) stages M is transposed.
//   m = 1, 2 : quad_perm            m = 4 : row_shr:4 / row_shl:4          m = 8 : row_ror:8
//   -> two v_cndmask_b32_dpp per pair (the DPP operand is the one taken where VCC = 0, so VCC is the
//      "keep" mask: clear-bit lanes for A, set-bit lanes for B);
//   m = 16 : one v_permlane16_swap_b32 per pair (swaps the odd 16-lane rows of A with the even rows of B).
// Written as inline asm: hipcc turns `cond ? a : mov_dpp(b)` into EXEC-masked branches, and a DPP read of a
// lane that EXEC disables returns 0 -- and its __builtin_amdgcn_permlane16_swap loses the second result
// (ROCm 7.2).  hipcc inserts no wait states inside or around asm, so every block opens with the two wait
// states a DPP / permlane read needs after a VALU write of its source (s_nop 1).  EXEC must be all ones.
#define TDK_CND_DPP(d, s0, s1, ctrl) "v_cndmask_b32_dpp " d ", " s0 ", " s1 ", vcc " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"

// one stage on four register pairs: n_i = keep-clear ? a_i : dpp(b_i);  b_i = keep-set ? b_i : dpp(a_i)
#define TDK_XPOSE4(CTRL_A, CTRL_B)                                                                                                     \
  asm volatile("s_nop 1\n\t"                                                                                                           \
               "s_mov_b64 vcc, %[lo]\n\t" TDK_CND_DPP("%[n0]", "%[b0]", "%[a0]", CTRL_A) TDK_CND_DPP("%[n1]", "%[b1]", "%[a1]", CTRL_A)  \
                   TDK_CND_DPP("%[n2]", "%[b2]", "%[a2]", CTRL_A) TDK_CND_DPP("%[n3]", "%[b3]", "%[a3]", CTRL_A)                         \
               "s_mov_b64 vcc, %[hi]\n\t" TDK_CND_DPP("%[b0]", "%[a0]", "%[b0]", CTRL_B) TDK_CND_DPP("%[b1]", "%[a1]", "%[b1]", CTRL_B)  \
                   TDK_CND_DPP("%[b2]", "%[a2]", "%[b2]", CTRL_B) TDK_CND_DPP("%[b3]", "%[a3]", "%[b3]", CTRL_B)                         \
               : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3) \
               : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [lo] "s"(keep_clear), [hi] "s"(keep_set)                      \
               : "vcc")

template <int B> __device__ __forceinline__ void xpose4(float& a0, float& a1, float& a2, float& a3, float& b0, float& b1, float& b2, float& b3) {
  // lanes whose bit B is clear / set (the pattern repeats every 16 lanes, so it is slot-size independent)
  constexpr unsigned long long SET = B == 0 ? 0xAAAAAAAAAAAAAAAAull : B == 1 ? 0xCCCCCCCCCCCCCCCCull : B == 2 ? 0xF0F0F0F0F0F0F0F0ull : 0xFF00FF00FF00FF00ull;
  const unsigned long long keep_clear = ~SET, keep_set = SET;
  float n0, n1, n2, n3;
  if constexpr (B == 0) TDK_XPOSE4("quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]");
  else if constexpr (B == 1) TDK_XPOSE4("quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]");
  else if constexpr (B == 2) TDK_XPOSE4("row_shr:4", "row_shl:4");   // set-bit lanes read lane - 4, clear-bit lanes read lane + 4
  else TDK_XPOSE4("row_ror:8", "row_ror:8");
  a0 = n0; a1 = n1; a2 = n2; a3 = n3;
}

__device__ __forceinline__ void swap_rows16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// Transpose the K x K tile of every slot of the wave in place (v[k] of lane r  <->  v[r] of lane k).
template <int K> __device__ __forceinline__ void transpose_inreg(float (&v)[K]) {
  constexpr int BITS = ilog2(K);
#pragma unroll
  for (int b = 0; b < BITS && b < 4; b++) {
    const int m = 1 << b;
#pragma unroll
    for (int p = 0; p < K / 2; p += 4) {
      // the p-th register whose bit b is clear: ((p >> b) << (b + 1)) | (p & (m - 1))
      const int r0 = (((p + 0) >> b) << (b + 1)) | ((p + 0) & (m - 1)), r1 = (((p + 1) >> b) << (b + 1)) | ((p + 1) & (m - 1));
      const int r2 = (((p + 2) >> b) << (b + 1)) | ((p + 2) & (m - 1)), r3 = (((p + 3) >> b) << (b + 1)) | ((p + 3) & (m - 1));
      if (b == 0) xpose4<0>(v[r0], v[r1], v[r2], v[r3], v[r0 | m], v[r1 | m], v[r2 | m], v[r3 | m]);
      else if (b == 1) xpose4<1>(v[r0], v[r1], v[r2], v[r3], v[r0 | m], v[r1 | m], v[r2 | m], v[r3 | m]);
      else if (b == 2) xpose4<2>(v[r0], v[r1], v[r2], v[r3], v[r0 | m], v[r1 | m], v[r2 | m], v[r3 | m]);
      else xpose4<3>(v[r0], v[r1], v[r2], v[r3], v[r0 | m], v[r1 | m], v[r2 | m], v[r3 | m]);
    }
  }
  if constexpr (K == 32) {
#pragma unroll
    for (int r = 0; r < 16; r++) swap_rows16(v[r], v[r | 16]);
  }
}

}  // namespace tdk_fft
