// tdk_wave_fft.h -- wave64 building block of the tiled-FFT Wiener filter: a K-point complex FFT held entirely
// in one lane's registers (a tile is processed one row per lane; K = 16 or 32 lanes = a "slot", a wave carries
// 64 / K slots), and the frequency -> lane assignment used between the two 1-D passes.
//
// The K x K transposition between the passes goes through a small wave-private LDS buffer (wiener.hip).  A
// cross-lane VALU transposition (v_cndmask_b32_dpp / v_permlane16_swap, tests/hip_unit/dpp_transpose.h) was
// built and verified, but measured slower on gfx950: the VOP2 v_cndmask_b32 -- the only select that takes a
// DPP operand -- costs ~23 cycles whenever it reads VCC, and the two-instruction forms (v_mov_b32_dpp +
// v_cndmask_b32_e64, which is half rate like every VALU instruction with an SGPR operand) need ~290
// issue slots per K = 32 array against ~190 LDS cycles (tests/hip_unit/*_bench.hip).
#pragma once

#include "tdk_common.h"

namespace tdk_fft {

// cos/sin(2 pi k / 32), k = 0..15 (doubles: the butterflies fold tan / cot from them at compile time)
constexpr double TW_COS_D[16] = {1.0, 0.98078528040323043, 0.92387953251128674, 0.83146961230254524, 0.70710678118654757,
                                 0.55557023301960229, 0.38268343236508984, 0.19509032201612833, 0.0, -0.19509032201612819,
                                 -0.38268343236508973, -0.55557023301960196, -0.70710678118654746, -0.83146961230254535,
                                 -0.92387953251128674, -0.98078528040323043};
constexpr double TW_SIN_D[16] = {0.0, 0.19509032201612825, 0.38268343236508978, 0.55557023301960218, 0.70710678118654746,
                                 0.83146961230254524, 0.92387953251128674, 0.98078528040323043, 1.0, 0.98078528040323043,
                                 0.92387953251128674, 0.83146961230254546, 0.70710678118654757, 0.55557023301960218,
                                 0.38268343236508989, 0.19509032201612861};

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n / 2); }
constexpr int bitrev(int v, int bits) {
  int r = 0;
  for (int b = 0; b < bits; b++) r |= ((v >> b) & 1) << (bits - 1 - b);
  return r;
}

// In-register radix-2 decimation-in-time FFT of N complex points (same butterfly network as
// the reference's shuffle FFT, fft.h:133-167).  Forward: e^{-i...}; inverse: e^{+i...}, UNSCALED
// (the caller folds the 1/N of each inverse pass into the Wiener gain).
// A general butterfly a +- w b is written with the twiddle factored as w = c (1 -+ i tan) (or
// s (cot -+ i) when |c| < |s|): two FMAs form b (1 -+ i tan), four more give both outputs -- 6
// instructions instead of the 4 multiplies/FMAs + 4 adds of the textbook form.
template <int N, bool INV> __device__ __forceinline__ void fft_inreg(float (&re)[N], float (&im)[N]) {
  constexpr int STAGES = ilog2(N);
#pragma unroll
  for (int t = 0; t < N; t++) {
    const int r = bitrev(t, STAGES);
    if (t < r) {
      const float a = re[t], b = im[t];
      re[t] = re[r]; im[t] = im[r];
      re[r] = a; im[r] = b;
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
        const float ar = re[t], ai = im[t], br = re[p], bi = im[p];
        if (k == 0) {            // w = 1
          re[t] = ar + br; im[t] = ai + bi;
          re[p] = ar - br; im[p] = ai - bi;
        } else if (k == 8) {     // w = -i (forward) / +i (inverse)
          const float xr = INV ? -bi : bi, xi = INV ? br : -br;
          re[t] = ar + xr; im[t] = ai + xi;
          re[p] = ar - xr; im[p] = ai - xi;
        } else {
          // w = c - i sg (forward), c + i sg (inverse):  w b = (br c + bi sg') + i (bi c - br sg'), sg' = -+sg
          const double cd = TW_COS_D[k], sd = INV ? -TW_SIN_D[k] : TW_SIN_D[k];
          if ((cd < 0 ? -cd : cd) >= (sd < 0 ? -sd : sd)) {
            const float c = (float)cd, tn = (float)(sd / cd);
            const float pr = __builtin_fmaf(tn, bi, br), pi = __builtin_fmaf(-tn, br, bi);   // b (1 - i tn)
            re[t] = __builtin_fmaf(c, pr, ar); im[t] = __builtin_fmaf(c, pi, ai);
            re[p] = __builtin_fmaf(-c, pr, ar); im[p] = __builtin_fmaf(-c, pi, ai);
          } else {
            const float sn = (float)sd, ct = (float)(cd / sd);
            const float pr = __builtin_fmaf(ct, br, bi), pi = __builtin_fmaf(ct, bi, -br);    // b (ct - i)
            re[t] = __builtin_fmaf(sn, pr, ar); im[t] = __builtin_fmaf(sn, pi, ai);
            re[p] = __builtin_fmaf(-sn, pr, ar); im[p] = __builtin_fmaf(-sn, pi, ai);
          }
        }
      }
    }
  }
}

// Frequency held by lane l of a slot after the forward transposition (and consumed before the inverse one):
// f(0) = 0, f(1) = K/2, f(2j) = j, f(2j+1) = K - j.  Lane l ^ 1 then holds -f(l) (mod K) for every l >= 2, and
// lanes 0 and 1 hold the two self-conjugate frequencies: the Hermitian partner of a bin is one quad_perm
// away instead of a ds_bpermute (~24 cycles each on gfx950).  The permutation itself is free: lane l simply
// reads column lane_freq(l) of the transposition buffer.
constexpr int lane_freq(int l, int K) { return l == 0 ? 0 : l == 1 ? K / 2 : (l & 1) ? K - (l >> 1) : (l >> 1); }

}  // namespace tdk_fft
