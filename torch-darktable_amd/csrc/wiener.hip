// wiener.hip -- tiled-FFT Wiener shrinkage denoiser.
//
// Replaces reference csrc/denoise/denoise.cu:84-364, fft.h, window.h (WienerImpl::process:
// one K x K-thread block per tile, shuffle FFT + LDS transposes, and an overlap-add of every
// tile into a padded image with 2 global float atomics per sample, then normalise-and-crop).
// Semantics kept: tile origins (g - ov) * s with s = K / ov, asymmetric reflect load
// (:118-122), per-tile mean, Gaussian analysis window (x - mean) * wf[tx] * wf[ty], forward
// 2-D FFT (along x, then y), gain max(|X|^2 + 1e-15 - sigma^2, 0) / (|X|^2 + 1e-15) (:181-185),
// inverse (along y then x, 1/K per pass), (y + mean * wf2d) * wi2d overlap-added, divided by
// the accumulated wf2d * wi2d mask + 1e-15.
//
// MI355X design.  The op is FP32-vector bound (~2.5 kFLOP/px at ov = 4, against 8-24 B/px), and a wave64
// VALU instruction holds its wave for 4 cycles but the SIMD-32 for ~2, so the kernel is built around waves per
// SIMD: 8.4 KB of LDS per wave (the first generation needed 20 KB: 2 waves per SIMD, each 42 % VALU-active).
//  * a wave carries 64/K "slots" of K lanes; a slot owns one TILE ROW and walks it left to right, two
//    adjacent tiles (a, b) per step riding ONE complex 2-D FFT (z = a + i b, Hermitian split);
//  * lane = row of the tile: the K-point FFTs run entirely in registers (tdk_wave_fft.h: radix-2, 6-FMA
//    butterflies, immediate twiddles); the two transpositions per direction go through a wave-private
//    (K + 1)-stride LDS buffer, the two slots of a K = 32 wave taking turns in ONE 4.2 KB buffer;
//  * the frequency -> lane assignment after the transposition puts the Hermitian partner of every bin in
//    lane ^ 1 (one quad_perm DPP move instead of a ~24-cycle ds_bpermute);
//  * the overlap-add ALONG x happens in registers: a lane keeps the K - s unfinished columns of its row
//    (tiles are s apart, so after tiles a, b the first 2 s columns are final for this tile row) -- no LDS
//    read-modify-write, no atomics (the reference issues ~32 global float atomics per pixel; memory-side
//    float atomics cap at ~1.3 TB/s on MI355X);
//  * the overlap-add ACROSS tile rows: the 4 waves of a workgroup (64/K * 4 consecutive tile rows) drop
//    their 2 s finished columns into their (now free) transposition buffer each step -- the buffers are
//    double-buffered by step parity, so ONE barrier per step -- and all threads fold the <= ov blocks that
//    cover an output row in a fixed order and store 16-B pieces to the group's partial slab.  Only the
//    seams between workgroups (K - s rows / columns) are summed later: a second streaming kernel adds the
//    <= 4 slabs that overlap a pixel, applies the analytic mask mask(x, y) = m1[x mod s] m1[y mod s],
//    m1[r] = sum_k wf[r + k s] wi[r + k s] (every pixel is covered by exactly ov x ov tiles, so the mask is
//    never accumulated), and -- fused -- modify_log_luminance;
//  * one workgroup per group, G tile columns wide; G is chosen per launch so that the groups fill the
//    resident workgroup slots of the chip in whole rounds (12 MP: 735 groups on 768 slots).
//  The whole op is deterministic, unlike the reference's atomic overlap-add.  Windows are evaluated on the
//  host in fp64 and rounded once to fp32 (the reference uses torch fp32 ops on the GPU; both are within
//  an ulp of each other).  This file allows FMA contraction: the reference's sums are
//  order-nondeterministic, parity is by tolerance.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "tdk_color.h"
#include "tdk_wave_fft.h"
#include "tdk_wave_fft_pk.h"

#pragma clang fp contract(fast)

namespace {

using tdk_fft::fft_inreg;
using tdk_fft::lane_freq;

constexpr int NWV = 4;  // waves per workgroup of the tile kernel
#ifndef TDK_WIENER_WAVES_PER_SIMD
#define TDK_WIENER_WAVES_PER_SIMD 3  // 159 VGPRs, no spills; 4 (128 VGPRs) spills 31 registers and measured 9 % slower
#endif

// The Gaussian analysis / synthesis window (window.h:22-35 with weight 0.3, the only weight the reference uses:
// denoise.cu:258) as compile-time constants: a VALU instruction with a LITERAL operand issues at full rate on gfx950,
// one with an SGPR operand (a kernel-argument table) at half rate (tests/hip_unit/valu_issue_bench.hip) -- and the
// tile kernel multiplies by a window value ~400 times per step.  make_params checks them against make_window().
template <int K> struct WindowK;
template <> struct WindowK<16> {
  static constexpr float w[16] = {0.0227954239f, 0.0472629406f, 0.0882987976f, 0.148644835f, 0.225478873f, 0.308193088f, 0.379577875f, 0.421249986f,
                                  0.421249986f, 0.379577875f, 0.308193088f, 0.225478873f, 0.148644835f, 0.0882987976f, 0.0472629406f, 0.0227954239f};
};
template <> struct WindowK<32> {
  static constexpr float w[32] = {0.0132160187f, 0.01953201f, 0.0281244125f, 0.0394557454f, 0.0539296083f, 0.0718182027f, 0.093182005f, 0.117793098f,
                                  0.145076767f, 0.174086913f, 0.203528211f, 0.231831998f, 0.257283777f, 0.278190076f, 0.293063104f, 0.300795197f,
                                  0.300795197f, 0.293063104f, 0.278190076f, 0.257283777f, 0.231831998f, 0.203528211f, 0.174086913f, 0.145076767f,
                                  0.117793098f, 0.093182005f, 0.0718182027f, 0.0539296083f, 0.0394557454f, 0.0281244125f, 0.01953201f, 0.0132160187f};
};

struct WParams {
  float wf[32];  // analysis (FFT) window
  float wi[32];  // synthesis (interpolation) window
  float m1[16];  // separable mask factor, index p mod s
};

// Slab geometry.  Tile t (0-based, origin index jmin + t) covers padded coordinates u in [t s, t s + K),
// u = pixel + (ov - 1) s.  A group = TR tile rows x G tile columns; its slab holds the partial sums of
// u_y in [gy BSY, gy BSY + RSY), u_x in [gx BSX, gx BSX + RSX), row stride RSXP (a multiple of 4 floats).
struct Geom {
  int s, K, jmin, ntx, nty, TR, G, ngx, ngy, BSX, BSY, RSX, RSXP, RSY;
  unsigned magic_x, magic_y;  // ceil(2^32 / BSX), ceil(2^32 / BSY): u / BS == umulhi(u, magic) for u < 2^32 / BS
};

__device__ __forceinline__ int reflect_index(int x, int limit) {
  x = x < 0 ? -x : x;
  return min(x, 2 * limit - 1 - x);  // == (x >= limit) ? 2 * limit - x - 1 : x, without the select
}

// x <- (lane is one of the slot's two self-conjugate lanes) ? x : x of lane ^ 1: v_mov_b32_dpp + v_cndmask_b32_e64
// with the lane mask in an SGPR pair.  (Measured on gfx950, tests/hip_unit/dpp_bench*.hip: the VOP2 form of
// v_cndmask_b32 -- the only one that takes a DPP operand -- costs ~23 cycles whenever it reads VCC, a
// ds_bpermute_b32 ~24; v_mov_b32_dpp and the e64 select ~4.5 each.)  Inline asm because hipcc sinks a DPP move
// into an EXEC-masked branch of the select, where the disabled source lanes read as 0; it inserts no wait
// states around asm, hence the s_nop (VALU write -> DPP read of the same register needs two).
__device__ __forceinline__ void partner2(float& a, float& b, unsigned long long self) {
  float ta, tb;
  asm volatile("s_nop 1\n\t"
               "v_mov_b32_dpp %[ta], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_mov_b32_dpp %[tb], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_cndmask_b32_e64 %[a], %[ta], %[a], %[m]\n\t"
               "v_cndmask_b32_e64 %[b], %[tb], %[b], %[m]"
               : [a] "+v"(a), [b] "+v"(b), [ta] "=&v"(ta), [tb] "=&v"(tb) : [m] "s"(self));
}

// Sum over the K lanes of a slot, result in every lane: quad butterflies and row rotations as DPP operands of the adds
// (v_add_f32_dpp, ~4.5 cycles), one ds_bpermute (~24 cycles, through the LDS pipe) only for the two 16-lane rows of a
// K = 32 slot -- the plain __shfl_xor butterfly is five of them.
template <int K> __device__ __forceinline__ float slot_sum(float v) {
  auto dpp = [](float x, auto CTRL) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(CTRL)::value, 0xF, 0xF, true)); };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x124>{});  // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{});  // row_ror:8  -> sum of the 16-lane row
  if constexpr (K == 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// Transpose the K x K tile each slot holds one row per lane through a wave-private LDS buffer (LDS operations
// of one wave execute in issue order, so only compiler fences are needed, no barrier).  K = 16: the four
// slots of the wave go at once (4 x 16 x 17 floats); K = 32: the two slots take turns in one 32 x 36 buffer --
// half the footprint, and a half-active wave's LDS access costs half the LDS cycles.  The read side
// applies the frequency -> lane assignment for free: lane l reads column lane_freq(l).
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int K, bool INV> __device__ __forceinline__ void transpose_lds(float (&v)[K], float* buf, int slot, int row, int col) {
  // forward: write [row][k], read [k][col];  inverse: write [k][col], read [row][k]   (col = lane_freq(row))
  if constexpr (K == 16) {
    constexpr int ST = K + 1;
    const int wbase = INV ? col : row * ST, rbase = INV ? row * ST : col;
    constexpr int WS = INV ? ST : 1, RS = INV ? 1 : ST;
    float* my = buf + slot * (K * ST);
#pragma unroll
    for (int k = 0; k < K; k++) my[wbase + k * WS] = v[k];
    wave_fence();
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = my[rbase + k * RS];
    wave_fence();
  } else {
    // K = 32: the two slots take turns in one 32 x TST buffer.  The side on which a lane walks its own row is
    // contiguous: 16-B accesses there (row stride 36 floats = 9 x 16 B: the eight lanes of a ds_*_b128 group hit eight
    // different 4-bank slots), 4-B accesses on the column side (lanes on consecutive columns).
    constexpr int ST = 36;
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
      if (slot == sl) {
        if constexpr (!INV) {
#pragma unroll
          for (int k = 0; k < K; k += 4) *reinterpret_cast<float4*>(buf + row * ST + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
        } else {
#pragma unroll
          for (int k = 0; k < K; k++) buf[k * ST + col] = v[k];
        }
      }
      wave_fence();
      if (slot == sl) {
        if constexpr (!INV) {
#pragma unroll
          for (int k = 0; k < K; k++) v[k] = buf[k * ST + col];
        } else {
#pragma unroll
          for (int k = 0; k < K; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(buf + row * ST + k);
            v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
          }
        }
      }
      wave_fence();
    }
  }
}

// Separate the spectra of the two real tiles packed as z = a + i b, apply the gains (denoise.cu:181-185), recombine.  With
// 2A = Z[k] + conj(Z[-k]) and 2B = -i (Z[k] - conj(Z[-k])):  Z'[k] = ga A + i gb B and, because A and B
// are spectra of real tiles, Z'[-k] = ga conj(A) + i gb conj(B) -- the same gains and products.  Z[-k]
// lives in the partner lane (column -kx) at register -ky and the partner needs exactly the mirrored
// pair, so every lane evaluates only ky = k (k = 0..K/2) plus the by-product for the partner's register
// K - k, and the two lanes swap by-products: half the gain arithmetic of evaluating every bin.  The 1/2
// of A, B and the 1/K^2 of the two unscaled inverse passes are powers of two folded into the gain.
// gain = max(1 - sigma^2 / p, 0) with p = |A|^2 + eps, evaluated on p4 = |2A|^2 + 4 eps as
// max(GS - (4 sigma^2 GS) / p4, 0): two FMAs for p4, then rcp + FMA + max.
// Input: lane = kx (frequency -> lane assignment lane_freq, Hermitian partner in lane ^ 1), register = ky, after the two
// unscaled forward passes; output: ready for the two unscaled inverse passes.
template <int K> __device__ __forceinline__ void wiener_gains(float (&re)[K], float (&im)[K], float sig2) {
  // the two self-conjugate lanes (kx = 0 and K/2) of every slot
  constexpr unsigned long long SELF = (K == 32) ? 0x0000000300000003ull : 0x0003000300030003ull;
  constexpr float GSCALE = 0.5f / (float)(K * K);
  float sgs = -4.0f * sig2 * GSCALE;
  asm volatile("" : "+v"(sgs));  // keep it in a VGPR: a VALU instruction with an SGPR operand issues at half rate
#pragma unroll
  for (int k = 0; k <= K / 2; k++) {
    const int k2 = (K - k) & (K - 1);
    const float zr = re[k], zi = im[k];
    float pr = re[k2], pi = im[k2];
    partner2(pr, pi, SELF);  // Z[-k]
    const float a2r = zr + pr, a2i = zi - pi, b2r = zi + pi, b2i = pr - zr;
    const float pa4 = __builtin_fmaf(a2i, a2i, __builtin_fmaf(a2r, a2r, 4e-15f)), pb4 = __builtin_fmaf(b2i, b2i, __builtin_fmaf(b2r, b2r, 4e-15f));
    const float ga = fmaxf(__builtin_fmaf(sgs, __builtin_amdgcn_rcpf(pa4), GSCALE), 0.0f);
    const float gb = fmaxf(__builtin_fmaf(sgs, __builtin_amdgcn_rcpf(pb4), GSCALE), 0.0f);
    const float gar = ga * a2r, gai = ga * a2i, gbr = gb * b2r, gbi = gb * b2i;
    re[k] = gar - gbi; im[k] = gai + gbr;                 // Z'[k]
    if (k2 != k) {
      float br_ = gar + gbi, bi_ = gbr - gai;             // my Z'[K-k] is the partner's by-product
      partner2(br_, bi_, SELF);
      re[k2] = br_; im[k2] = bi_;
    }
  }
}

// One workgroup = one group: TR = 4 * (64 / K) consecutive tile rows x G tile columns of one plane.
// NBUF = 2 double-buffers the hand-off block (one barrier per step); 1 when that would not leave room
// for two workgroups per CU (K = 32, ov = 2: 32 KB per buffer).
template <typename T, int K, int OV>
__global__ __launch_bounds__(64 * NWV, TDK_WIENER_WAVES_PER_SIMD) void wiener_stream(const T* __restrict__ img, float* __restrict__ slabs, int W, int H, int C, int chan0, int vec_ok,
                                                         Geom g, const float* __restrict__ sigmas, WParams prm, int nplanes, size_t plane_stride) {
  constexpr int S = K / OV, TPW = 64 / K, TR = NWV * TPW;
  constexpr int EM = 2 * S;      // columns of a tile row that become final per step (tiles a and b)
  constexpr int CAR = K - S;     // unfinished columns carried to the next step
  constexpr int WIN = K + S;     // input samples of one step: tile a = w[0, K), tile b = w[S, K + S)
  constexpr int NQ = EM / 4;     // 16-B pieces per emitted row
  constexpr int SH = EM >= 32 ? 0 : (EM == 16 ? 1 : (EM == 8 ? 2 : 3));  // rows that share the 32 banks
  constexpr int FLUSH = (CAR + EM - 1) / EM;
  // Per-wave LDS region: the transposition buffer during a step, then the wave's EM finished columns for the fold.
  constexpr int TBUF = (K == 16) ? 4 * 16 * 17 : 32 * 36, EBLK = TPW * K * EM;
  constexpr int REG = ((TBUF > EBLK ? TBUF : EBLK) + 3) & ~3;
  constexpr int NBUF = (2 * REG * 4 * 16 <= 160 * 1024) ? 2 : 1;  // two regions (one barrier per step) if 16 waves still fit a CU
  __shared__ __align__(16) float lds[NBUF * NWV * REG];

  int chan = chan0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int slot = lane / K, row = lane & (K - 1);
  const int sidx = wave * TPW + slot;  // tile row inside the band
  const int fcol = (row == 0) ? 0 : (row == 1) ? K / 2 : (row & 1) ? K - (row >> 1) : (row >> 1);  // == lane_freq(row, K)
  const int ngroups = g.ngx * g.ngy;
  const int plane = blockIdx.x / ngroups, gin = blockIdx.x - plane * ngroups;
  const int gy = gin / g.ngx, gx = gin - gy * g.ngx;
  const int t_row = gy * TR + sidx;
  const bool row_active = t_row < g.nty;
  const int jy = g.jmin + t_row;
  const T* src_row = img + (size_t)plane * plane_stride + (size_t)reflect_index(row_active ? jy * S + row : 0, H) * W * C;
  const float sigma = sigmas[chan + plane];
  if (C == 3) chan += plane;  // interleaved input: group set p works on channel p of the same image
  const float sig2 = sigma * sigma;
  const float wy = prm.wf[row], iy = prm.wi[row];
  float* slab = slabs + (size_t)blockIdx.x * (size_t)(g.RSY * g.RSXP);
  const int steps = g.G >> 1;

  float carry[CAR];
#pragma unroll
  for (int i = 0; i < CAR; i++) carry[i] = 0.0f;

  for (int m = 0; m < steps + FLUSH; m++) {
    float S_[WIN];  // sums over this tile row's tiles for columns [2 m S, 2 m S + K + S)
    if (m < steps) {
      const int ta = gx * g.G + 2 * m;
      const bool act_a = row_active && ta < g.ntx, act_b = row_active && ta + 1 < g.ntx;
      const int ox = (g.jmin + ta) * S;
      float re[K], im[K], mwa, mwb;
      {
        // ---- load the K + S samples of this lane's image row, per-tile means, analysis window
        float w[WIN];
        if (vec_ok && C == 3 && ox >= 0 && ox + ((WIN + 3) & ~3) <= W) {
          // interleaved RGB: read whole pixels (4 pixels = 12 values per chunk, 16-B / 8-B vector loads) and keep this
          // group's channel -- no de-interleaving pass over the image, no fp32 planes (600 MB at 50 MP)
          auto pick = [&](auto CH) {
#pragma unroll
            for (int k = 0; k < ((WIN + 3) & ~3); k += 4) {
              float t12[12];
              rgb4_io<T>::load(src_row, (size_t)((ox + k) >> 2), t12);
#pragma unroll
              for (int j = 0; j < 4; j++)
                if (k + j < WIN) w[k + j] = t12[3 * j + decltype(CH)::value];
            }
          };
          if (chan == 0) pick(std::integral_constant<int, 0>{});
          else if (chan == 1) pick(std::integral_constant<int, 1>{});
          else pick(std::integral_constant<int, 2>{});
        } else if (vec_ok && C == 1 && ox >= 0 && ox + ((WIN + 3) & ~3) <= W) {
          const T* p = src_row + ox;
#pragma unroll
          for (int k = 0; k + 4 <= WIN; k += 4) {
            float t4[4];
            s4_io<T>::load(p, k / 4, t4);
            w[k] = t4[0]; w[k + 1] = t4[1]; w[k + 2] = t4[2]; w[k + 3] = t4[3];
          }
#pragma unroll
          for (int k = WIN & ~3; k < WIN; k++) w[k] = ld(p, k);
        } else {
#pragma unroll
          for (int k = 0; k < WIN; k++) {
            const bool need = (k < K) ? act_a : act_b;  // an inactive tile may lie beyond one reflection of the frame
            w[k] = need ? ld(src_row, (size_t)reflect_index(ox + k, W) * C + chan) : 0.0f;
          }
        }
        if (!act_a) {
#pragma unroll
          for (int k = 0; k < K; k++) w[k] = 0.0f;
        }
        if (!act_b) {
#pragma unroll
          for (int k = K; k < WIN; k++) w[k] = 0.0f;
        }
        float sa = 0.0f, sb = 0.0f, head = 0.0f, tail = 0.0f;
#pragma unroll
        for (int k = 0; k < S; k++) head += w[k];
#pragma unroll
        for (int k = S; k < K; k++) sa += w[k];
#pragma unroll
        for (int k = K; k < WIN; k++) tail += w[k];
        sb = act_b ? sa + tail : 0.0f;  // tile b = columns [S, K + S)
        sa = sa + head;                 // tile a = columns [0, K)
        sa = slot_sum<K>(sa);
        sb = slot_sum<K>(sb);
        const float mean_a = sa / (float)(K * K), mean_b = sb / (float)(K * K);
        const float wyb = act_b ? wy : 0.0f;  // tile b absent: its row factor is 0 (one select instead of K)
        const float ca = -mean_a * wy, cb = -mean_b * wyb;
#pragma unroll
        for (int k = 0; k < K; k++) {  // (x - mean) * wf[ty] * wf[tx]; the per-column factor is a literal
          re[k] = __builtin_fmaf(w[k], wy, ca) * WindowK<K>::w[k];
          im[k] = __builtin_fmaf(w[k + S], wyb, cb) * WindowK<K>::w[k];
        }
        // mean * wf2d * wi2d is added back after the inverse transform: keep the per-lane factors
        mwa = mean_a * (wy * iy);
        mwb = mean_b * (wy * iy);
      }

      // ---- forward 2-D FFT of z = a + i b: along x in registers, transpose, along y in registers.
      // Lane l reads column kx = lane_freq(l) back: the partner bin -kx then sits in lane l ^ 1.
      float* tbuf = lds + ((NBUF == 2 ? (m & 1) : 0) * NWV + wave) * REG;
      fft_inreg<K, false>(re, im);
      transpose_lds<K, false>(re, tbuf, slot, row, fcol);
      transpose_lds<K, false>(im, tbuf, slot, row, fcol);
      fft_inreg<K, false>(re, im);  // lane = kx, register = ky

      wiener_gains<K>(re, im, sig2);  // Hermitian split of the two tiles, gains, recombination

      // ---- inverse: along y, transpose back (lane = y again, register = kx in natural order), along x
      fft_inreg<K, true>(re, im);
      transpose_lds<K, true>(re, tbuf, slot, row, fcol);
      transpose_lds<K, true>(im, tbuf, slot, row, fcol);
      fft_inreg<K, true>(re, im);  // re = tile a, im = tile b (row `row` of each)

      // ---- overlap-add along x in registers: (v + mean wf[tx] wf[ty]) * wi[tx] wi[ty]  (denoise.cu:172-175)
      // written wi[k] * (v * wi[ty] + (mean wf[ty] wi[ty]) * wf[k]): every per-column factor is a scalar operand
#pragma unroll
      for (int u = 0; u < WIN; u++) {
        float acc = (u < CAR) ? carry[u] : 0.0f;
        if (u < K) acc = __builtin_fmaf(WindowK<K>::w[u], __builtin_fmaf(mwa, WindowK<K>::w[u], re[u] * iy), acc);
        if (u >= S) acc = __builtin_fmaf(WindowK<K>::w[u - S], __builtin_fmaf(mwb, WindowK<K>::w[u - S], im[u - S] * iy), acc);
        S_[u] = acc;
      }
    } else {
      // flush: no more tiles in this group, the carried columns leave as they are (partial sums for the seam)
#pragma unroll
      for (int u = 0; u < WIN; u++) S_[u] = (u < CAR) ? carry[u] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < CAR; i++) carry[i] = S_[EM + i];

    // ---- hand the EM finished columns of this tile row to the workgroup: one 16-B piece per quad, XOR-swizzled
    // so that the 8 lanes of a ds_write_b128 group hit 8 different 4-bank slots
    float* buf = lds + (NBUF == 2 ? (m & 1) : 0) * (NWV * REG);  // this step's region of wave 0; wave w's is w * REG further
    {
      float* dst = buf + wave * REG + (slot * K + row) * EM;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int qs = q ^ ((row >> SH) & (NQ - 1));
        *reinterpret_cast<float4*>(dst + 4 * qs) = make_float4(S_[4 * q], S_[4 * q + 1], S_[4 * q + 2], S_[4 * q + 3]);
      }
    }
    __syncthreads();
    // ---- fold across the band's tile rows: band row r gets tile row sg's row r - sg S for the (at most ov)
    // tile rows with 0 <= r - sg S < K, summed in increasing sg, and goes to the slab as 16-B pieces
    {
      const int col0 = m * EM;
      for (int i = tid; i < g.RSY * NQ; i += 64 * NWV) {
        const int r = i / NQ, q = i - r * NQ;
        if (col0 + 4 * q >= g.RSXP) continue;
        const int s_hi = min(r / S, TR - 1), s_lo = (r >= K) ? (r - K) / S + 1 : 0;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int sg = s_lo; sg <= s_hi; sg++) {
          const int lr = r - sg * S;
          const int qs = q ^ ((lr >> SH) & (NQ - 1));
          const float4 a = *reinterpret_cast<const float4*>(buf + (sg / TPW) * REG + ((sg % TPW) * K + lr) * EM + 4 * qs);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *reinterpret_cast<float4*>(slab + (size_t)r * g.RSXP + col0 + 4 * q) = v;
      }
    }
    if (NBUF == 1) __syncthreads();  // single buffer: the fold must finish before the next step overwrites it
  }
}

// ---------------------------------------------------------------- finish kernels
// Position of padded coordinate u along one axis: group index, offset inside the group, and whether the
// previous group's slab also covers it (its last K - s rows / columns).
struct AxisPos {
  int grp, off;
  bool prev;
};
__device__ __forceinline__ AxisPos axis_pos(int u, int BS, unsigned magic, int overlap) {
  AxisPos p;
  p.grp = (int)__umulhi((unsigned)u, magic);
  p.off = u - p.grp * BS;
  p.prev = (p.off < overlap) && p.grp > 0;
  return p;
}

// Sum of the <= 4 slabs that cover pixel (x, row): row-level pointers row0 (group row gy) / row1 (group row
// gy - 1, or null) already include the row offset.
__device__ __forceinline__ float fold4(const float* __restrict__ row0, const float* __restrict__ row1, int x, int u0, const Geom& g, size_t slab_sz) {
  const AxisPos px = axis_pos(x + u0, g.BSX, g.magic_x, g.K - g.s);
  float v = row0[px.grp * slab_sz + px.off];
  if (px.prev) v += row0[(px.grp - 1) * slab_sz + px.off + g.BSX];
  if (row1) {
    v += row1[px.grp * slab_sz + px.off];
    if (px.prev) v += row1[(px.grp - 1) * slab_sz + px.off + g.BSX];
  }
  return v;
}

// The same fold for four consecutive pixels x0 .. x0 + 3 with (x0 + u0) % 4 == 0 when s % 4 == 0: group width, overlap
// and pixel offset are then multiples of 4, so the four pixels sit in the same slabs at a 16-B aligned offset -- one
// 16-B load per slab instead of four scattered 4-B loads (the per-pixel form made the finish kernels
// texture-addresser bound).  Same values, same order of additions per pixel.
__device__ __forceinline__ float4 fold4x4(const float* __restrict__ row0, const float* __restrict__ row1, int x0, int u0, const Geom& g, size_t slab_sz) {
  const AxisPos px = axis_pos(x0 + u0, g.BSX, g.magic_x, g.K - g.s);
  auto at = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
  auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
  float4 v = at(row0 + px.grp * slab_sz + px.off);
  if (px.prev) add(v, at(row0 + (px.grp - 1) * slab_sz + px.off + g.BSX));
  if (row1) {
    add(v, at(row1 + px.grp * slab_sz + px.off));
    if (px.prev) add(v, at(row1 + (px.grp - 1) * slab_sz + px.off + g.BSX));
  }
  return v;
}
__device__ __forceinline__ bool fold_vec_ok(const float* slabs, const Geom& g) { return (g.s & 3) == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0; }

struct RowPtrs {
  const float *row0, *row1;
  float my;
};
__device__ __forceinline__ RowPtrs row_ptrs(const float* __restrict__ slabs, int y, int u0, const Geom& g, size_t slab_sz, const WParams& prm) {
  const AxisPos py = axis_pos(y + u0, g.BSY, g.magic_y, g.K - g.s);
  RowPtrs r;
  r.row0 = slabs + (size_t)py.grp * g.ngx * slab_sz + (size_t)py.off * g.RSXP;
  r.row1 = py.prev ? slabs + (size_t)(py.grp - 1) * g.ngx * slab_sz + (size_t)(py.off + g.BSY) * g.RSXP : nullptr;
  r.my = prm.m1[y & (g.s - 1)];
  return r;
}

// Sum the overlapping slabs of one channel, normalise by the analytic mask, crop.  One workgroup
// walks whole image rows.
template <typename T>
__global__ __launch_bounds__(256) void wiener_finish(const float* __restrict__ slabs, T* __restrict__ out, int W, int H, int C, int chan, Geom g, WParams prm) {
  const int u0 = -g.jmin * g.s;  // = (ov - 1) * s: pixel 0 sits at this offset inside group 0
  const size_t slab_sz = (size_t)g.RSXP * g.RSY;
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const RowPtrs rp = row_ptrs(slabs, y, u0, g, slab_sz, prm);
    for (int x = blockIdx.x * 256 + threadIdx.x; x < W; x += gridDim.x * 256) {
      const float v = fold4(rp.row0, rp.row1, x, u0, g, slab_sz);
      const float mask = prm.m1[x & (g.s - 1)] * rp.my;
      st(out, ((size_t)y * W + x) * C + chan, v * __builtin_amdgcn_rcpf(mask + 1e-15f));  // 1-ulp reciprocal: an IEEE divide is ~20 issue slots
    }
  }
}

// Fused epilogue of Wiener.process_log_luminance (reference denoise.py:54-58): the slab fold and
// normalisation of wiener_finish followed directly by modify_log_luminance on the RGB pixel, so the
// denoised log-luminance plane is never written to HBM.  The colour math must not be contracted.
// lum_out (nullable): also write the fp32 (log-)lightness plane of the pixels as stored -- exactly what
// compute_[log_]luminance(out) would produce -- for a consumer that extracts it next (Bilateral.process_rgb).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void wiener_finish_modify(const float* __restrict__ slabs, const T* __restrict__ rgb, T* __restrict__ out, int W, int H,
                                                            Geom g, WParams prm, float* __restrict__ lum_out, int lum_log, float lum_eps) {
#pragma clang fp contract(off)
  const int u0 = -g.jmin * g.s;
  const size_t slab_sz = (size_t)g.RSXP * g.RSY;
  const int ngroup = W / VEC;  // VEC == 4 requires W % 4 == 0: a group never straddles a row
  const bool vfold = fold_vec_ok(slabs, g);
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const RowPtrs rp = row_ptrs(slabs, y, u0, g, slab_sz, prm);
    for (int gi_ = blockIdx.x * 256 + threadIdx.x; gi_ < ngroup; gi_ += gridDim.x * 256) {
      const int x0 = gi_ * VEC;
      const size_t gi = (size_t)y * ngroup + gi_;
      float v[3 * VEC], l[VEC], accs[VEC];
      if constexpr (VEC == 4) rgb4_io<T>::load(rgb, gi, v);
      else { v[0] = ld(rgb, gi * 3); v[1] = ld(rgb, gi * 3 + 1); v[2] = ld(rgb, gi * 3 + 2); }
      if (VEC == 4 && vfold) {
        const float4 a4 = fold4x4(rp.row0, rp.row1, x0, u0, g, slab_sz);
        accs[0] = a4.x;
        if constexpr (VEC == 4) { accs[1] = a4.y; accs[2] = a4.z; accs[3] = a4.w; }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) accs[k] = fold4(rp.row0, rp.row1, x0 + k, u0, g, slab_sz);
      }
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        const int x = x0 + k;
        const float acc = accs[k];
        const float mask = prm.m1[x & (g.s - 1)] * rp.my;
        const f3 r = cA::modify_log_luminance(mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2]), acc * __builtin_amdgcn_rcpf(mask + 1e-15f));
        v[3 * k] = r.x; v[3 * k + 1] = r.y; v[3 * k + 2] = r.z;
        if (lum_out) {  // wave-uniform; same expression as lum_extract_vec4 (color.hip)
          const float yv = cA::rgb_to_lab_l(clip3(mk3(as_stored<T>(r.x), as_stored<T>(r.y), as_stored<T>(r.z))));
          l[k] = lum_log ? tdk_log(fmaxf(lum_eps, yv)) : yv;
        }
      }
      if constexpr (VEC == 4) rgb4_io<T>::store(out, gi, v);
      else { st(out, gi * 3, v[0]); st(out, gi * 3 + 1, v[1]); st(out, gi * 3 + 2, v[2]); }
      if (lum_out) {
        if constexpr (VEC == 4) s4_io<float>::store(lum_out, gi, l);
        else lum_out[gi] = l[0];
      }
    }
  }
}

// Finish of the Lab hand-over chain (color.hip: lum_lab_extract): fold + normalise as above, then the lightness of the pixel
// modify_log_luminance would produce -- WITHOUT producing it.  L' = clamp(exp(log L'), 0, 1) with the pixel's chroma (a, b) is
// converted to LINEAR RGB (three cubes and a 3 x 3 matrix, no transcendental); the reference clips the sRGB-encoded pixel to
// [0, 1] (device_conversions.h:213-225), the encoding is monotone with 0 -> 0 and 1 -> 1, so a pixel is clipped iff a linear
// channel leaves [0, 1].  Not clipped (all but ~5e-4 of the pixels of a natural frame): the result's lightness is L' and its
// chroma is unchanged.  Clipped: L, a, b are re-derived from the clipped linear pixel (three cube roots) and (a, b) is rewritten.
// Output: the fp32 lightness plane of the result (what compute_luminance(result) gives) + the (a, b) plane, for tdk_bilateral_lab.
template <int VEC>
__global__ __launch_bounds__(256) void wiener_finish_lab(const float* __restrict__ slabs, float* __restrict__ ab, float* __restrict__ lum_out, int W, int H,
                                                         Geom g, WParams prm) {
  TDK_STREAMING_KERNEL_PROLOGUE();
  const int u0 = -g.jmin * g.s;
  const size_t slab_sz = (size_t)g.RSXP * g.RSY;
  const int ngroup = W / VEC;
  const bool vfold = fold_vec_ok(slabs, g);
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const RowPtrs rp = row_ptrs(slabs, y, u0, g, slab_sz, prm);
    for (int gi_ = blockIdx.x * 256 + threadIdx.x; gi_ < ngroup; gi_ += gridDim.x * 256) {
      const int x0 = gi_ * VEC;
      const size_t gi = (size_t)y * ngroup + gi_;
      float c2[2 * VEC], l[VEC], accs[VEC];
      if constexpr (VEC == 4) { s4_io<float>::load(ab, 2 * gi, c2); s4_io<float>::load(ab, 2 * gi + 1, c2 + 4); }
      else { c2[0] = ab[2 * gi]; c2[1] = ab[2 * gi + 1]; }
      if (VEC == 4 && vfold) {
        const float4 a4 = fold4x4(rp.row0, rp.row1, x0, u0, g, slab_sz);
        accs[0] = a4.x;
        if constexpr (VEC == 4) { accs[1] = a4.y; accs[2] = a4.z; accs[3] = a4.w; }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) accs[k] = fold4(rp.row0, rp.row1, x0 + k, u0, g, slab_sz);
      }
      f3 lin[VEC];
      bool clipped = false;
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        const float mask = prm.m1[(x0 + k) & (g.s - 1)] * rp.my;
        l[k] = fmaxf(0.0f, fminf(1.0f, tdk_exp(accs[k] * __builtin_amdgcn_rcpf(mask + 1e-15f))));
        lin[k] = xyz_to_rgb_lin(cA::lab_to_xyz(mk3(l[k], c2[2 * k], c2[2 * k + 1])));
        clipped = clipped || !(lin[k].x >= 0.0f && lin[k].x <= 1.0f && lin[k].y >= 0.0f && lin[k].y <= 1.0f && lin[k].z >= 0.0f && lin[k].z <= 1.0f);
      }
      const bool any_clipped = __builtin_amdgcn_ballot_w64(clipped) != 0;
      if (any_clipped) {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const f3 q = lin[k];
          if (!(q.x >= 0.0f && q.x <= 1.0f && q.y >= 0.0f && q.y <= 1.0f && q.z >= 0.0f && q.z <= 1.0f)) {
            const f3 lab = cA::xyz_to_lab(rgb_to_xyz_lin(clip3(q)));
            l[k] = fmaxf(0.0f, lab.x);
            c2[2 * k] = lab.y; c2[2 * k + 1] = lab.z;
          }
        }
        if (clipped) {
          if constexpr (VEC == 4) { s4_io<float>::store(ab, 2 * gi, c2); s4_io<float>::store(ab, 2 * gi + 1, c2 + 4); }
          else { ab[2 * gi] = c2[0]; ab[2 * gi + 1] = c2[1]; }
        }
      }
      if constexpr (VEC == 4) s4_io<float>::store(lum_out, gi, l);
      else lum_out[gi] = l[0];
    }
  }
}

// Finish for three planes at once: fold the slabs of each channel, normalise, write interleaved RGB.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void wiener_finish3(const float* __restrict__ slabs, T* __restrict__ out, int W, int H, Geom g, WParams prm) {
  const int u0 = -g.jmin * g.s;
  const size_t slab_sz = (size_t)g.RSXP * g.RSY, chan_sz = slab_sz * g.ngx * g.ngy;
  const int ngroup = W / VEC;
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const RowPtrs rp = row_ptrs(slabs, y, u0, g, slab_sz, prm);
    for (int gi_ = blockIdx.x * 256 + threadIdx.x; gi_ < ngroup; gi_ += gridDim.x * 256) {
      const size_t gi = (size_t)y * ngroup + gi_;
      float v[3 * VEC];
      if (VEC == 4 && fold_vec_ok(slabs, g) && (chan_sz & 3) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float4 a4 = fold4x4(rp.row0 + c * chan_sz, rp.row1 ? rp.row1 + c * chan_sz : nullptr, gi_ * VEC, u0, g, slab_sz);
          v[c] = a4.x;
          if constexpr (VEC == 4) { v[3 + c] = a4.y; v[6 + c] = a4.z; v[9 + c] = a4.w; }
        }
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const float rnorm = __builtin_amdgcn_rcpf(prm.m1[(gi_ * VEC + k) & (g.s - 1)] * rp.my + 1e-15f);
#pragma unroll
          for (int c = 0; c < 3; c++) v[3 * k + c] *= rnorm;
        }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const int x = gi_ * VEC + k;
          const float rnorm = __builtin_amdgcn_rcpf(prm.m1[x & (g.s - 1)] * rp.my + 1e-15f);
#pragma unroll
          for (int c = 0; c < 3; c++) v[3 * k + c] = fold4(rp.row0 + c * chan_sz, rp.row1 ? rp.row1 + c * chan_sz : nullptr, x, u0, g, slab_sz) * rnorm;
        }
      }
      if constexpr (VEC == 4) rgb4_io<T>::store(out, gi, v);
      else { st(out, gi * 3, v[0]); st(out, gi * 3 + 1, v[1]); st(out, gi * 3 + 2, v[2]); }
    }
  }
}

#include "tdk_wiener_ystream.h"

void make_window(int K, double weight, float* w) {
  const double half = K / 2.0, scale = weight * half * half;
  double v[32], nrm = 0.0;
  for (int i = 0; i < K; i++) {
    const double r = -half + 0.5 + i;
    v[i] = exp(-(r * r) / scale);
    nrm += v[i] * v[i];
  }
  nrm = sqrt(nrm);
  for (int i = 0; i < K; i++) w[i] = (float)(v[i] / nrm);
}

constexpr int G_MIN = 8, G_MAX = 64;  // tile columns per group (even)

// G == 0: the worst case for the workspace size (smallest groups = largest seam overhead).
Geom geometry(int W, int H, int K, int ov, int G) {
  Geom g = {};
  g.s = K / ov;
  g.K = K;
  g.jmin = -(ov - 1);                                  // first origin index whose tile covers pixel 0
  g.ntx = (W - 1) / g.s - g.jmin + 1;                  // origins jmin .. floor((W-1)/s)
  g.nty = (H - 1) / g.s - g.jmin + 1;
  g.TR = NWV * (64 / K);
  g.G = G > 0 ? G : G_MIN;
  g.ngx = tdk_div_up(g.ntx, g.G);
  g.ngy = tdk_div_up(g.nty, g.TR);
  g.BSX = g.G * g.s;
  g.BSY = g.TR * g.s;
  g.RSX = g.BSX + K - g.s;
  g.RSXP = (g.RSX + 3) & ~3;
  g.RSY = g.BSY + K - g.s;
  g.magic_x = (unsigned)((0x100000000ull + g.BSX - 1) / g.BSX);
  g.magic_y = (unsigned)((0x100000000ull + g.BSY - 1) / g.BSY);
  return g;
}

size_t slab_floats(const Geom& g) { return (size_t)g.ngx * g.ngy * g.RSXP * g.RSY; }

// Geometry of the y-streaming kernel (tdk_wiener_ystream.h; K = 32, ov = 4): a group = a strip of 16 tile columns x TR
// tile rows, same slab layout.
constexpr int YS_TR_MIN = 8;
Geom geometry_ys(int W, int H, int TR) {
  Geom g = {};
  g.s = ys::S;
  g.K = ys::K;
  g.jmin = -(ys::K / ys::S - 1);
  g.ntx = (W - 1) / g.s - g.jmin + 1;
  g.nty = (H - 1) / g.s - g.jmin + 1;
  g.TR = TR;
  g.G = ys::NTC;
  g.ngx = tdk_div_up(g.ntx, g.G);
  g.ngy = tdk_div_up(g.nty, g.TR);
  g.BSX = g.G * g.s;
  g.BSY = g.TR * g.s;
  g.RSX = g.BSX + g.K - g.s;
  g.RSXP = (g.RSX + 3) & ~3;
  g.RSY = g.BSY + g.K - g.s;
  g.magic_x = (unsigned)((0x100000000ull + g.BSX - 1) / g.BSX);
  g.magic_y = (unsigned)((0x100000000ull + g.BSY - 1) / g.BSY);
  return g;
}
static_assert(ys::SW % 4 == 0, "slab row pitch = strip width");

inline bool use_ystream(int K, int ov) {
#ifdef TDK_EXPERIMENTS
  if (const char* e = getenv("TDK_WIENER_YSTREAM")) return atoi(e) != 0 && K == 32 && ov == 4;
#endif
  return K == 32 && ov == 4;
}

// Slab floats the workspace holds per plane: the worst case of whichever tile kernel serves (K, ov).
size_t slab_cap_floats(int W, int H, int K, int ov) {
  size_t cap = slab_floats(geometry(W, H, K, ov, 0));
  if (K == 32 && ov == 4) {
    const size_t c2 = slab_floats(geometry_ys(W, H, YS_TR_MIN));
    if (c2 > cap) cap = c2;
  }
  return cap;
}

// Tile rows per strip segment: the launch is one workgroup per (strip, segment) with 2 resident per CU; a segment of TR
// tile rows costs TR full steps plus about 4 steps' worth of pipeline fill and drain.  Fewest rounds x steps wins.
int pick_segment_rows(int W, int H, int nplanes) {
  const Geom g0 = geometry_ys(W, H, YS_TR_MIN);
  const size_t cap = slab_cap_floats(W, H, 32, 4);
  const long slots = (long)tdk_device_cus() * 2;
  int best = YS_TR_MIN;
  double best_cost = 1e30;
#ifdef TDK_EXPERIMENTS
  if (const char* e = getenv("TDK_WIENER_TR")) {
    const int tr = atoi(e);
    if (tr >= YS_TR_MIN && slab_floats(geometry_ys(W, H, tr)) <= cap) return tr;
  }
#endif
  for (int tr = YS_TR_MIN; tr <= (g0.nty > YS_TR_MIN ? g0.nty : YS_TR_MIN); tr++) {
    const Geom g = geometry_ys(W, H, tr);
    if (slab_floats(g) > cap) continue;
    const long groups = (long)g.ngx * g.ngy * nplanes;
    const long rounds = (groups + slots - 1) / slots;
    const double cost = (double)rounds * (tr + 4.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = tr; }
  }
  return best;
}

ys::YParams make_yparams() {
  float wf[32];
  make_window(32, 0.3, wf);
  ys::YParams yp = {};
  for (int kx = 0; kx < 32; kx++) {
    double re = 0.0, im = 0.0;
    for (int k = 0; k < 32; k++) {
      const double a = -2.0 * M_PI * (double)((k * kx) & 31) / 32.0;
      re += (double)wf[k] * cos(a);
      im += (double)wf[k] * sin(a);
    }
    yp.whr[kx] = (float)re;
    yp.whi[kx] = (float)im;
  }
  return yp;
}

template <typename T>
int launch_tiles_ys(const T* in, float* slabs, int W, int H, int C, int c, const float* sigmas, const Geom& g, hipStream_t st_, int nplanes) {
  const int vec_ok = W % 4 == 0 && (C == 1 ? tdk_aligned(in, 4 * sizeof(T)) : (C == 3 && tdk_aligned(in, 16)));
  static const ys::YParams yp = make_yparams();
  size_t lds_pad = 0;
#ifdef TDK_EXPERIMENTS
  if (const char* e = getenv("TDK_WIENER_LDS_PAD")) {  // one resident workgroup per CU (co-residency experiment)
    lds_pad = (size_t)atoi(e);
    const int rcl = tdk_raise_lds_limit(reinterpret_cast<const void*>(&ys::wiener_ystream<T, false>), 160 * 1024 - (int)sizeof(ys::Smem), "tdk_wiener(hipFuncSetAttribute)");
    if (rcl != TDK_OK) return rcl;
  }
#endif
  TDK_LAUNCH("tdk_wiener(tiles)", (ys::wiener_ystream<T, false>), dim3((unsigned)(g.ngx * g.ngy * nplanes)), dim3(256), lds_pad, st_, in, slabs, W, H, C, c, vec_ok, g, sigmas,
             yp, C == 1 ? (size_t)W * H : (size_t)0, 0.0f);
  return TDK_OK;
}

// The tile kernel on the log-lightness of an interleaved RGB image, extracted on the fly (no plane in HBM).
template <typename T>
int launch_tiles_ys_lum(const T* rgb, float* slabs, int W, int H, const float* sigma, float eps, const Geom& g, hipStream_t st_) {
  const int vec_ok = W % 4 == 0 && tdk_aligned(rgb, 16);
  static const ys::YParams yp = make_yparams();
  TDK_LAUNCH("tdk_wiener(tiles)", (ys::wiener_ystream<T, true>), dim3((unsigned)(g.ngx * g.ngy)), dim3(256), 0, st_, rgb, slabs, W, H, 3, 0, vec_ok, g, sigma, yp, (size_t)0,
             eps);
  return TDK_OK;
}

// Group width: the launch is one workgroup per group with `per_cu` workgroups resident per CU, so the run
// time is ceil(groups / slots) rounds of one group's duration -- pick the width whose last round is
// fullest (ties: wider groups = fewer seam columns), never using more slab than the workspace (G_MIN) holds.
int pick_group_width(int W, int H, int K, int ov, int nplanes, int per_cu) {
  const int slots = tdk_device_cus() * per_cu;
  const size_t cap = slab_floats(geometry(W, H, K, ov, 0));
#ifdef TDK_EXPERIMENTS
  if (const char* e = getenv("TDK_WIENER_G")) {
    const int G = atoi(e);
    if (G >= G_MIN && G <= G_MAX && (G & 1) == 0 && slab_floats(geometry(W, H, K, ov, G)) <= cap) return G;
  }
#endif
  int best = G_MIN;
  double best_eff = -1.0;
  for (int G = G_MIN; G <= G_MAX; G += 2) {
    const Geom g = geometry(W, H, K, ov, G);
    if (slab_floats(g) > cap) continue;
    const long groups = (long)g.ngx * g.ngy * nplanes;
    const long rounds = (groups + slots - 1) / slots;
    // work actually done per round-slot: a group's duration grows with its steps (+ flush rounds)
    const double steps = G / 2 + (K - g.s + 2 * g.s - 1) / (2 * g.s);
    const double useful = (double)g.ntx / 2 * g.ngy * nplanes;  // tile-pair steps that carry data
    const double eff = useful / ((double)rounds * slots * steps);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = G; }
  }
  return best;
}

// The tile kernels use the window as compile-time literals: true iff the table still matches make_window().
template <int K> bool window_table_ok() {
  static const bool ok = [] {
    float w[32];
    make_window(K, 0.3, w);
    for (int i = 0; i < K; i++)
      if (w[i] != WindowK<K>::w[i]) return false;
    return true;
  }();
  return ok;
}

template <int K> WParams make_params(const Geom& g, int ov) {
  WParams prm = {};
  make_window(K, 0.3, prm.wf);
  make_window(K, 0.3, prm.wi);
  for (int r = 0; r < g.s; r++) {
    float m = 0.0f;
    for (int k = 0; k < ov; k++) m += prm.wf[r + k * g.s] * prm.wi[r + k * g.s];
    prm.m1[r] = m;
  }
  return prm;
}

template <typename T, int K, int OV>
int launch_tiles_ov(const T* in, float* slabs, int W, int H, int C, int c, const float* sigmas, const Geom& g, const WParams& prm, hipStream_t st_, int nplanes) {
  // vector row loads of 4 samples (planar: 16 B fp32 / 8 B fp16 per access) or 4 interleaved RGB pixels: rows and tile
  // origins must start on a 4-sample boundary
  const int vec_ok = W % 4 == 0 && g.s % 4 == 0 && (C == 1 ? tdk_aligned(in, 4 * sizeof(T)) : (C == 3 && tdk_aligned(in, 16)));
  TDK_LAUNCH("tdk_wiener(tiles)", (wiener_stream<T, K, OV>), dim3((unsigned)(g.ngx * g.ngy * nplanes)), dim3(64 * NWV), 0, st_, in, slabs, W, H, C, c, vec_ok, g,
             sigmas, prm, nplanes, C == 1 ? (size_t)W * H : (size_t)0);
  return TDK_OK;
}

template <typename T, int K>
int launch_tiles(const T* in, float* slabs, int W, int H, int C, int c, int ov, const float* sigmas, const Geom& g, const WParams& prm, hipStream_t st_,
                 int nplanes = 1) {
  switch (ov) {
    case 2: return launch_tiles_ov<T, K, 2>(in, slabs, W, H, C, c, sigmas, g, prm, st_, nplanes);
    case 4: return launch_tiles_ov<T, K, 4>(in, slabs, W, H, C, c, sigmas, g, prm, st_, nplanes);
    default: return launch_tiles_ov<T, K, 8>(in, slabs, W, H, C, c, sigmas, g, prm, st_, nplanes);
  }
}

// workgroups of the tile kernel that fit one CU (LDS: 2 buffers x TR x K x 2s floats, 1 buffer when 2s = 32; 16 waves)
int tiles_per_cu(int K, int ov) {
  const int s = K / ov, TR = NWV * (64 / K), em = 2 * s;
  const int lds = (em >= 32 ? 1 : 2) * TR * K * em * 4;
  const int by_lds = 160 * 1024 / lds, by_waves = 4 * TDK_WIENER_WAVES_PER_SIMD / NWV;
  return by_lds < by_waves ? by_lds : by_waves;
}

inline unsigned stream_blocks(int64_t npix) {
  int64_t b = tdk_div_up64(npix, 256);
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

template <typename T, int K>
int launch(const void* in, void* out, void* workspace, int W, int H, int C, int ov, const float* sigmas, hipStream_t st_) {
  TDK_REQUIRE(window_table_ok<K>(), "tdk_wiener: the compiled-in window table does not match make_window()");
  const bool ysk = use_ystream(K, ov);
  const Geom g = ysk ? geometry_ys(W, H, pick_segment_rows(W, H, C == 3 ? 3 : 1)) : geometry(W, H, K, ov, pick_group_width(W, H, K, ov, C == 3 ? 3 : 1, tiles_per_cu(K, ov)));
  const WParams prm = make_params<K>(g, ov);
  float* slabs = reinterpret_cast<float*>(workspace);
  if (C == 3) {
    // one tile launch over 3 x groups (group set p = channel p, read straight from the interleaved image) -> one
    // finish that writes whole RGB pixels
    const bool vec = (W % 4) == 0 && tdk_aligned(in, 16) && tdk_aligned(out, 16);
    const int rc = ysk ? launch_tiles_ys<T>(reinterpret_cast<const T*>(in), slabs, W, H, 3, 0, sigmas, g, st_, 3)
                       : launch_tiles<T, K>(reinterpret_cast<const T*>(in), slabs, W, H, 3, 0, ov, sigmas, g, prm, st_, 3);
    if (rc != TDK_OK) return rc;
    const dim3 fgrid((unsigned)tdk_div_up(vec ? W / 4 : W, 256), (unsigned)(H < 32768 ? H : 32768));
    if (vec) TDK_LAUNCH("tdk_wiener(finish)", (wiener_finish3<T, 4>), fgrid, dim3(256), 0, st_, slabs, reinterpret_cast<T*>(out), W, H, g, prm);
    else TDK_LAUNCH("tdk_wiener(finish)", (wiener_finish3<T, 1>), fgrid, dim3(256), 0, st_, slabs, reinterpret_cast<T*>(out), W, H, g, prm);
    return TDK_OK;
  }
  for (int c = 0; c < C; c++) {
    const int rc = ysk ? launch_tiles_ys<T>(reinterpret_cast<const T*>(in), slabs, W, H, C, c, sigmas, g, st_, 1)
                       : launch_tiles<T, K>(reinterpret_cast<const T*>(in), slabs, W, H, C, c, ov, sigmas, g, prm, st_);
    if (rc != TDK_OK) return rc;
    TDK_LAUNCH("tdk_wiener(finish)", wiener_finish<T>, dim3((unsigned)tdk_div_up(W, 256), (unsigned)(H < 32768 ? H : 32768)), dim3(256), 0, st_, slabs,
               reinterpret_cast<T*>(out), W, H, C, c, g, prm);
  }
  return TDK_OK;
}

// Wiener.process_log_luminance as one call: extract log-L (fp32 plane in the workspace) -> tiles -> fused fold + modify.
template <typename T, int K>
int launch_log_luminance(const void* rgb_in, void* rgb_out, void* workspace, int W, int H, int ov, const float* sigma, float eps, int dtype, hipStream_t st_,
                         float* lum_out = nullptr, int lum_log = 0, float lum_eps = 1e-6f) {
  TDK_REQUIRE(window_table_ok<K>(), "tdk_wiener: the compiled-in window table does not match make_window()");
  const bool ysk = use_ystream(K, ov);
  const Geom g = ysk ? geometry_ys(W, H, pick_segment_rows(W, H, 1)) : geometry(W, H, K, ov, pick_group_width(W, H, K, ov, 1, tiles_per_cu(K, ov)));
  const WParams prm = make_params<K>(g, ov);
  float* slabs = reinterpret_cast<float*>(workspace);
  float* plane = slabs + tdk_align_up(slab_cap_floats(W, H, K, ov), 64);
  int rc;
#ifdef TDK_EXPERIMENTS
  bool fused_lum = false;
  // log-lightness extracted inside the tile kernel: measured SLOWER than the streaming extraction kernel + plane (the
  // conversion's ~9 transcendentals per pixel cost the issue-bound tile kernel 53 us; the HBM-bound extraction kernel 22 us)
  fused_lum = ysk && getenv("TDK_WIENER_FUSED_LUM") != nullptr;
#endif
#ifdef TDK_EXPERIMENTS
  if (fused_lum) {
    rc = launch_tiles_ys_lum<T>(reinterpret_cast<const T*>(rgb_in), slabs, W, H, sigma, eps, g, st_);
  } else
#endif
  {
    rc = tdk_compute_luminance(rgb_in, plane, (int64_t)W * H, 1, eps, dtype, TDK_F32, reinterpret_cast<tdk_stream_t>(st_));
    if (rc != TDK_OK) return rc;
    rc = ysk ? launch_tiles_ys<float>(plane, slabs, W, H, 1, 0, sigma, g, st_, 1) : launch_tiles<float, K>(plane, slabs, W, H, 1, 0, ov, sigma, g, prm, st_);
  }
  if (rc != TDK_OK) return rc;
  if ((W % 4) == 0 && tdk_aligned(rgb_in, 16) && tdk_aligned(rgb_out, 16) && (!lum_out || tdk_aligned(lum_out, 16)))
    TDK_LAUNCH("tdk_wiener(finish+modify)", (wiener_finish_modify<T, 4>), dim3((unsigned)tdk_div_up(W / 4, 256), (unsigned)(H < 32768 ? H : 32768)), dim3(256), 0, st_, slabs,
               reinterpret_cast<const T*>(rgb_in), reinterpret_cast<T*>(rgb_out), W, H, g, prm, lum_out, lum_log, lum_eps);
  else
    TDK_LAUNCH("tdk_wiener(finish+modify)", (wiener_finish_modify<T, 1>), dim3((unsigned)tdk_div_up(W, 256), (unsigned)(H < 32768 ? H : 32768)), dim3(256), 0, st_, slabs,
               reinterpret_cast<const T*>(rgb_in), reinterpret_cast<T*>(rgb_out), W, H, g, prm, lum_out, lum_log, lum_eps);
  return TDK_OK;
}

// The Lab hand-over form of Wiener.process_log_luminance: extract log-L + (a, b) -> tiles -> wiener_finish_lab.
template <int K>
int launch_log_luminance_lab(const void* rgb_in, void* workspace, int W, int H, int ov, const float* sigma, float eps, int dtype, hipStream_t st_, float* lum_out,
                             float* ab_out, const float* bounds) {
  TDK_REQUIRE(window_table_ok<K>(), "tdk_wiener: the compiled-in window table does not match make_window()");
  const bool ysk = use_ystream(K, ov);
  const Geom g = ysk ? geometry_ys(W, H, pick_segment_rows(W, H, 1)) : geometry(W, H, K, ov, pick_group_width(W, H, K, ov, 1, tiles_per_cu(K, ov)));
  const WParams prm = make_params<K>(g, ov);
  float* slabs = reinterpret_cast<float*>(workspace);
  float* plane = slabs + tdk_align_up(slab_cap_floats(W, H, K, ov), 64);
  int rc = TDK_OK;
#ifdef TDK_EXPERIMENTS
  if (!getenv("TDK_FAKE_SKIP_EXTRACT"))  // timing experiment (profiles/rcd_lab_fusion_exp.py): the producer is assumed to have filled the planes
#endif
  rc = tdk_compute_log_luminance_lab(rgb_in, plane, ab_out, (int64_t)W * H, eps, bounds, dtype, reinterpret_cast<tdk_stream_t>(st_));
  if (rc != TDK_OK) return rc;
  rc = ysk ? launch_tiles_ys<float>(plane, slabs, W, H, 1, 0, sigma, g, st_, 1) : launch_tiles<float, K>(plane, slabs, W, H, 1, 0, ov, sigma, g, prm, st_);
  if (rc != TDK_OK) return rc;
  const dim3 fgrid((unsigned)tdk_div_up((W % 4) == 0 ? W / 4 : W, 256), (unsigned)(H < 32768 ? H : 32768));
  if ((W % 4) == 0 && tdk_aligned(ab_out, 16) && tdk_aligned(lum_out, 16))
    TDK_LAUNCH("tdk_wiener(finish+lab)", (wiener_finish_lab<4>), fgrid, dim3(256), 0, st_, slabs, ab_out, lum_out, W, H, g, prm);
  else
    TDK_LAUNCH("tdk_wiener(finish+lab)", (wiener_finish_lab<1>), dim3((unsigned)tdk_div_up(W, 256), fgrid.y), dim3(256), 0, st_, slabs, ab_out, lum_out, W, H, g, prm);
  return TDK_OK;
}

}  // namespace

#if defined(TDK_EXPERIMENTS) && defined(TDK_YS_TIMING)
TDK_EXPORT int tdk_debug_ys_wg_times(unsigned long long* out4096) {
  return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(ys::g_ys_wg_times), sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : -1;
}
TDK_EXPORT int tdk_debug_ys_phase_cycles(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(ys::g_ys_phase_cycles), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(ys::g_ys_phase_cycles), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif

TDK_EXPORT size_t tdk_wiener_workspace_bytes(int width, int height, int channels, int tile_size, int overlap_factor) {
  if (width <= 0 || height <= 0 || !(tile_size == 16 || tile_size == 32) || !(overlap_factor == 2 || overlap_factor == 4 || overlap_factor == 8)) return 0;
  const size_t slabs = slab_cap_floats(width, height, tile_size, overlap_factor);
  return tdk_align_up((size_t)(channels == 3 ? 3 : 1) * slabs * sizeof(float), 256);  // one slab set per channel
}

TDK_EXPORT int tdk_wiener(const void* in, void* out, void* workspace, int width, int height, int channels, int tile_size, int overlap_factor,
                          const float* sigmas, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(in && out && workspace && sigmas, "tdk_wiener: null pointer");
  TDK_REQUIRE(channels == 1 || channels == 3, "input channels must be 1 or 3, got %d", channels);
  TDK_REQUIRE(tile_size == 16 || tile_size == 32, "tile_size must be 16 or 32, got %d", tile_size);
  TDK_REQUIRE(overlap_factor == 2 || overlap_factor == 4 || overlap_factor == 8, "overlap_factor must be 2, 4, or 8");
  TDK_REQUIRE(width >= tile_size && height >= tile_size, "tdk_wiener: image %dx%d smaller than the tile size %d (reflect padding undefined)",
              width, height, tile_size);
  hipStream_t s = tdk_stream(stream);
  if (tile_size == 16) TDK_DISPATCH_DTYPE(dtype, T, return (launch<T, 16>(in, out, workspace, width, height, channels, overlap_factor, sigmas, s)));
  TDK_DISPATCH_DTYPE(dtype, T, return (launch<T, 32>(in, out, workspace, width, height, channels, overlap_factor, sigmas, s)));
  return TDK_OK;
}

TDK_EXPORT size_t tdk_wiener_log_luminance_workspace_bytes(int width, int height, int tile_size, int overlap_factor) {
  if (tdk_wiener_workspace_bytes(width, height, 1, tile_size, overlap_factor) == 0) return 0;
  const size_t slabs = slab_cap_floats(width, height, tile_size, overlap_factor);
  return tdk_align_up((tdk_align_up(slabs, 64) + (size_t)width * height) * sizeof(float), 256);
}

TDK_EXPORT int tdk_wiener_log_luminance(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, int tile_size, int overlap_factor,
                                        const float* sigma, float eps, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(rgb_in && rgb_out && workspace && sigma, "tdk_wiener_log_luminance: null pointer");
  TDK_REQUIRE(tile_size == 16 || tile_size == 32, "tile_size must be 16 or 32, got %d", tile_size);
  TDK_REQUIRE(overlap_factor == 2 || overlap_factor == 4 || overlap_factor == 8, "overlap_factor must be 2, 4, or 8");
  TDK_REQUIRE(width >= tile_size && height >= tile_size, "tdk_wiener_log_luminance: image %dx%d smaller than the tile size %d", width, height, tile_size);
  TDK_REQUIRE(eps > 0.0f, "Epsilon must be positive");
  hipStream_t s = tdk_stream(stream);
  if (tile_size == 16) TDK_DISPATCH_DTYPE(dtype, T, return (launch_log_luminance<T, 16>(rgb_in, rgb_out, workspace, width, height, overlap_factor, sigma, eps, dtype, s)));
  TDK_DISPATCH_DTYPE(dtype, T, return (launch_log_luminance<T, 32>(rgb_in, rgb_out, workspace, width, height, overlap_factor, sigma, eps, dtype, s)));
  return TDK_OK;
}

TDK_EXPORT int tdk_wiener_log_luminance_lum(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, int tile_size, int overlap_factor,
                                            const float* sigma, float eps, int dtype, float* lum_out, int lum_log_mode, float lum_eps, tdk_stream_t stream) {
  TDK_REQUIRE(rgb_in && rgb_out && workspace && sigma && lum_out, "tdk_wiener_log_luminance_lum: null pointer");
  TDK_REQUIRE(tile_size == 16 || tile_size == 32, "tile_size must be 16 or 32, got %d", tile_size);
  TDK_REQUIRE(overlap_factor == 2 || overlap_factor == 4 || overlap_factor == 8, "overlap_factor must be 2, 4, or 8");
  TDK_REQUIRE(width >= tile_size && height >= tile_size, "tdk_wiener_log_luminance_lum: image %dx%d smaller than the tile size %d", width, height, tile_size);
  TDK_REQUIRE(eps > 0.0f && (!lum_log_mode || lum_eps > 0.0f), "Epsilon must be positive");
  hipStream_t s = tdk_stream(stream);
  if (tile_size == 16)
    TDK_DISPATCH_DTYPE(dtype, T, return (launch_log_luminance<T, 16>(rgb_in, rgb_out, workspace, width, height, overlap_factor, sigma, eps, dtype, s, lum_out, lum_log_mode, lum_eps)));
  TDK_DISPATCH_DTYPE(dtype, T, return (launch_log_luminance<T, 32>(rgb_in, rgb_out, workspace, width, height, overlap_factor, sigma, eps, dtype, s, lum_out, lum_log_mode, lum_eps)));
  return TDK_OK;
}

TDK_EXPORT int tdk_wiener_log_luminance_lab(const void* rgb_in, void* workspace, int width, int height, int tile_size, int overlap_factor, const float* sigma,
                                            float eps, const float* bounds, int dtype, float* lum_out, float* ab_out, tdk_stream_t stream) {
  TDK_REQUIRE(rgb_in && workspace && sigma && lum_out && ab_out, "tdk_wiener_log_luminance_lab: null pointer");
  TDK_REQUIRE(tile_size == 16 || tile_size == 32, "tile_size must be 16 or 32, got %d", tile_size);
  TDK_REQUIRE(overlap_factor == 2 || overlap_factor == 4 || overlap_factor == 8, "overlap_factor must be 2, 4, or 8");
  TDK_REQUIRE(width >= tile_size && height >= tile_size, "tdk_wiener_log_luminance_lab: image %dx%d smaller than the tile size %d", width, height, tile_size);
  TDK_REQUIRE(eps > 0.0f, "Epsilon must be positive");
  TDK_REQUIRE(dtype == TDK_F32 || dtype == TDK_F16, "unsupported dtype tag %d", dtype);
  hipStream_t s = tdk_stream(stream);
  if (tile_size == 16) return launch_log_luminance_lab<16>(rgb_in, workspace, width, height, overlap_factor, sigma, eps, dtype, s, lum_out, ab_out, bounds);
  return launch_log_luminance_lab<32>(rgb_in, workspace, width, height, overlap_factor, sigma, eps, dtype, s, lum_out, ab_out, bounds);
}
