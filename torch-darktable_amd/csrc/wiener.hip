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
// MI355X design (memory-side float atomics cap at ~1.3 TB/s, so the reference's ~32 global
// atomics per pixel cannot be kept; the op is FP32-vector bound: ~2.5 kFLOP/px at ov = 4):
//  * a wave processes 64/K tiles at once, one tile ROW per lane: the K-point FFTs run entirely
//    in registers (fully unrolled radix-2 DIT, immediate twiddles, no cross-lane traffic); the
//    two transposes per direction go through a per-wave padded LDS tile;
//  * a 512-thread workgroup owns 8 tile rows x (64/K * ov) tile columns (64 x 8s output pixels);
//    each wave overlap-adds its tile row into a private LDS accumulator with plain 16-B
//    read-modify-writes -- no atomics anywhere (LDS float atomics are ~150 cycles each);
//  * each workgroup writes its (64 - s + K) x (7s + K) partial slab once; a second streaming
//    kernel sums the <= 4 slabs that overlap an output pixel in a fixed order and applies the mask.  The
//    mask is input-independent and separable (every pixel is covered by exactly ov x ov tiles):
//    mask(x, y) = m1[x mod s] * m1[y mod s], m1[r] = sum_k wf[r + k s] * wi[r + k s], so it is
//    never accumulated.
//  Windows are evaluated on the host in fp64 and rounded once to fp32 (the reference uses torch
//  fp32 ops on the GPU; both are within an ulp of each other).  This file allows FMA
//  contraction: the reference's sums are order-nondeterministic, parity is by tolerance.
#include <math.h>
#include <string.h>

#include "tdk_color.h"

#pragma clang fp contract(fast)

namespace {

constexpr int BS = 64;  // output pixels per workgroup edge = GT * s

struct WParams {
  float wf[32];  // analysis (FFT) window
  float wi[32];  // synthesis (interpolation) window
  float m1[16];  // separable mask factor, index p mod s
};

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

__device__ __forceinline__ int reflect_index(int x, int limit) {
  if (x < 0) x = -x;
  if (x >= limit) x = 2 * limit - x - 1;
  return x;
}

// Transpose one K x K tile held one row per lane through a padded per-wave LDS buffer.  The
// buffer belongs to one wave (LDS operations of a wave are processed in issue order), so no
// workgroup barrier is needed -- only a fence that keeps the compiler from reordering.
template <int K> __device__ __forceinline__ void transpose_tile(float (&v)[K], float* buf, int row) {
#pragma unroll
  for (int k = 0; k < K; k++) buf[row * (K + 1) + k] = v[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = buf[k * (K + 1) + row];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Load K consecutive samples of one tile row (channel `chan` of an HWC image).  Tiles that lie
// inside the image along x read contiguous memory (16-B vector loads when aligned); only edge
// tiles pay for the per-sample reflect.
template <typename T, int K>
__device__ __forceinline__ void load_row(const T* __restrict__ row_ptr, int ox, int W, int C, int chan, float (&v)[K]) {
  if (C == 1 && ox >= 0 && ox + K <= W) {
    const T* p = row_ptr + ox;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < K; k += 4) {
        float t[4];
        s4_io<T>::load(p, k / 4, t);
        v[k] = t[0]; v[k + 1] = t[1]; v[k + 2] = t[2]; v[k + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = ld(p, k);
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = ld(row_ptr, (size_t)reflect_index(ox + k, W) * C + chan);
  }
}

// dst[k] += (v[k] + mean * wf[k] wf[ty]) * (wi[k] wi[ty])  (reference denoise.cu:172-175), written as
// wi[k] * (v[k] * wi[ty] + (mean * wf[ty] wi[ty]) * wf[k]) so that every per-column factor is a scalar
// (SGPR) operand instead of 64 per-lane registers: 3 instructions per sample.  16-B
// read-modify-writes when the column offset allows.
template <int K>
__device__ __forceinline__ void accumulate_row(float* dst, const float (&v)[K], float mean, float wy, float iy, const WParams& prm, bool vec) {
  const float mw = mean * (wy * iy);
  auto term = [&](int k, float a) { return __builtin_fmaf(prm.wi[k], __builtin_fmaf(mw, prm.wf[k], v[k] * iy), a); };
  if (vec) {
#pragma unroll
    for (int k = 0; k < K; k += 4) {
      float4 a = *reinterpret_cast<float4*>(dst + k);
      a.x = term(k, a.x); a.y = term(k + 1, a.y); a.z = term(k + 2, a.z); a.w = term(k + 3, a.w);
      *reinterpret_cast<float4*>(dst + k) = a;
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; k++) dst[k] = term(k, dst[k]);
  }
}

constexpr int NW = 8;  // waves per workgroup == tile rows per group

// LDS row stride of a per-wave accumulator: >= n, a multiple of 4 floats with (stride / 4) odd,
// so the 16-B accesses of 16 lanes on consecutive rows fall in 16 different 4-bank slots.
__host__ __device__ inline int acc_stride(int n) {
  int st = (n + 3) & ~3;
  if (((st >> 2) & 1) == 0) st += 4;
  return st;
}

// One workgroup (8 waves) = one group of 8 tile rows x (TPW * ov) tile columns of one channel,
// i.e. a 64 (x) by 8 s (y) block of output pixels.  Wave w owns tile row w and overlap-adds its
// tiles into a PRIVATE K-row LDS accumulator with plain 16-B read-modify-writes: the tiles a
// wave handles at the same time are ov columns apart (they do not overlap) and consecutive tiles
// are sequential, so no atomics are needed (ds_add_f32 measured ~150 cycles per wave
// instruction on gfx950 -- it dominated the first version of this kernel).  The eight private
// accumulators are folded in a fixed order when the group's slab is written: the whole op is
// deterministic, unlike the reference's atomic overlap-add.
template <typename T, int K>
__global__ __launch_bounds__(64 * NW) void wiener_tiles(const T* __restrict__ img, float* __restrict__ slabs, int W, int H, int C, int chan,
                                                        int s, int ov, int jmin, int ntile_x, int ntile_y, int ngx, int ngroups,
                                                        const float* __restrict__ sigmas, WParams prm, int nplanes, size_t plane_stride) {
  constexpr int TPW = 64 / K;  // tile pairs (slots) per wave
  extern __shared__ __align__(16) float lds[];
  const int GTX = TPW * ov;
  const int RSX = BS - s + K, RSY = (NW - 1) * s + K;
  const int AST = acc_stride(RSX);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per_wave = K * AST + TPW * K * (K + 1);
  float* acc = lds + wave * per_wave;  // K rows of AST
  float* tbuf = acc + K * AST;         // per-wave transpose scratch
  const int row = lane & (K - 1), slot = lane / K;
  float* my_t = tbuf + slot * (K * (K + 1));
  // nplanes > 1: `img` holds that many separate planes (C == 1 each, plane_stride samples apart), group
  // index = plane * ngroups + group, noise sigma = sigmas[chan + plane]
  const int total_groups = ngroups * nplanes;
  const int partner = (lane & ~(K - 1)) | ((K - row) & (K - 1));  // lane holding column -kx of the same slot
  const float wy = prm.wf[row], iy = prm.wi[row];
  const int nsteps = ov >> 1;

  // Work item = (group, step).  Persistent workgroups (one per CU: the LDS footprint allows no
  // more) walk the groups with a grid stride.  (Fetching the rows of the NEXT item into registers
  // during the transforms was measured slower: 253 VGPRs, 0.58 vs 0.29 ms.)
  struct Item {
    int txa, txb, oxa, oxb;
    bool act_a, act_b;
    const T* src_row;
  };
  // (plane, gx, gy) of a group are computed once per group (three integer divisions), not per step
  auto make_item = [&](int plane, int gx, int gy, int base) {
    Item it;
    const int jx0 = jmin + gx * GTX, jy = jmin + gy * NW + wave;
    const bool row_active = jy < jmin + ntile_y;
    it.txa = slot * ov + 2 * base;
    it.txb = it.txa + 1;
    it.act_a = row_active && (jx0 + it.txa < jmin + ntile_x);
    it.act_b = row_active && (jx0 + it.txb < jmin + ntile_x);
    it.oxa = (jx0 + it.txa) * s;
    it.oxb = (jx0 + it.txb) * s;
    it.src_row = img + (size_t)plane * plane_stride + (size_t)reflect_index(jy * s + row, H) * W * C;
    return it;
  };
  auto fetch = [&](const Item& it, float (&ra)[K], float (&rb)[K]) {
    if (it.act_a) {
      load_row<T, K>(it.src_row, it.oxa, W, C, chan, ra);
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) ra[k] = 0.0f;
    }
    if (it.act_b) {
      load_row<T, K>(it.src_row, it.oxb, W, C, chan, rb);
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) rb[k] = 0.0f;
    }
  };


  for (int grp = blockIdx.x; grp < total_groups; grp += gridDim.x) {
    const int plane = grp / ngroups, grp_in_plane = grp - plane * ngroups;
    const int ggy = grp_in_plane / ngx, ggx = grp_in_plane - ggy * ngx;
    const float sigma = sigmas[chan + plane];
    const float sig2 = sigma * sigma;
    for (int i = lane; i < K * AST / 4; i += 64) reinterpret_cast<float4*>(acc)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

    // Two real tiles ride through ONE complex 2-D FFT: z = a + i b.  After the forward transform
    // the spectra are separated with the Hermitian identities A[k] = (Z[k] + conj(Z[-k])) / 2,
    // B[k] = (Z[k] - conj(Z[-k])) / 2i (Z[-k] sits in lane -kx, register -ky), each gets its own
    // Wiener gain, and Z' = A' + i B' goes back through one inverse transform: re = a', im = b'.
    // Tile a / b of slot i in step `base`: columns i*ov + 2*base (+1).  The a (b) tiles of
    // different slots are ov columns apart, and a and b are accumulated one after the other, so
    // the plain read-modify-writes below never collide.
    for (int base = 0; base < nsteps; base++) {
      float re[K], im[K];
      const Item it = make_item(plane, ggx, ggy, base);
      fetch(it, re, im);

      float mean_a, mean_b;
      {
        float sa = 0.0f, sb = 0.0f;
#pragma unroll
        for (int k = 0; k < K; k++) { sa += re[k]; sb += im[k]; }
#pragma unroll
        for (int o = K / 2; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
        mean_a = sa / (float)(K * K);
        mean_b = sb / (float)(K * K);
        const float ca = -mean_a * wy, cb = -mean_b * wy;
#pragma unroll
        for (int k = 0; k < K; k++) {  // (x - mean) * wf[ty] * wf[tx]; the per-column factor stays a scalar operand
          re[k] = __builtin_fmaf(re[k], wy, ca) * prm.wf[k];
          im[k] = __builtin_fmaf(im[k], wy, cb) * prm.wf[k];
        }
      }

      fft_inreg<K, false>(re, im);            // along x
      transpose_tile<K>(re, my_t, row);
      transpose_tile<K>(im, my_t, row);
      fft_inreg<K, false>(re, im);            // along y (lane = kx)

      // Separate the two spectra, apply the gains (denoise.cu:181-185), recombine.  With
      // 2A = Z[k] + conj(Z[-k]) and 2B = -i (Z[k] - conj(Z[-k])):  Z'[k] = ga A + i gb B and, because A
      // and B are spectra of real tiles, Z'[-k] = ga conj(A) + i gb conj(B) -- the same gains and
      // products.  Z[-k] lives in the partner lane (column -kx) at register -ky, and the partner
      // needs exactly the mirrored pair, so every lane evaluates only index ky = k (k = 0..K/2 and
      // its by-product for the partner's register K-k) and the two lanes swap by-products: half the
      // gain arithmetic of evaluating every bin.  The 1/2 of A, B and the 1/K^2 of the two
      // unscaled inverse passes are powers of two folded into the gain (exact).
      // gain = max(|A|^2 + eps - sigma^2, 0) / (|A|^2 + eps) = max(1 - sigma^2 / p, 0) with p = |A|^2 + eps and
      // |A|^2 = |2A|^2 / 4: evaluated on p4 = |2A|^2 + 4 eps as max(GS - (4 sigma^2 GS) / p4, 0), GS = the folded
      // 1/2 * 1/K^2 -- two FMAs for p4, then rcp + FMA + max.
      constexpr float GSCALE = 0.5f / (float)(K * K);
      const float sgs = -4.0f * sig2 * GSCALE;
#pragma unroll
      for (int k = 0; k <= K / 2; k++) {
        const int k2 = (K - k) & (K - 1);
        const float zr = re[k], zi = im[k];
        const float pr = __shfl(re[k2], partner, 64), pi = __shfl(im[k2], partner, 64);  // Z[-k]
        const float a2r = zr + pr, a2i = zi - pi, b2r = zi + pi, b2i = pr - zr;
        const float pa4 = __builtin_fmaf(a2i, a2i, __builtin_fmaf(a2r, a2r, 4e-15f)), pb4 = __builtin_fmaf(b2i, b2i, __builtin_fmaf(b2r, b2r, 4e-15f));
        const float ga = fmaxf(__builtin_fmaf(sgs, __builtin_amdgcn_rcpf(pa4), GSCALE), 0.0f);
        const float gb = fmaxf(__builtin_fmaf(sgs, __builtin_amdgcn_rcpf(pb4), GSCALE), 0.0f);
        const float gar = ga * a2r, gai = ga * a2i, gbr = gb * b2r, gbi = gb * b2i;
        re[k] = gar - gbi; im[k] = gai + gbr;                 // Z'[k]
        if (k2 != k) {
          re[k2] = __shfl(gar + gbi, partner, 64);            // my Z'[K-k] is the partner's by-product
          im[k2] = __shfl(gbr - gai, partner, 64);
        }
      }

      fft_inreg<K, true>(re, im);             // inverse along y
      transpose_tile<K>(re, my_t, row);
      transpose_tile<K>(im, my_t, row);
      fft_inreg<K, true>(re, im);             // inverse along x (lane = y again): re = tile a, im = tile b

      if (it.act_a) accumulate_row<K>(acc + row * AST + it.txa * s, re, mean_a, wy, iy, prm, (s & 3) == 0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (it.act_b) accumulate_row<K>(acc + row * AST + it.txb * s, im, mean_b, wy, iy, prm, (s & 3) == 0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // fold the private accumulators: slab row r gets wave w's row r - w*s for the (at most ov)
    // waves with 0 <= r - w*s < K, summed in increasing w (s = K / ov is a power of two)
    float* slab = slabs + (size_t)grp * (size_t)(RSX * RSY);
    const int ls = 31 - __clz(s);
    if ((RSX & 3) == 0) {
      const int QX = RSX >> 2;
      const float inv_qx = 1.0f / (float)QX;
      for (int i = threadIdx.x; i < QX * RSY; i += 64 * NW) {
        const int r = (int)(((float)i + 0.5f) * inv_qx), q = i - r * QX;  // exact: i < 2^20, QX < 2^10
        const int w_hi = min(r >> ls, NW - 1), w_lo = (r >= K) ? ((r - K) >> ls) + 1 : 0;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int w = w_lo; w <= w_hi; w++) {
          const float4 a = *reinterpret_cast<const float4*>(lds + w * per_wave + (r - (w << ls)) * AST + 4 * q);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        reinterpret_cast<float4*>(slab)[i] = v;
      }
    } else {
      const float inv_rsx = 1.0f / (float)RSX;
      for (int i = threadIdx.x; i < RSX * RSY; i += 64 * NW) {
        const int r = (int)(((float)i + 0.5f) * inv_rsx), c = i - r * RSX;
        const int w_hi = min(r >> ls, NW - 1), w_lo = (r >= K) ? ((r - K) >> ls) + 1 : 0;
        float v = 0.0f;
        for (int w = w_lo; w <= w_hi; w++) v += lds[w * per_wave + (r - (w << ls)) * AST + c];
        slab[i] = v;
      }
    }
    __syncthreads();  // the accumulators are zeroed again at the top of the loop
  }
}

// Sum the overlapping slabs of one channel, normalise by the analytic mask, crop.  One workgroup
// walks whole image rows (no per-pixel index division; s and BS are powers of two).
template <typename T>
__global__ __launch_bounds__(256) void wiener_finish(const float* __restrict__ slabs, T* __restrict__ out, int W, int H, int C, int chan, int s,
                                                     int K, int jmin, int ngx, WParams prm) {
  const int RSX = BS - s + K, RSY = (NW - 1) * s + K, BSY = NW * s;
  const int u0 = -jmin * s;  // = (ov - 1) * s: pixel 0 sits at this offset inside group 0
  const size_t slab_sz = (size_t)RSX * RSY;
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const int uy = y + u0, gy = uy / BSY, offy = uy - gy * BSY;
    const bool py = (offy < K - s) && gy > 0;
    const float my = prm.m1[y & (s - 1)];
    const float* row0 = slabs + (size_t)gy * ngx * slab_sz + (size_t)offy * RSX;                   // group row gy
    const float* row1 = py ? slabs + (size_t)(gy - 1) * ngx * slab_sz + (size_t)(offy + BSY) * RSX : nullptr;  // group row gy - 1
    for (int x = blockIdx.x * 256 + threadIdx.x; x < W; x += gridDim.x * 256) {
      const int ux = x + u0, gx = ux / BS, offx = ux - gx * BS;
      const bool px = (offx < K - s) && gx > 0;
      float v = row0[gx * slab_sz + offx];
      if (px) v += row0[(gx - 1) * slab_sz + offx + BS];
      if (py) v += row1[gx * slab_sz + offx];
      if (px && py) v += row1[(gx - 1) * slab_sz + offx + BS];
      const float mask = prm.m1[x & (s - 1)] * my;
      st(out, ((size_t)y * W + x) * C + chan, v / (mask + 1e-15f));
    }
  }
}

// Fused epilogue of Wiener.process_log_luminance (reference denoise.py:54-58): the slab fold and
// normalisation of wiener_finish followed directly by modify_log_luminance on the RGB pixel, so the
// denoised log-luminance plane is never written to HBM.  The colour math must not be contracted.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void wiener_finish_modify(const float* __restrict__ slabs, const T* __restrict__ rgb, T* __restrict__ out, int W, int H,
                                                            int s, int K, int jmin, int ngx, WParams prm) {
#pragma clang fp contract(off)
  const int RSX = BS - s + K, RSY = (NW - 1) * s + K, BSY = NW * s;
  const int u0 = -jmin * s;
  const size_t slab_sz = (size_t)RSX * RSY;
  const int ngroup = W / VEC;  // VEC == 4 requires W % 4 == 0: a group never straddles a row
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const int uy = y + u0, gy = uy / BSY, offy = uy - gy * BSY;
    const bool py = (offy < K - s) && gy > 0;
    const float my = prm.m1[y & (s - 1)];
    const float* row0 = slabs + (size_t)gy * ngx * slab_sz + (size_t)offy * RSX;
    const float* row1 = py ? slabs + (size_t)(gy - 1) * ngx * slab_sz + (size_t)(offy + BSY) * RSX : nullptr;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ngroup; g += gridDim.x * 256) {
      const int x0 = g * VEC;
      const size_t gi = (size_t)y * ngroup + g;
      float v[3 * VEC];
      if constexpr (VEC == 4) rgb4_io<T>::load(rgb, gi, v);
      else { v[0] = ld(rgb, gi * 3); v[1] = ld(rgb, gi * 3 + 1); v[2] = ld(rgb, gi * 3 + 2); }
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        const int x = x0 + k;
        const int ux = x + u0, gx = ux / BS, offx = ux - gx * BS;
        const bool px = (offx < K - s) && gx > 0;
        float acc = row0[gx * slab_sz + offx];
        if (px) acc += row0[(gx - 1) * slab_sz + offx + BS];
        if (py) acc += row1[gx * slab_sz + offx];
        if (px && py) acc += row1[(gx - 1) * slab_sz + offx + BS];
        const float mask = prm.m1[x & (s - 1)] * my;
        const f3 r = cA::modify_log_luminance(mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2]), acc / (mask + 1e-15f));
        v[3 * k] = r.x; v[3 * k + 1] = r.y; v[3 * k + 2] = r.z;
      }
      if constexpr (VEC == 4) rgb4_io<T>::store(out, gi, v);
      else { st(out, gi * 3, v[0]); st(out, gi * 3 + 1, v[1]); st(out, gi * 3 + 2, v[2]); }
    }
  }
}

// Multi-channel input: de-interleave HWC into fp32 planes once, so that the tile kernel reads
// contiguous rows with 16-B loads (a strided channel read costs one cache line per sample).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void split_planes3(const T* __restrict__ rgb, float* __restrict__ planes, int64_t npix) {
  const int64_t ng = npix / VEC;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ng; g += (int64_t)gridDim.x * 256) {
    float v[3 * VEC];
    if constexpr (VEC == 4) rgb4_io<T>::load(rgb, (size_t)g, v);
    else { v[0] = ld(rgb, (size_t)g * 3); v[1] = ld(rgb, (size_t)g * 3 + 1); v[2] = ld(rgb, (size_t)g * 3 + 2); }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float o[VEC];
#pragma unroll
      for (int k = 0; k < VEC; k++) o[k] = v[3 * k + c];
      if constexpr (VEC == 4) s4_io<float>::store(planes + (size_t)c * npix, (size_t)g, o);
      else planes[(size_t)c * npix + g] = o[0];
    }
  }
}

// Finish for three planes at once: fold the slabs of each channel, normalise, write interleaved RGB.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void wiener_finish3(const float* __restrict__ slabs, T* __restrict__ out, int W, int H, int s, int K, int jmin, int ngx,
                                                      int ngroups, WParams prm) {
  const int RSX = BS - s + K, RSY = (NW - 1) * s + K, BSY = NW * s;
  const int u0 = -jmin * s;
  const size_t slab_sz = (size_t)RSX * RSY, chan_sz = slab_sz * ngroups;
  const int ngroup = W / VEC;
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const int uy = y + u0, gy = uy / BSY, offy = uy - gy * BSY;
    const bool py = (offy < K - s) && gy > 0;
    const float my = prm.m1[y & (s - 1)];
    const float* row0 = slabs + (size_t)gy * ngx * slab_sz + (size_t)offy * RSX;
    const float* row1 = py ? slabs + (size_t)(gy - 1) * ngx * slab_sz + (size_t)(offy + BSY) * RSX : nullptr;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ngroup; g += gridDim.x * 256) {
      const size_t gi = (size_t)y * ngroup + g;
      float v[3 * VEC];
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        const int x = g * VEC + k;
        const int ux = x + u0, gx = ux / BS, offx = ux - gx * BS;
        const bool px = (offx < K - s) && gx > 0;
        const float norm = prm.m1[x & (s - 1)] * my + 1e-15f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float* r0 = row0 + c * chan_sz;
          float acc = r0[gx * slab_sz + offx];
          if (px) acc += r0[(gx - 1) * slab_sz + offx + BS];
          if (py) acc += row1[c * chan_sz + gx * slab_sz + offx];
          if (px && py) acc += row1[c * chan_sz + (gx - 1) * slab_sz + offx + BS];
          v[3 * k + c] = acc / norm;
        }
      }
      if constexpr (VEC == 4) rgb4_io<T>::store(out, gi, v);
      else { st(out, gi * 3, v[0]); st(out, gi * 3 + 1, v[1]); st(out, gi * 3 + 2, v[2]); }
    }
  }
}

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

struct Geometry {
  int s, jmin, ntx, nty, ngx, ngy, RSX, RSY;
};

Geometry geometry(int W, int H, int K, int ov) {
  Geometry g;
  g.s = K / ov;
  g.jmin = -(ov - 1);                                  // first origin index whose tile covers pixel 0
  g.ntx = (W - 1) / g.s - g.jmin + 1;                  // origins jmin .. floor((W-1)/s)
  g.nty = (H - 1) / g.s - g.jmin + 1;
  const int GTX = (64 / K) * ov;                       // == BS / s
  g.ngx = tdk_div_up(g.ntx, GTX);
  g.ngy = tdk_div_up(g.nty, NW);
  g.RSX = BS - g.s + K;
  g.RSY = (NW - 1) * g.s + K;
  return g;
}

template <int K> WParams make_params(const Geometry& g, int ov) {
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

template <typename T, int K>
int launch_tiles(const T* in, float* slabs, int W, int H, int C, int c, int ov, const float* sigmas, const Geometry& g, const WParams& prm, hipStream_t st_,
                 int nplanes = 1) {
  constexpr int TPW = 64 / K;
  const size_t lds_bytes = (size_t)NW * ((size_t)K * acc_stride(g.RSX) + TPW * K * (K + 1)) * sizeof(float);
  TDK_HIP_CALL(hipFuncSetAttribute(reinterpret_cast<const void*>(&wiener_tiles<T, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes),
               "tdk_wiener(hipFuncSetAttribute)");
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int ngroups = g.ngx * g.ngy;
  const int blocks = ngroups * nplanes < cus ? ngroups * nplanes : cus;  // persistent: one workgroup per CU (LDS-limited), grid-stride over the groups
  TDK_LAUNCH("tdk_wiener(tiles)", (wiener_tiles<T, K>), dim3(blocks), dim3(64 * NW), lds_bytes, st_, in, slabs, W, H, C, c, g.s, ov, g.jmin, g.ntx,
             g.nty, g.ngx, ngroups, sigmas, prm, nplanes, (size_t)W * H);
  return TDK_OK;
}

inline unsigned stream_blocks(int64_t npix) {
  int64_t b = tdk_div_up64(npix, 256);
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

template <typename T, int K>
int launch(const void* in, void* out, void* workspace, int W, int H, int C, int ov, const float* sigmas, hipStream_t st_) {
  const Geometry g = geometry(W, H, K, ov);
  const WParams prm = make_params<K>(g, ov);
  float* slabs = reinterpret_cast<float*>(workspace);
  if (C == 3) {
    // de-interleave -> one tile launch over 3 x groups -> one finish that writes whole RGB pixels
    const size_t slab_floats = (size_t)g.ngx * g.ngy * g.RSX * g.RSY;
    float* planes = slabs + tdk_align_up(3 * slab_floats, 64);
    const int64_t npix = (int64_t)W * H;
    const bool vec = (W % 4) == 0 && tdk_aligned(in, 16) && tdk_aligned(out, 16);
    const T* tin = reinterpret_cast<const T*>(in);
    if (vec) TDK_LAUNCH("tdk_wiener(split)", (split_planes3<T, 4>), dim3(stream_blocks(npix / 4)), dim3(256), 0, st_, tin, planes, npix);
    else TDK_LAUNCH("tdk_wiener(split)", (split_planes3<T, 1>), dim3(stream_blocks(npix)), dim3(256), 0, st_, tin, planes, npix);
    const int rc = launch_tiles<float, K>(planes, slabs, W, H, 1, 0, ov, sigmas, g, prm, st_, 3);
    if (rc != TDK_OK) return rc;
    const dim3 fgrid((unsigned)tdk_div_up(vec ? W / 4 : W, 256), (unsigned)(H < 32768 ? H : 32768));
    if (vec) TDK_LAUNCH("tdk_wiener(finish)", (wiener_finish3<T, 4>), fgrid, dim3(256), 0, st_, slabs, reinterpret_cast<T*>(out), W, H, g.s, K, g.jmin, g.ngx, g.ngx * g.ngy, prm);
    else TDK_LAUNCH("tdk_wiener(finish)", (wiener_finish3<T, 1>), fgrid, dim3(256), 0, st_, slabs, reinterpret_cast<T*>(out), W, H, g.s, K, g.jmin, g.ngx, g.ngx * g.ngy, prm);
    return TDK_OK;
  }
  for (int c = 0; c < C; c++) {
    const int rc = launch_tiles<T, K>(reinterpret_cast<const T*>(in), slabs, W, H, C, c, ov, sigmas, g, prm, st_);
    if (rc != TDK_OK) return rc;
    TDK_LAUNCH("tdk_wiener(finish)", wiener_finish<T>, dim3((unsigned)tdk_div_up(W, 256), (unsigned)(H < 32768 ? H : 32768)), dim3(256), 0, st_, slabs, reinterpret_cast<T*>(out), W, H, C, c, g.s,
               K, g.jmin, g.ngx, prm);
  }
  return TDK_OK;
}

// Wiener.process_log_luminance as one call: extract log-L (fp32 plane in the workspace) -> tiles -> fused fold + modify.
template <typename T, int K>
int launch_log_luminance(const void* rgb_in, void* rgb_out, void* workspace, int W, int H, int ov, const float* sigma, float eps, int dtype, hipStream_t st_) {
  const Geometry g = geometry(W, H, K, ov);
  const WParams prm = make_params<K>(g, ov);
  float* slabs = reinterpret_cast<float*>(workspace);
  float* plane = slabs + tdk_align_up((size_t)g.ngx * g.ngy * g.RSX * g.RSY, 64);
  int rc = tdk_compute_luminance(rgb_in, plane, (int64_t)W * H, 1, eps, dtype, TDK_F32, reinterpret_cast<tdk_stream_t>(st_));
  if (rc != TDK_OK) return rc;
  rc = launch_tiles<float, K>(plane, slabs, W, H, 1, 0, ov, sigma, g, prm, st_);
  if (rc != TDK_OK) return rc;
  if ((W % 4) == 0 && tdk_aligned(rgb_in, 16) && tdk_aligned(rgb_out, 16))
    TDK_LAUNCH("tdk_wiener(finish+modify)", (wiener_finish_modify<T, 4>), dim3((unsigned)tdk_div_up(W / 4, 256), (unsigned)(H < 32768 ? H : 32768)), dim3(256), 0, st_, slabs,
               reinterpret_cast<const T*>(rgb_in), reinterpret_cast<T*>(rgb_out), W, H, g.s, K, g.jmin, g.ngx, prm);
  else
    TDK_LAUNCH("tdk_wiener(finish+modify)", (wiener_finish_modify<T, 1>), dim3((unsigned)tdk_div_up(W, 256), (unsigned)(H < 32768 ? H : 32768)), dim3(256), 0, st_, slabs,
               reinterpret_cast<const T*>(rgb_in), reinterpret_cast<T*>(rgb_out), W, H, g.s, K, g.jmin, g.ngx, prm);
  return TDK_OK;
}

}  // namespace

TDK_EXPORT size_t tdk_wiener_workspace_bytes(int width, int height, int channels, int tile_size, int overlap_factor) {
  if (width <= 0 || height <= 0 || !(tile_size == 16 || tile_size == 32) || !(overlap_factor == 2 || overlap_factor == 4 || overlap_factor == 8)) return 0;
  const Geometry g = geometry(width, height, tile_size, overlap_factor);
  const size_t slab_floats = (size_t)g.ngx * g.ngy * g.RSX * g.RSY;
  if (channels == 3)  // one slab set per channel + the three de-interleaved fp32 planes
    return tdk_align_up((tdk_align_up(3 * slab_floats, 64) + 3 * (size_t)width * height) * sizeof(float), 256);
  return tdk_align_up(slab_floats * sizeof(float), 256);
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
  const size_t slabs = tdk_wiener_workspace_bytes(width, height, 1, tile_size, overlap_factor);
  if (slabs == 0) return 0;
  const Geometry g = geometry(width, height, tile_size, overlap_factor);
  return tdk_align_up((tdk_align_up((size_t)g.ngx * g.ngy * g.RSX * g.RSY, 64) + (size_t)width * height) * sizeof(float), 256);
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
