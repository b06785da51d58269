// tdk_wiener_ystream.h -- the tile kernel of the Wiener denoiser for the default tile geometry K = 32, ov = 4
// (reference csrc/denoise/denoise.cu:151-242 with DenoiseParams tile_size 32, overlap_factor 4: pipeline/config.py).
// Included by wiener.hip inside its anonymous namespace (shares Geom, WindowK, wiener_gains, the finish kernels).
//
// Same arithmetic per tile as wiener_stream (mean removal, separable Gaussian analysis window, 2-D FFT, gain
// max(|X|^2 + eps - sigma^2, 0) / (|X|^2 + eps), inverse, synthesis window, overlap-add), re-associated so that the
// work shared by overlapping tiles is done once (profiles/wiener_ystream_proto.py checks the algebra in numpy):
//  * a workgroup owns a STRIP of 16 tile columns (8 tile pairs: two adjacent tiles ride one complex transform) and
//    walks DOWN the image, one tile row = s = 8 new image rows per step;
//  * forward row FFT: an 8-row block of the strip is transformed ONCE (64 lanes = 8 block rows x 8 tile pairs, one
//    32-point complex FFT of (a + i b) * wf[x] per lane, in registers) and used by the 4 tile rows that contain it:
//    the tile mean and the row factor wf[y] of the window are applied in the (row, kx) domain,
//    Z[y][kx] = wf[y] (R[y][kx] - mean W[kx]),  W = FFT(wf);
//  * column pipeline: lane = kx, register = row.  A lane keeps the 32 rows of its tile pair's current window in
//    registers (8 new rows per step from LDS), runs FFT_y -> gains -> IFFT_y in registers, and overlap-adds ACROSS
//    tile rows still in the (row, kx) domain: 24 carried rows per lane, the 8 oldest rows are final after each step;
//  * inverse row FFT: the finished 8-row block is transformed once (again 64 lanes = 8 rows x 8 pairs), multiplied by
//    the synthesis window, overlap-added along x (inside the tile pair in registers, across pairs by DPP row shifts)
//    and stored to the strip's slab -- the same slab layout as wiener_stream, so the finish kernels are shared.
// Per tile this is 2 + 1/4 + 1/4 FFT passes instead of 4, and the only transpositions are the two hand-overs
// between the row and the column stage (8 rows per step each) instead of four 32 x 32 LDS transposes.
// The three roles rotate over the 4 waves of a workgroup: every wave runs the column pipeline for its two tile
// pairs; in step i wave i % 4 also runs the forward row stage and wave (i + 2) % 4 the inverse row stage.
// Samples reach the row stage through an LDS staging block that all 256 threads fill one step ahead (global loads
// issued at the top of a step, consumed at its end), so no wave waits on HBM.
#pragma once

namespace ys {

#if defined(TDK_EXPERIMENTS) && defined(TDK_YS_TIMING)
// experiments: clock64() deltas per phase and wave of one workgroup, summed over its steps (profiles/wiener_ablate_exp.py)
__device__ unsigned long long g_ys_phase_cycles[4][8];
__device__ unsigned long long g_ys_wg_times[2 * 2048];  // [2 b], [2 b + 1]: wall_clock64() (100 MHz) at the start / end of workgroup b
#define YS_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); ys_acc[k] += t_ - ys_t0; ys_t0 = t_; } while (0)  // wave-uniform: lives in SGPRs
#else
#define YS_MARK(k)
#endif

#if defined(TDK_EXPERIMENTS) && defined(TDK_YS_ABLATE)
#define YS_ABLATE(n) (TDK_YS_ABLATE == (n))  // timing-only builds (wrong results): profiles/wiener_ablate_exp.py
#else
#define YS_ABLATE(n) false
#endif

constexpr int K = 32, S = 8, NPC = 8;     // tile size, hop, tile PAIRS per strip
constexpr int NTC = 2 * NPC;              // tile columns per strip (Geom::G)
constexpr int SW = NTC * S + K - S;       // samples per strip row: 152
constexpr int PITCH = 68;                 // floats per LDS row of 32 complex values (+4: conflict-free 16-B row accesses)
constexpr int BUF = 64 * PITCH;           // one hand-over block: 8 block rows x 8 tile pairs
// floats per LDS row of the staging block.  The forward row stage reads it as ds_read_b128 at row * PP + 16 * pair + k; a
// ds_read_b128 serves four fixed 16-lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), each of which holds, in
// this kernel's lane order, four block rows x four tile pairs.  The pairs of a row sit 16 floats = four 16-byte slots apart, so
// the four rows must start on four different slots modulo 4: PP / 4 odd.  With PP = SW = 152 (PP / 4 = 38) rows r and r + 2
// landed on the same slots -- a two-way conflict on every one of the stage's ten reads, a third of the kernel's LDS cycles
// (profiles/r04/experiments/lds_conflicts.txt); 156 is conflict-free.
constexpr int PP = SW + 4;
static_assert((PP / 4) % 2 == 1 && PP % 4 == 0, "staging pitch: odd number of 16-byte slots");
constexpr int GRP_PER_ROW = SW / 4;       // 16-B groups per staged row: 38
constexpr int GRP_PER_WAVE = 8 * GRP_PER_ROW / 4;  // 76

struct YParams {
  float whr[32], whi[32];  // FFT of the analysis window wf, natural kx order
};

struct __align__(16) Smem {
  float fwd[2][BUF];        // row stage -> column stage: R[block row][kx] (complex interleaved)
  float inv[2][BUF];        // column stage -> row stage: finished rows [block row][kx]
  float plane[2][8 * PP];   // staged samples of one 8-row block
  float meta[8][NPC * 2];   // ring over the last 8 blocks: per tile pair, the sums of the block's samples under tile a / tile b
};
static_assert(sizeof(Smem) <= 80 * 1024, "two workgroups per CU");

// LDS row of (tile pair c, block row r) = the lane that owns it in the row stages
__device__ __forceinline__ constexpr int row_of(int c, int r) { return (r & 1) + 2 * c + 16 * (r >> 1); }

template <int CTRL> __device__ __forceinline__ float dpp0(float x) {  // DPP move, 0 where the source lane does not exist
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

using tdk_fft::fft_inreg_pk;
using tdk_fft::v2f;

// x * wf[y] and wf[y] * x + c with the window value as a scalar operand: the 32 values (16 distinct: the window is
// symmetric) travel as 8 constant pairs
__device__ __forceinline__ v2f win_scale(int y, v2f x) {
  const int j = y < 16 ? y : 31 - y;
  const v2f wp = {WindowK<32>::w[j & ~1], WindowK<32>::w[j | 1]};
  return (j & 1) ? tdk_fft::pk_scale<1>(wp, x) : tdk_fft::pk_scale<0>(wp, x);
}
__device__ __forceinline__ v2f win_fma(int y, v2f x, v2f c) {
  const int j = y < 16 ? y : 31 - y;
  const v2f wp = {WindowK<32>::w[j & ~1], WindowK<32>::w[j | 1]};
  return (j & 1) ? tdk_fft::pk_fma_s<1>(wp, x, c) : tdk_fft::pk_fma_s<0>(wp, x, c);
}

// Partner exchange of the gains step, in place: x <- x of lane ^ 1 in every lane except the two self-conjugate lanes of each
// 32-lane slot, which keep their own value.  EXEC is narrowed to the exchanging lanes around the DPP moves (a lane that is
// switched off keeps its register, and the moves that would read it are switched off too), which saves the select that
// partner2() needs per value; one EXEC round trip serves a batch of up to 14 registers.  (The exchange assumes that both
// lanes of a pair are active, as they are under the wave-uniform branches of the column stage.)
#define YS_DPP1(r) "v_mov_b32_dpp %[" #r "], %[" #r "] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
template <int N> __device__ __forceinline__ void partner_swap(v2f* (&r)[N]) {
  static_assert(N >= 1 && N <= 7, "at most 14 registers per batch (asm operand limit)");
  constexpr unsigned long long EXCH = ~0x0000000300000003ull;
  float v[14];
#pragma unroll
  for (int i = 0; i < N; i++) { v[2 * i] = r[i]->x; v[2 * i + 1] = r[i]->y; }
#pragma unroll
  for (int i = 2 * N; i < 14; i++) v[i] = 0.0f;
  // s_nop 1: a VALU write needs two wait states before a DPP read of the same register.  EXEC is saved and restored (not forced
  // back to all ones), so the exchange stays correct if a caller ever sits under a narrowed EXEC.
  unsigned long long sv;
  if constexpr (N == 7)
    asm volatile("s_nop 1\n\ts_and_saveexec_b64 %[sv], %[m]\n\t" YS_DPP1(a) YS_DPP1(b) YS_DPP1(c) YS_DPP1(d) YS_DPP1(e) YS_DPP1(f) YS_DPP1(g) YS_DPP1(h) YS_DPP1(i) YS_DPP1(j)
                 YS_DPP1(k) YS_DPP1(l) YS_DPP1(n) YS_DPP1(o) "s_mov_b64 exec, %[sv]"
                 : [a] "+v"(v[0]), [b] "+v"(v[1]), [c] "+v"(v[2]), [d] "+v"(v[3]), [e] "+v"(v[4]), [f] "+v"(v[5]), [g] "+v"(v[6]), [h] "+v"(v[7]), [i] "+v"(v[8]),
                   [j] "+v"(v[9]), [k] "+v"(v[10]), [l] "+v"(v[11]), [n] "+v"(v[12]), [o] "+v"(v[13]), [sv] "=&s"(sv)
                 : [m] "s"(EXCH)
                 : "scc");
  else if constexpr (N == 2)
    asm volatile("s_nop 1\n\ts_and_saveexec_b64 %[sv], %[m]\n\t" YS_DPP1(a) YS_DPP1(b) YS_DPP1(c) YS_DPP1(d) "s_mov_b64 exec, %[sv]"
                 : [a] "+v"(v[0]), [b] "+v"(v[1]), [c] "+v"(v[2]), [d] "+v"(v[3]), [sv] "=&s"(sv)
                 : [m] "s"(EXCH)
                 : "scc");
  else
    asm volatile("s_nop 1\n\ts_and_saveexec_b64 %[sv], %[m]\n\t" YS_DPP1(a) YS_DPP1(b) "s_mov_b64 exec, %[sv]"
                 : [a] "+v"(v[0]), [b] "+v"(v[1]), [sv] "=&s"(sv)
                 : [m] "s"(EXCH)
                 : "scc");
#pragma unroll
  for (int i = 0; i < N; i++) *r[i] = v2f{v[2 * i], v[2 * i + 1]};
}

// wiener_gains on {re, im} pairs (same arithmetic; see wiener_gains in wiener.hip for the derivation).  With u = Z[k], p = Z[-k]
// (from the partner lane): a2 = u + conj(p) = 2A, d = u - conj(p) = 2 i B, and
//   Z'[k] = ga a2 + gb d,   the partner's Z'[-k] = conj(ga a2 - gb d).
// Three phases: (1) every register -k (k = 1 .. 15) is swapped with the partner lane in place (its own value is only
// needed there), k = 0 and 16 through copies; (2) the arithmetic, free of cross-lane operations, so the scheduler can
// interleave the 17 chains; (3) the by-products travel back.
__device__ __forceinline__ void wiener_gains_pk(v2f (&z)[32], float sig2) {
  constexpr float GSCALE = 0.5f / (float)(32 * 32);
  float sgs = -4.0f * sig2 * GSCALE;
  asm volatile("" : "+v"(sgs));
  v2f p0 = z[0], p16 = z[16];
  {
    v2f* r1[7] = {&z[31], &z[30], &z[29], &z[28], &z[27], &z[26], &z[25]};
    partner_swap(r1);
    v2f* r2[7] = {&z[24], &z[23], &z[22], &z[21], &z[20], &z[19], &z[18]};
    partner_swap(r2);
    v2f* r3[2] = {&z[17], &p0};
    partner_swap(r3);
    v2f* r4[1] = {&p16};
    partner_swap(r4);
  }
#pragma unroll
  for (int k = 0; k <= 16; k++) {
    const int k2 = (32 - k) & 31;
    const v2f u = z[k];
    const v2f p = k == 0 ? p0 : (k == 16 ? p16 : z[k2]);
    v2f a2, d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(a2) : "v"(u), "v"(p));  // u + conj(p)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(u), "v"(p));   // u - conj(p)
    const float pa4 = __builtin_fmaf(a2.y, a2.y, __builtin_fmaf(a2.x, a2.x, 4e-15f)), pb4 = __builtin_fmaf(d.y, d.y, __builtin_fmaf(d.x, d.x, 4e-15f));
    v2f gg;  // {ga, gb}
    gg.x = fmaxf(__builtin_fmaf(sgs, __builtin_amdgcn_rcpf(pa4), GSCALE), 0.0f);
    gg.y = fmaxf(__builtin_fmaf(sgs, __builtin_amdgcn_rcpf(pb4), GSCALE), 0.0f);
    v2f t, zk;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(gg), "v"(d));              // gb d
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(zk) : "v"(gg), "v"(a2), "v"(t));            // ga a2 + gb d
    z[k] = zk;
    if (k2 != k) {
      v2f w;
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[1,0,0]" : "=v"(w) : "v"(gg), "v"(a2), "v"(t));  // conj(ga a2 - gb d)
      z[k2] = w;
    }
  }
  {
    v2f* r1[7] = {&z[31], &z[30], &z[29], &z[28], &z[27], &z[26], &z[25]};
    partner_swap(r1);
    v2f* r2[7] = {&z[24], &z[23], &z[22], &z[21], &z[20], &z[19], &z[18]};
    partner_swap(r2);
    v2f* r3[1] = {&z[17]};
    partner_swap(r3);
  }
}
#undef YS_DPP1

// Raw staging registers: the global loads of a step's samples are issued at the top of the step and only converted /
// written to LDS at its bottom, so nothing waits on them.
template <typename T> struct Raw4;
template <> struct Raw4<float> {
  float4 v;
  __device__ __forceinline__ void load4(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ float4 get() const { return v; }
};
template <> struct Raw4<__half> {
  uint2 v;
  __device__ __forceinline__ void load4(const __half* p) { v = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ float4 get() const {
    auto h = [](unsigned bits) { return __half2float(__ushort_as_half((unsigned short)bits)); };
    return make_float4(h(v.x & 0xffffu), h(v.x >> 16), h(v.y & 0xffffu), h(v.y >> 16));
  }
};

// Four interleaved RGB pixels as loaded (fused log-luminance input): converted at the bottom of the step.
template <typename T> struct RawRgb4;
template <> struct RawRgb4<float> {
  float4 a, b, c;
  __device__ __forceinline__ void load(const float* p) { const float4* q = reinterpret_cast<const float4*>(p); a = q[0]; b = q[1]; c = q[2]; }
  __device__ __forceinline__ void get(float (&v)[12]) const {
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
  }
};
template <> struct RawRgb4<__half> {
  uint2 a, b, c;
  __device__ __forceinline__ void load(const __half* p) { const uint2* q = reinterpret_cast<const uint2*>(p); a = q[0]; b = q[1]; c = q[2]; }
  __device__ __forceinline__ void get(float (&v)[12]) const {
    const unsigned w[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
#pragma unroll
    for (int k = 0; k < 6; k++) {
      v[2 * k] = __half2float(__ushort_as_half((unsigned short)(w[k] & 0xffffu)));
      v[2 * k + 1] = __half2float(__ushort_as_half((unsigned short)(w[k] >> 16)));
    }
  }
};

// LUM = false: the samples are one plane (C = 1) or one channel of an interleaved image (C = 3).
// LUM = true: the image is interleaved RGB and the samples are its log-lightness log(max(eps, Lab L)) -- exactly the
// values compute_log_luminance (color.hip, lum_extract_vec4) would have written to a plane first (reference
// denoise.py:54-56).  The conversion is done by the two waves of a step that run no row stage, in the time they
// would otherwise wait at the barrier for the two that do.
template <typename T, bool LUM>
__global__ __launch_bounds__(256, 2) void wiener_ystream(const T* __restrict__ img, float* __restrict__ slabs, int W, int H, int C, int chan0, int vec_ok, Geom g,
                                                         const float* __restrict__ sigmas, YParams yp, size_t plane_stride, float lum_eps) {
  __shared__ Smem sm;
#if defined(TDK_EXPERIMENTS) && defined(TDK_YS_TIMING)
  if (threadIdx.x == 0 && blockIdx.x < 2048u) g_ys_wg_times[2 * blockIdx.x] = wall_clock64();
#endif
  int chan = chan0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ngroups = g.ngx * g.ngy;
  const int plane = blockIdx.x / ngroups, gin = blockIdx.x - plane * ngroups;
  const int gy = gin / g.ngx, gx = gin - gy * g.ngx;
  const float sigma = sigmas[chan + plane];
  if (C == 3) chan += plane;  // interleaved input: group set p works on channel p of the same image
  const float sig2 = sigma * sigma;
  const T* base = img + (size_t)plane * plane_stride;
  const int t0 = gy * g.TR, t1 = min(t0 + g.TR, g.nty);
  const int NB = t1 - t0 + 3;                 // 8-row blocks this strip segment reads = blocks its slab holds
  const int px0 = (g.jmin + gx * NTC) * S;    // image x of strip sample 0
  // samples beyond the last active tile of the strip are never used (and may lie beyond one reflection of the frame)
  const int sx_lim = S * min(g.ntx - gx * NTC, NTC) + K - S;
  float* slab = slabs + (size_t)blockIdx.x * (size_t)(g.RSY * g.RSXP);

  // column-stage coordinates: a 32-lane slot per tile pair, lane = kx (partner bin -kx in lane ^ 1)
  const int slot = lane >> 5, l = lane & 31, yc = 2 * wave + slot;
  const int kx = (l == 0) ? 0 : (l == 1) ? K / 2 : (l & 1) ? K - (l >> 1) : (l >> 1);  // == lane_freq(l, K)
  const bool ya = gx * NTC + 2 * yc < g.ntx, yb = gx * NTC + 2 * yc + 1 < g.ntx;
  const float whr = yp.whr[kx], whi = yp.whi[kx];
  const int col_off = 2 * yc * PITCH + 2 * kx;  // this lane's element of block row 0 of its tile pair in a hand-over block
  // row-stage coordinates: lane = row_of(xc, xr)
  const int xr = (lane & 1) + 2 * (lane >> 4), xc = (lane >> 1) & 7;
  const bool xa = gx * NTC + 2 * xc < g.ntx, xb = gx * NTC + 2 * xc + 1 < g.ntx;
  // staging coordinates: this wave's 76 groups of 4 samples of an 8 x 152 block (lanes < 12 take a second group)
  const int sg0 = GRP_PER_WAVE * wave + lane, sg1 = sg0 + 64;
  const bool has1 = lane < GRP_PER_WAVE - 64;
  const int so0 = (sg0 / GRP_PER_ROW) * PP + 4 * (sg0 % GRP_PER_ROW), so1 = (sg1 / GRP_PER_ROW) * PP + 4 * (sg1 % GRP_PER_ROW);  // their places in the staging block
  const bool vec = vec_ok && C == 1;

  v2f win[32];    // R[y][kx] of the current window (tile-relative row y), {re, im} register pairs
  v2f carry[24];  // carried (unfinished) rows of the overlap-add across tile rows, tile-relative
#pragma unroll
  for (int k = 0; k < 32; k++) win[k] = v2f{0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 24; k++) carry[k] = v2f{0.0f, 0.0f};

  // 4 samples of block q (image rows 8 q - 24 ...), strip columns 4 (grp % 38) ...  fetch() issues the one 16-B / 8-B load
  // of an in-frame group at the top of a step; a group that touches the frame edge (reflected single samples, or nothing
  // beyond the last active tile) is read by fetch_edge() at the bottom instead.  Keeping the two apart matters: loads into the
  // same registers from both paths would make the compiler drain vmcnt between them -- behind the step's slab stores.
  auto fetch = [&](int grp, int q, Raw4<T>& raw) -> bool {
    const int brow = grp / GRP_PER_ROW, x = px0 + 4 * (grp - brow * GRP_PER_ROW);
    if (!(vec && x >= 0 && x + 4 <= W)) return false;
    raw.load4(base + (size_t)(unsigned)reflect_index(8 * q + brow + g.jmin * S, H) * (unsigned)W + x);
    return true;
  };
  auto fetch_edge = [&](int grp, int q) -> float4 {
    const int brow = grp / GRP_PER_ROW, col = 4 * (grp - brow * GRP_PER_ROW);
    const T* rowp = base + (size_t)reflect_index(8 * q + brow + g.jmin * S, H) * W * C;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (col + j < sx_lim) ? ld(rowp, (size_t)reflect_index(px0 + col + j, W) * C + chan) : 0.0f;
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  auto emit = [&](float* dst, const v2f (&o)[8]) {
#pragma unroll
    for (int r = 0; r < 8; r++) *reinterpret_cast<v2f*>(dst + row_of(0, r) * PITCH) = o[r];
  };

  // The per-lane constants above come from global loads (kernel-argument tables indexed by lane).  Retire them here: a
  // first use inside the loop makes the compiler wait with vmcnt(0) there in EVERY step -- behind the step's staging loads,
  // whose latency the whole design is built to hide.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::"v"(whr), "v"(whi));
#if defined(TDK_EXPERIMENTS) && defined(TDK_YS_TIMING)
  unsigned long long ys_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long ys_t0 = __builtin_readcyclecounter();
#endif
  // Two 8-row blocks per iteration, two phases with a barrier after each:
  //  A (row stages): waves 0 / 1 transform blocks 2 (it - 1), 2 (it - 1) + 1 of the staging buffer forward, waves 2 / 3 take the two
  //    blocks the column stage finished in the previous iteration through the inverse row stage to the slab -- every wave runs
  //    exactly one row stage, so none waits for another at the barrier (one block per step with the roles rotating left two of
  //    the four waves idle at the barrier for the length of a row stage: 19 % of a wave's time, profiles/r03);
  //  B (column stage): every wave moves its two tile pairs down by the two blocks, then stores the samples of blocks 2 it, 2 it + 1
  //    (their global loads were issued at the top of the phase) for the next iteration's forward stage.
  // The hand-over buffers are single: fwd / meta are written in A and read in B, inv and the staging block in B and A.
  const int NJ = (NB + 4) / 2 + 2;
  const int rblk = wave & 1, sblk = wave >> 1;  // the block of the pair this wave takes in its row stage / stages
  const int wslot = TDK_FAIR_PRIO ? tdk_wave_slot() : 0;
  for (int it = 0; it < NJ; it++) {
    tdk_rotate_prio<2>(it, wslot);
    // ---- phase A
    if (wave >= 2) {
      // inverse row stage: the block emitted in the previous iteration (slab block e)
      const int e = 2 * it - 7 + rblk;
      if (e >= 0 && e < NB && !YS_ABLATE(1)) {
        const float* src = sm.inv[rblk] + lane * PITCH;
        v2f z[K];
#pragma unroll
        for (int k = 0; k < K; k++) z[k] = *reinterpret_cast<const v2f*>(src + 2 * k);
        fft_inreg_pk<32, true>(z);  // .x = row of tile a, .y = row of tile b (columns S further)
        float s_[K + S];
#pragma unroll
        for (int u = 0; u < K + S; u++) {
          float v = (u < K) ? z[u].x * WindowK<32>::w[u] : 0.0f;
          if (u >= S) v = (u < K) ? __builtin_fmaf(z[u - S].y, WindowK<32>::w[u - S], v) : z[u - S].y * WindowK<32>::w[u - S];
          s_[u] = v;
        }
        // overlap-add along x across tile pairs: pair c's samples [16, 32) belong to pair c + 1's [0, 16), its [32, 40) to
        // pair c + 2's [0, 8) -- two and four lanes up in the 16-lane DPP row (lane = (r & 1) + 2 c + 16 (r >> 1))
        float* srow = slab + (size_t)(8 * e + xr) * g.RSXP;
        float tl[K - S];
#pragma unroll
        for (int k = 0; k < K - S; k++) {                       // the strip's last 24 columns (held by pair 7; pair 6 adds 8 of them)
          float v = s_[2 * S + k];
          if (k < S) v += dpp0<0x112>(s_[4 * S + k]);
          asm volatile("" : "+v"(v));                           // keep the DPP reads out of the lane-masked branch below
          tl[k] = v;
        }
#pragma unroll
        for (int k = 0; k < 2 * S; k += 4) {
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            o[j] = s_[k + j] + dpp0<0x112>(s_[2 * S + k + j]);        // row_shr:2
            if (k + j < S) o[j] += dpp0<0x114>(s_[4 * S + k + j]);    // row_shr:4
          }
          *reinterpret_cast<float4*>(srow + 2 * S * xc + k) = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (xc == NPC - 1) {
#pragma unroll
          for (int k = 0; k < K - S; k += 4) *reinterpret_cast<float4*>(srow + NTC * S + k) = make_float4(tl[k], tl[k + 1], tl[k + 2], tl[k + 3]);
        }
      }
      YS_MARK(3);  // inverse row stage
    } else {
      // forward row stage: block bb from the staging buffer -> R rows
      const int bb = 2 * (it - 1) + rblk;
      if (bb >= 0 && bb < NB && !YS_ABLATE(1)) {
        const float* pl = sm.plane[rblk] + xr * PP + 2 * S * xc;
        float w[K + S];
#pragma unroll
        for (int k = 0; k < K + S; k += 4) {
          const float4 t = *reinterpret_cast<const float4*>(pl + k);
          w[k] = t.x; w[k + 1] = t.y; w[k + 2] = t.z; w[k + 3] = t.w;
        }
        float head = 0.0f, mid = 0.0f, tail = 0.0f;
#pragma unroll
        for (int k = 0; k < S; k++) head += w[k];
#pragma unroll
        for (int k = S; k < K; k++) mid += w[k];
#pragma unroll
        for (int k = K; k < K + S; k++) tail += w[k];
        // block sums under tile a / tile b: over the 8 block rows = lane bits 0, 4, 5
        float ba = xa ? head + mid : 0.0f, bb_ = xb ? mid + tail : 0.0f;
        ba += dpp0<0xB1>(ba); bb_ += dpp0<0xB1>(bb_);  // quad_perm [1,0,3,2]
        // over lane bits 4 and 5 with the gfx950 row swaps, both sums at once (the same pairs in the same order as two xor
        // shuffles each, at VALU speed instead of four trips through the LDS crossbar).  16-lane rows of (ba | bb_):
        // (a0 a1 a2 a3 | b0 b1 b2 b3) -- v_permlane16_swap: odd rows of the first with even rows of the second -->
        // (a0 b0 a2 b2 | a1 b1 a3 b3), their sum (a01 b01 a23 b23); then both copies of it through v_permlane32_swap (upper half
        // of the first with the lower half of the second): (a01 b01 a01 b01 | a23 b23 a23 b23) -> rows 0 / 2 hold the sum under
        // tile a, rows 1 / 3 the sum under tile b.  (s_nop 1: a VALU write needs two wait states before a lane swap reads it.)
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ba), "+v"(bb_));
        float s1 = ba + bb_, s2 = s1;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s1), "+v"(s2));
        if ((lane & 1) == 0 && lane < 32) sm.meta[bb & 7][2 * xc + (lane >> 4)] = s1 + s2;
        v2f z[K];
        if (__builtin_amdgcn_ballot_w64(!(xa && xb)) == 0) {  // every tile of the strip exists (all strips but the last one)
#pragma unroll
          for (int k = 0; k < K; k++) z[k] = v2f{w[k] * WindowK<32>::w[k], w[k + S] * WindowK<32>::w[k]};
        } else {
          const float fa = xa ? 1.0f : 0.0f, fb = xb ? 1.0f : 0.0f;
#pragma unroll
          for (int k = 0; k < K; k++) z[k] = v2f{(w[k] * fa) * WindowK<32>::w[k], (w[k + S] * fb) * WindowK<32>::w[k]};
        }
        fft_inreg_pk<32, false>(z);
        float* dst = sm.fwd[rblk] + lane * PITCH;
#pragma unroll
        for (int k = 0; k < K; k++) *reinterpret_cast<v2f*>(dst + 2 * k) = z[k];
      }
      YS_MARK(2);  // forward row stage
    }
    __syncthreads();
    YS_MARK(5);  // barrier

    // ---- phase B, once per block of the pair: issue the global loads of the samples of block qs = 2 it + blk, run the column stage
    // on block b = 2 (it - 1) + blk (tile row t0 + b - 3 is complete from b = 3 on; after the last block the three carried blocks
    // leave as they are: partial sums for the seam with the next strip segment), then store the samples for the next iteration's
    // forward row stage (a wave's share: 76 groups of 4 samples; LUM: waves 0 / 1 convert block rows 0 .. 3 / 4 .. 7 in the round
    // of block 0, waves 2 / 3 in the round of block 1)
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {
      const int b = 2 * (it - 1) + blk;
      const int qs = 2 * it + blk;
      Raw4<T> st0, st1;
      RawRgb4<T> px[3];
      bool pk0 = false, pk1 = false, pkx[3] = {false, false, false};
      if constexpr (!LUM) {
        if (qs < NB) {
          pk0 = fetch(sg0, t0 + qs, st0);
          if (has1) pk1 = fetch(sg1, t0 + qs, st1);
        }
      } else {
        if (qs < NB && sblk == blk) {
#pragma unroll
          for (int u = 0; u < 3; u++) {
            const int grp = lane + 64 * u;  // of the 4 x 38 groups of this half block
            if (grp < 4 * GRP_PER_ROW) {
              const int brow = 4 * rblk + grp / GRP_PER_ROW, x = px0 + 4 * (grp % GRP_PER_ROW);
              if (vec_ok && x >= 0 && x + 4 <= W) {
                px[u].load(base + ((size_t)reflect_index(8 * (t0 + qs) + brow + g.jmin * S, H) * W + x) * 3);
                pkx[u] = true;
              }
            }
          }
        }
      }
      YS_MARK(0);  // staging loads issued
      if (b >= 0 && b < NB) {
        // the window moves down one block: rows 8 .. 31 become rows 0 .. 23 (24 pair moves: cheaper than what the compiler
        // makes of a code variant per window phase), the arriving block becomes rows 24 .. 31
#pragma unroll
        for (int y = 0; y < 24; y++) win[y] = win[y + 8];
        const float* f = sm.fwd[blk] + col_off;
#pragma unroll
        for (int r = 0; r < 8; r++) win[24 + r] = *reinterpret_cast<const v2f*>(f + row_of(0, r) * PITCH);
        if (b >= 3 && !YS_ABLATE(2)) {
          float sa = 0.0f, sb = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float2 t = *reinterpret_cast<const float2*>(sm.meta[(b - q) & 7] + 2 * yc);
            sa += t.x;
            sb += t.y;
          }
          const float mean_a = ya ? sa * (1.0f / (K * K)) : 0.0f, mean_b = yb ? sb * (1.0f / (K * K)) : 0.0f;
          const v2f m = {mean_a * whr - mean_b * whi, mean_a * whi + mean_b * whr};  // (mean_a + i mean_b) W[kx]
          v2f z[32];
#pragma unroll
          for (int y = 0; y < 32; y++) z[y] = win_scale(y, win[y] - m);  // (R - mean W) wf[y]
          if (!YS_ABLATE(5)) fft_inreg_pk<32, false>(z);
          if (!YS_ABLATE(4) && !YS_ABLATE(5)) wiener_gains_pk(z, sig2);
          if (!YS_ABLATE(5)) fft_inreg_pk<32, true>(z);
          // (v + mean wf[y] W[kx]) * wi[y], in units of 1/32 (the inverse row pass is unscaled); overlap-add across tile rows
          const v2f ma = m * (1.0f / K);
          {
            v2f o[8];
#pragma unroll
            for (int y = 0; y < 8; y++) o[y] = win_fma(y, win_fma(y, ma, z[y]), carry[y]);
            emit(sm.inv[blk] + col_off, o);
          }
          // the carried rows move up one block while they are updated: row y takes the sum of row y + 8.  In this order every
          // register is read before it is rewritten; the scheduling barriers keep the order, so the move costs no copies
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int y = 0; y < 8; y++) carry[y] = win_fma(y + 8, win_fma(y + 8, ma, z[y + 8]), carry[y + 8]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int y = 8; y < 16; y++) carry[y] = win_fma(y + 8, win_fma(y + 8, ma, z[y + 8]), carry[y + 8]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int y = 16; y < 24; y++) carry[y] = win_scale(y + 8, win_fma(y + 8, ma, z[y + 8]));
        }
      } else if (b >= NB && b < NB + 3) {
        v2f o[8];
#pragma unroll
        for (int y = 0; y < 8; y++) o[y] = carry[y];
        emit(sm.inv[blk] + col_off, o);
#pragma unroll
        for (int y = 0; y < 16; y++) carry[y] = carry[y + 8];
      }
      YS_MARK(1);  // column stage

      // ---- staging, second half: the samples loaded at the top of the round go to LDS for the next iteration's forward row stage
      if constexpr (!LUM) {
        if (qs < NB) {
          float* pl = sm.plane[blk];
          *reinterpret_cast<float4*>(pl + so0) = pk0 ? st0.get() : fetch_edge(sg0, t0 + qs);
          if (has1) *reinterpret_cast<float4*>(pl + so1) = pk1 ? st1.get() : fetch_edge(sg1, t0 + qs);
        }
      } else {
#pragma clang fp contract(off)
        if (qs < NB && sblk == blk) {
#pragma unroll
          for (int u = 0; u < 3; u++) {
            const int grp = lane + 64 * u;
            if (grp < 4 * GRP_PER_ROW) {
              const int brow = 4 * rblk + grp / GRP_PER_ROW, col = 4 * (grp % GRP_PER_ROW);
              float v[12], lv[4];
              if (pkx[u]) {
                px[u].get(v);
              } else {  // frame edges: reflected single pixels (or nothing beyond the last active tile), loaded here
                const T* rowp = base + (size_t)reflect_index(8 * (t0 + qs) + brow + g.jmin * S, H) * W * 3;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                  const bool need = col + j < sx_lim;
                  const T* q = rowp + (size_t)(need ? reflect_index(px0 + col + j, W) : 0) * 3;
                  v[3 * j] = need ? ld(q, 0) : 0.0f; v[3 * j + 1] = need ? ld(q, 1) : 0.0f; v[3 * j + 2] = need ? ld(q, 2) : 0.0f;
                }
              }
#pragma unroll
              for (int j = 0; j < 4; j++) {  // same expression as lum_extract_vec4 (color.hip)
                const float yv = YS_ABLATE(3) ? v[3 * j] + v[3 * j + 1] + v[3 * j + 2] : cA::rgb_to_lab_l(clip3(mk3(v[3 * j], v[3 * j + 1], v[3 * j + 2])));
                lv[j] = YS_ABLATE(3) ? yv : tdk_log(fmaxf(lum_eps, yv));
              }
              *reinterpret_cast<float4*>(sm.plane[blk] + brow * PP + col) = make_float4(lv[0], lv[1], lv[2], lv[3]);
            }
          }
        }
      }
      // Every global operation of the round has retired by now (the slab stores were issued in phase A, the loads are consumed
      // just above).  Saying so explicitly keeps the compiler from placing its own vmcnt(0) at the top of the next round or behind
      // the next iteration's slab stores, where it would expose their latency (its scoreboard cannot see that the load and the
      // use of the staging registers sit under the same condition).
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
      YS_MARK(4);  // staging conversion / store
    }
    __syncthreads();
    YS_MARK(5);  // barrier
    YS_MARK(6);  // nothing: the cost of a mark itself
  }
#if defined(TDK_EXPERIMENTS) && defined(TDK_YS_TIMING)
  if (lane == 0 && blockIdx.x == 200)
    for (int k = 0; k < 7; k++) g_ys_phase_cycles[wave][k] += ys_acc[k];
  if (threadIdx.x == 0 && blockIdx.x < 2048u) g_ys_wg_times[2 * blockIdx.x + 1] = wall_clock64();
#endif
}

}  // namespace ys
