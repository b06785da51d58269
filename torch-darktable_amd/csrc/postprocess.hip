// postprocess.hip -- demosaic post-processing and per-site white balance.
//
// Replaces reference csrc/debayer/postprocess.cu:24-402 (colour smoothing, global and local
// green equilibration; PostProcessImpl::process) and csrc/white_balance.cu:10-42,164-183
// (apply_white_balance).
//
// MI355X design
//  * colour smoothing: 64 x 16 tile per 512-thread workgroup; up to four passes run in one launch on
//    (R-G, B-G) planes in LDS (zero outside the image, as the reference's halo fill), the median
//    of nine is 12 min3/med3/max3 instructions, each thread emits 4 pixels as three 16-B stores.
//  * global green equilibration: the reference reduces per block, sums the partials with a
//    torch op and reads two scalars back to the host (postprocess.cu:362-366).  Here a fixed
//    grid of workgroups writes partial sums, a single-workgroup kernel folds them in a fixed
//    order and leaves the ratio in device memory, and the apply kernel reads it from there:
//    no host sync, deterministic run to run.
//  * ping-pong between the caller's output and one scratch image so the last stage lands in
//    the output; the reference's copy-in and final clone disappear.
#include "tdk_stencils.h"

namespace {

// Median of nine (reference csrc/reduction.h:93-116 uses a 19-exchange sorting network).  The
// median is a VALUE, not an order of operations, so any exact selection gives the same bits:
// with lo/mid/hi = min3/med3/max3 of each row of three,
//   median9 = med3(max3(lo0, lo1, lo2), med3(mid0, mid1, mid2), min3(hi0, hi1, hi2))
// -- 12 three-input VALU instructions instead of 57 compare/selects.
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ float median9(float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7, float s8) {
  const float lo = max3f(min3f(s0, s1, s2), min3f(s3, s4, s5), min3f(s6, s7, s8));
  const float mid = med3f(med3f(s0, s1, s2), med3f(s3, s4, s5), med3f(s6, s7, s8));
  const float hi = min3f(max3f(s0, s1, s2), max3f(s3, s4, s5), max3f(s6, s7, s8));
  return med3f(lo, mid, hi);
}

// four consecutive values to storage type T at element index e (16 / 8-byte aligned)
template <typename T> __device__ __forceinline__ void st4(T* p, size_t e, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, size_t e, float4 v) { *reinterpret_cast<float4*>(p + e) = v; }
template <> __device__ __forceinline__ void st4<__half>(__half* p, size_t e, float4 v) {
  const float q[4] = {v.x, v.y, v.z, v.w};
  s4_io<__half>::store(p + e, 0, q);
}

constexpr int STW = 64, STH = 16, SNT = 512;  // tile and threads per workgroup: 36 KB of LDS -> 4 workgroups x 8 waves per CU

// Up to FMAXP smoothing passes in ONE kernel: the tile + P-px halo is read once, the (R-G, B-G)
// planes ping-pong in LDS while the valid region shrinks by one pixel per pass, and only the last
// pass touches HBM again -- 24 B/px of traffic instead of 24 B/px per pass.  Every pass computes
// exactly one reference pass (postprocess.cu:24-78; differences are zero outside the image at every pass).
//
// Round 5 (was 115 us for three passes at 12 MP, a third of its LDS cycles in bank conflicts):
//  * staging: one thread = 4 consecutive pixels of a row = three 16-B loads (the region is taken from the 4-aligned column
//    x0 - 4), instead of three 4-byte loads 12 bytes apart per pixel; the planes go to LDS as 16-B row groups;
//  * passes: a thread walks a short vertical run of one column with the 3 x 3 window in registers -- 3 new samples per plane
//    and pixel instead of 9, and the lanes of a wave sit on consecutive columns of one row: unit-stride, conflict-free
//    (two pixels side by side per thread made every read a stride-2 access);
//  * last pass: results go through the two dead planes as an interleaved RGB tile and leave as 16-B stores.
constexpr int FMAXP = 4;
constexpr int FCO = 4;                 // LDS column of image column x0 (>= FMAXP, a multiple of 4: staged groups are 16-B aligned)
constexpr int FLW = STW + 2 * FCO;     // 72 staged columns
constexpr int FLH = STH + 2 * FMAXP;   // 24 rows; LDS row of image row y0 = FMAXP
constexpr int FLS = 76;                // row stride: a multiple of 4
constexpr int FPL = (FLH + 2) * FLS;   // one plane (+2 rows: the last run of a pass reads up to two rows past the region; never used)
static_assert(4 * 5 * FPL * sizeof(float) <= 160 * 1024, "four workgroups per CU");
constexpr int FGR = FLW / 4;           // 18 four-pixel groups per staged row
static_assert(FGR * FLH <= SNT, "staging: one group per thread");
static_assert(2 * FPL >= STH * STW * 3, "the last pass's RGB tile fits two planes");

// vec_in / vec_ok: width % 4 == 0 and `in` / `out` aligned to four of their elements (16 bytes fp32, 8 bytes binary16).
// TI / TO: storage types of `in` and `out` (fp32 between the stages of a call, the caller's type at its two ends)
template <typename TI, typename TO>
__global__ __launch_bounds__(SNT) void smoothing_fused_kernel(const TI* __restrict__ in, TO* __restrict__ out, int width, int height, int vec_in, int vec_ok,
                                                              int passes) {
  __shared__ __align__(16) float lds[5 * FPL];  // G | DR0 | DB0 | DR1 | DB1
  float* G = lds;
  auto DR = [&](int k) { return lds + (1 + 2 * k) * FPL; };
  auto DB = [&](int k) { return lds + (2 + 2 * k) * FPL; };
  const int P = passes;  // 1..FMAXP
  const int x0 = blockIdx.x * STW, y0 = blockIdx.y * STH;
  const int tid = threadIdx.x;
  // ---- staging: rows y0 - P .. y0 + STH + P - 1, columns x0 - FCO .. x0 + STW + FCO - 1
  {
    const int rh = STH + 2 * P;
    if (tid < FGR * rh) {
      const int r = tid / FGR, g4 = tid - r * FGR;
      const int gy = y0 - P + r, gx = x0 - FCO + 4 * g4;
      float g[4] = {0.0f, 0.0f, 0.0f, 0.0f}, a[4] = {0.0f, 0.0f, 0.0f, 0.0f}, b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (gy >= 0 && gy < height) {
        if (vec_in && gx >= 0 && gx + 4 <= width) {  // width % 4 == 0 and a 16-B aligned image: the group is three aligned 16-B loads
          float v[12];
          rgb4_io<TI>::load(in, ((size_t)gy * width + gx) >> 2, v);
#pragma unroll
          for (int k = 0; k < 4; k++) { g[k] = v[3 * k + 1]; a[k] = v[3 * k] - g[k]; b[k] = v[3 * k + 2] - g[k]; }
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (gx + k >= 0 && gx + k < width) {
              const size_t p = ((size_t)gy * width + gx + k) * 3;
              g[k] = ld<TI>(in, p + 1); a[k] = ld<TI>(in, p) - g[k]; b[k] = ld<TI>(in, p + 2) - g[k];
            }
        }
      }
      const int q = (FMAXP - P + r) * FLS + 4 * g4;
      const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      *reinterpret_cast<float4*>(G + q) = make_float4(g[0], g[1], g[2], g[3]);
      *reinterpret_cast<float4*>(DR(0) + q) = make_float4(a[0], a[1], a[2], a[3]);
      *reinterpret_cast<float4*>(DB(0) + q) = make_float4(b[0], b[1], b[2], b[3]);
      *reinterpret_cast<float4*>(DR(1) + q) = z;  // out-of-image sites stay zero in both buffers
      *reinterpret_cast<float4*>(DB(1) + q) = z;
    }
  }
  __syncthreads();
  int cur = 0;
  // ---- all passes but the last: region shrinks by one pixel per pass; a thread = one column, RS consecutive rows
  for (int p = 0; p + 1 < P; p++) {
    constexpr int RS = 3;
    const int m = P - 1 - p;  // halo still needed after this pass
    const int cw = STW + 2 * m, ch = STH + 2 * m, nseg = (ch + RS - 1) / RS;
    const float* dr = DR(cur);
    const float* db = DB(cur);
    float* dro = DR(cur ^ 1);
    float* dbo = DB(cur ^ 1);
    const float inv_cw = 1.0f / (float)cw;  // i / n by float reciprocal: exact for i < 2^20, n < 2^10
    for (int i = tid; i < cw * nseg; i += SNT) {
      const int sg = (int)(((float)i + 0.5f) * inv_cw), col = i - sg * cw;
      const int c = FCO - m + col, r0 = FMAXP - m + RS * sg;
      const int gx = x0 - FCO + c;
      if (gx < 0 || gx >= width) continue;
      const float* pr = dr + (r0 - 1) * FLS + c;
      const float* pb = db + (r0 - 1) * FLS + c;
      float wr[RS + 2][3], wb[RS + 2][3];
#pragma unroll
      for (int j = 0; j < RS + 2; j++) {  // (rows beyond the region are inside the plane: FMAXP rows / FCO columns of margin)
#pragma unroll
        for (int d = 0; d < 3; d++) { wr[j][d] = pr[j * FLS + d - 1]; wb[j][d] = pb[j * FLS + d - 1]; }
      }
#pragma unroll
      for (int k = 0; k < RS; k++) {
        const int r = r0 + k, gy = y0 - FMAXP + r;
        if (RS * sg + k >= ch || gy < 0 || gy >= height) continue;
        const int q = r * FLS + c;
        const float rm = median9(wr[k][0], wr[k][1], wr[k][2], wr[k + 1][0], wr[k + 1][1], wr[k + 1][2], wr[k + 2][0], wr[k + 2][1], wr[k + 2][2]);
        const float bm = median9(wb[k][0], wb[k][1], wb[k][2], wb[k + 1][0], wb[k + 1][1], wb[k + 1][2], wb[k + 2][0], wb[k + 2][1], wb[k + 2][2]);
        const float g = G[q];
        const float nr = fmaxf(fmaxf(rm + g, 0.0f), 0.0f), ng = fmaxf(g, 0.0f), nb = fmaxf(fmaxf(bm + g, 0.0f), 0.0f);
        G[q] = ng;  // only this thread reads G[q]
        dro[q] = nr - ng;
        dbo[q] = nb - ng;
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  // ---- last pass: wave = two tile rows, lane = column; the RGB tile is assembled in the two dead planes
  {
    const float* dr = DR(cur);
    const float* db = DB(cur);
    float* tile = DR(cur ^ 1);  // DR(k), DB(k) are adjacent: 2 FPL floats >= 16 x 192
    const int col = tid & 63, sg = tid >> 6;
    const int c = FCO + col, r0 = FMAXP + 2 * sg;
    const float* pr = dr + (r0 - 1) * FLS + c;
    const float* pb = db + (r0 - 1) * FLS + c;
    float wr[4][3], wb[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int d = 0; d < 3; d++) { wr[j][d] = pr[j * FLS + d - 1]; wb[j][d] = pb[j * FLS + d - 1]; }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const float rm = median9(wr[k][0], wr[k][1], wr[k][2], wr[k + 1][0], wr[k + 1][1], wr[k + 1][2], wr[k + 2][0], wr[k + 2][1], wr[k + 2][2]);
      const float bm = median9(wb[k][0], wb[k][1], wb[k][2], wb[k + 1][0], wb[k + 1][1], wb[k + 1][2], wb[k + 2][0], wb[k + 2][1], wb[k + 2][2]);
      const float g = G[(r0 + k) * FLS + c];
      float* t = tile + (2 * sg + k) * (3 * STW) + 3 * col;
      t[0] = fmaxf(fmaxf(rm + g, 0.0f), 0.0f);
      t[1] = fmaxf(g, 0.0f);
      t[2] = fmaxf(fmaxf(bm + g, 0.0f), 0.0f);
    }
    __syncthreads();
    const int cols = min(STW, width - x0);  // pixels of a tile row inside the image
    if (vec_ok && cols == STW) {            // whole 16-B groups: 48 per row
      for (int i = tid; i < STH * (3 * STW / 4); i += SNT) {
        const int r = i / (3 * STW / 4), k = i - r * (3 * STW / 4);
        if (y0 + r < height) st4<TO>(out, ((size_t)(y0 + r) * width + x0) * 3 + 4 * k, *reinterpret_cast<const float4*>(tile + r * (3 * STW) + 4 * k));
      }
    } else {
      for (int i = tid; i < STH * 3 * STW; i += SNT) {
        const int r = i / (3 * STW), f = i - r * (3 * STW);
        if (y0 + r < height && f < 3 * cols) st<TO>(out, ((size_t)(y0 + r) * width + x0) * 3 + f, tile[r * (3 * STW) + f]);
      }
    }
  }
}

// ---- global green equilibration
constexpr int GEQ_BLOCKS = 1024;

// partial[b] = (sum G on even rows, sum G on odd rows) over x < 2*(W/2), y < 2*(H/2)
template <typename TI>
__global__ __launch_bounds__(256) void green_sums_kernel(const TI* __restrict__ in, int width, int height, uint32_t pattern,
                                                         float2* __restrict__ partial) {
  __shared__ float s1[4], s2[4];
  const int we = 2 * (width / 2), he = 2 * (height / 2);
  float a = 0.0f, b = 0.0f;
  for (int y = blockIdx.x; y < he; y += gridDim.x)  // whole rows per workgroup: no index division
    for (int x = threadIdx.x; x < we; x += 256)
      if (cfa_color(y, x, pattern) == 1) {
        const float g = ld<TI>(in, ((size_t)y * width + x) * 3 + 1);
        if (y & 1) b += g;
        else a += g;
      }
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s1[wave] = a; s2[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = make_float2((s1[0] + s1[1]) + (s1[2] + s1[3]), (s2[0] + s2[1]) + (s2[2] + s2[3]));
}

// ratio = sum2 / sum1 if both > 0 else 1 (postprocess.cu:364-366); fixed-order fold
__global__ __launch_bounds__(256) void green_ratio_kernel(const float2* __restrict__ partial, int nblocks, float* __restrict__ ratio) {
  __shared__ float s1[4], s2[4];
  float a = 0.0f, b = 0.0f;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += partial[i].x; b += partial[i].y; }
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s1[wave] = a; s2[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float sum1 = (s1[0] + s1[1]) + (s1[2] + s1[3]), sum2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    ratio[0] = (sum1 > 0.0f && sum2 > 0.0f) ? sum2 / sum1 : 1.0f;
    ratio[1] = sum1;
    ratio[2] = sum2;
  }
}

// Streaming kernels below: grid = (x chunks, rows), one thread per VEC consecutive pixels of a row
// (VEC == 4: 48-B vector accesses; requires width % 4 == 0 and 16-B aligned images).

// postprocess.cu:234-255
template <int VEC, typename TI, typename TO>
__global__ __launch_bounds__(256) void green_apply_kernel(const TI* __restrict__ in, TO* __restrict__ out, int width, int height,
                                                          uint32_t pattern, const float* __restrict__ ratio) {
  const float r = ratio[0];
  const int ngroup = width / VEC;
  for (int y = blockIdx.y; y < height; y += gridDim.y)
    for (int gi = blockIdx.x * 256 + threadIdx.x; gi < ngroup; gi += gridDim.x * 256) {
      const size_t g4 = (size_t)y * ngroup + gi;
      float v[3 * VEC];
      if constexpr (VEC == 4) rgb4_io<TI>::load(in, g4, v);
      else { v[0] = ld<TI>(in, g4 * 3); v[1] = ld<TI>(in, g4 * 3 + 1); v[2] = ld<TI>(in, g4 * 3 + 2); }
#pragma unroll
      for (int k = 0; k < VEC; k++) {
        const bool g1 = (cfa_color(y, gi * VEC + k, pattern) == 1) && !(y & 1);
        const float g = v[3 * k + 1] * (g1 ? r : 1.0f);
        v[3 * k] = fmaxf(v[3 * k], 0.0f);
        v[3 * k + 1] = fmaxf(g, 0.0f);
        v[3 * k + 2] = fmaxf(v[3 * k + 2], 0.0f);
      }
      if constexpr (VEC == 4) rgb4_io<TO>::store(out, g4, v);
      else { st<TO>(out, g4 * 3, v[0]); st<TO>(out, g4 * 3 + 1, v[1]); st<TO>(out, g4 * 3 + 2, v[2]); }
    }
}

// postprocess.cu:84-169; threshold already divided by 100.  One workgroup = 64 x 16 pixels: the
// green channel of the tile + 2-px halo is staged in LDS once (zero outside the image, as the
// reference's guarded loads), so the eight neighbour greens of a G2 site are LDS reads instead of
// eight strided gathers from the interleaved image.
// Round 5 (was 81 us at 12 MP, 46 % of its LDS cycles in bank conflicts, the image read twice): a thread loads its four
// pixels ONCE (three 16-B loads), keeps them across the barrier and writes their greens to the LDS tile; only the 2-px halo is
// gathered separately.  The tile's columns are de-interleaved by 4 (column c at (c >> 2) + (c & 3) * GLQ), so the lanes of
// a wave -- four pixels apart -- touch consecutive words for every tap; row stride 80 (== 16 mod 32) keeps the two rows of a
// 32-lane group on disjoint banks.
constexpr int GLW = 64, GLH = 16;
constexpr int GLQ = (GLW + 8) / 4;       // 18 words per column class: the tile carries 4 halo columns each side (2 used; keeps groups aligned)
constexpr int GLS = 80;                  // row stride
__device__ __forceinline__ int gl_at(int r, int c) { return r * GLS + (c >> 2) + (c & 3) * GLQ; }  // r in [0, 20), c in [0, 72): image (y0 - 2 + r, x0 - 4 + c)

template <int VEC, typename TI, typename TO>
__global__ __launch_bounds__(256) void green_local_kernel(const TI* __restrict__ in, TO* __restrict__ out, int width, int height,
                                                          uint32_t pattern, float threshold) {
  __shared__ float gt[(GLH + 4) * GLS];
  const int x0 = blockIdx.x * GLW, y0 = blockIdx.y * GLH;
  const int tid = threadIdx.x;
  auto green_at = [&](int gx, int gy) { return (gx >= 0 && gy >= 0 && gx < width && gy < height) ? ld<TI>(in, ((size_t)gy * width + gx) * 3 + 1) : 0.0f; };
  if constexpr (VEC == 4) {
    // own pixels: 16 threads x 4 px per row, 16 rows
    const int lx = (tid & 15) * 4, ly = tid >> 4;
    const int x = x0 + lx, y = y0 + ly;
    const bool live = x < width && y < height;  // width % 4 == 0: a group is inside or outside as a whole
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = 0.0f;
    const size_t p0 = (size_t)y * width + x;
    if (live) rgb4_io<TI>::load(in, p0 >> 2, v);
    // halo: rows y0 - 2, y0 - 1, y0 + 16, y0 + 17 over columns x0 - 2 .. x0 + 65 (4 x 68), then columns x0 - 2, x0 - 1, x0 + 64,
    // x0 + 65 of the 16 tile rows (16 x 4): 336 greens
    float hv[2];
    int hq[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int i = tid + 256 * u;
      hq[u] = -1;
      hv[u] = 0.0f;
      if (i < 4 * 68) {
        const int rr = i / 68, c = i - rr * 68, r = rr < 2 ? rr : GLH + rr;  // tile rows 0, 1, 18, 19
        hq[u] = gl_at(r, c + 2);
        hv[u] = green_at(x0 - 2 + c, y0 - 2 + r);
      } else if (i < 4 * 68 + GLH * 4) {
        const int j = i - 4 * 68, rr = j >> 2, cc = j & 3, c = cc < 2 ? 2 + cc : GLW + 2 + cc;  // tile columns 2, 3, 68, 69
        hq[u] = gl_at(rr + 2, c);
        hv[u] = green_at(x0 - 4 + c, y0 + rr);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) gt[gl_at(ly + 2, lx + 4 + k)] = v[3 * k + 1];  // (zero outside the image)
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (hq[u] >= 0) gt[hq[u]] = hv[u];
    __syncthreads();
    if (!live) return;
    if (y & 1) {  // wave-uniform per 16-lane row group; G2 sites sit on odd rows only
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float o = v[3 * k + 1];
        if (cfa_color(y, x + k, pattern) == 1) {
          // taps of column lx + 4 + k + d: class and word offset are compile-time per (k, d)
          const int rb = (ly + 2) * GLS + (lx >> 2);
          auto g = [&](int dy, int dx) { const int kk = 4 + k + dx; return gt[rb + dy * GLS + (kk >> 2) + (kk & 3) * GLQ]; };
          const float maximum = 1.0f;
          const float o1_1 = g(-1, -1), o1_2 = g(-1, 1), o1_3 = g(1, -1), o1_4 = g(1, 1);
          const float o2_1 = g(-2, 0), o2_2 = g(2, 0), o2_3 = g(0, -2), o2_4 = g(0, 2);
          const float m1 = (o1_1 + o1_2 + o1_3 + o1_4) / 4.0f;
          const float m2 = (o2_1 + o2_2 + o2_3 + o2_4) / 4.0f;
          if ((m2 > 0.0f) && (m1 > 0.0f) && (m1 / m2 < maximum * 2.0f)) {
            const float c1 = (fabsf(o1_1 - o1_2) + fabsf(o1_1 - o1_3) + fabsf(o1_1 - o1_4) + fabsf(o1_2 - o1_3) + fabsf(o1_3 - o1_4) + fabsf(o1_2 - o1_4)) / 6.0f;
            const float c2 = (fabsf(o2_1 - o2_2) + fabsf(o2_1 - o2_3) + fabsf(o2_1 - o2_4) + fabsf(o2_2 - o2_3) + fabsf(o2_3 - o2_4) + fabsf(o2_2 - o2_4)) / 6.0f;
            if ((o < maximum * 0.95f) && (c1 < maximum * threshold) && (c2 < maximum * threshold)) o *= m1 / m2;
          }
        }
        v[3 * k + 1] = o;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) v[3 * k + 1] = fmaxf(v[3 * k + 1], 0.0f);
    rgb4_io<TO>::store(out, p0 >> 2, v);
  } else {
    // unaligned images: the scalar form (green tile gathered, then one pixel per thread)
    for (int i = tid; i < (GLW + 4) * (GLH + 4); i += 256) {
      const int r = i / (GLW + 4), c = i - r * (GLW + 4);
      gt[gl_at(r, c + 2)] = green_at(x0 - 2 + c, y0 - 2 + r);
    }
    __syncthreads();
    const int lx = tid & 63;
    for (int ly = tid >> 6; ly < GLH; ly += 4) {
      const int x = x0 + lx, y = y0 + ly;
      if (x >= width || y >= height) continue;
      const size_t p0 = (size_t)y * width + x;
      float o = ld<TI>(in, p0 * 3 + 1);
      if (cfa_color(y, x, pattern) == 1 && (y & 1)) {
        auto g = [&](int dy, int dx) { return gt[gl_at(ly + 2 + dy, lx + 4 + dx)]; };
        const float maximum = 1.0f;
        const float o1_1 = g(-1, -1), o1_2 = g(-1, 1), o1_3 = g(1, -1), o1_4 = g(1, 1);
        const float o2_1 = g(-2, 0), o2_2 = g(2, 0), o2_3 = g(0, -2), o2_4 = g(0, 2);
        const float m1 = (o1_1 + o1_2 + o1_3 + o1_4) / 4.0f;
        const float m2 = (o2_1 + o2_2 + o2_3 + o2_4) / 4.0f;
        if ((m2 > 0.0f) && (m1 > 0.0f) && (m1 / m2 < maximum * 2.0f)) {
          const float c1 = (fabsf(o1_1 - o1_2) + fabsf(o1_1 - o1_3) + fabsf(o1_1 - o1_4) + fabsf(o1_2 - o1_3) + fabsf(o1_3 - o1_4) + fabsf(o1_2 - o1_4)) / 6.0f;
          const float c2 = (fabsf(o2_1 - o2_2) + fabsf(o2_1 - o2_3) + fabsf(o2_1 - o2_4) + fabsf(o2_2 - o2_3) + fabsf(o2_3 - o2_4) + fabsf(o2_2 - o2_4)) / 6.0f;
          if ((o < maximum * 0.95f) && (c1 < maximum * threshold) && (c2 < maximum * threshold)) o *= m1 / m2;
        }
      }
      st<TO>(out, p0 * 3, ld<TI>(in, p0 * 3));
      st<TO>(out, p0 * 3 + 1, fmaxf(o, 0.0f));
      st<TO>(out, p0 * 3 + 2, ld<TI>(in, p0 * 3 + 2));
    }
  }
}

// white_balance.cu:10-42
template <typename T>
__global__ __launch_bounds__(256) void white_balance_kernel(const T* __restrict__ in, T* __restrict__ out, const float* __restrict__ gains,
                                                            int width, int height, uint32_t pattern) {
  const float gr = gains[0], gg = gains[1], gb = gains[2];
  for (int y = blockIdx.y; y < height; y += gridDim.y)
    for (int x = blockIdx.x * 256 + threadIdx.x; x < width; x += gridDim.x * 256) {
      const int c = cfa_color(y, x, pattern);
      const float g = (c == 0) ? gr : (c == 2 ? gb : gg);
      const size_t i = (size_t)y * width + x;
      st<T>(out, i, clampf(ld<T>(in, i) * g, 0.0f, 1.0f));
    }
}

inline dim3 row_grid(int threads_per_row, int rows) { return dim3((unsigned)tdk_div_up(threads_per_row, 256), (unsigned)(rows < 32768 ? rows : 32768)); }

inline int stream_grid(int64_t nthreads) {
  int64_t b = tdk_div_up64(nthreads, 256);
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

namespace {
struct Img {
  void* p;
  int dtype;
  bool aligned4() const { return tdk_aligned(p, dtype == TDK_F16 ? 8 : 16); }
};
// TI / TO = the storage types of a stage's source and destination
#define PP_TYPES(src_, dst_, ...)                                                                            \
  do {                                                                                                       \
    if ((src_).dtype == TDK_F32 && (dst_).dtype == TDK_F32) { using TI = float; using TO = float; const TI* src = reinterpret_cast<const TI*>((src_).p); TO* dst = reinterpret_cast<TO*>((dst_).p); __VA_ARGS__; } \
    else if ((src_).dtype == TDK_F16 && (dst_).dtype == TDK_F32) { using TI = __half; using TO = float; const TI* src = reinterpret_cast<const TI*>((src_).p); TO* dst = reinterpret_cast<TO*>((dst_).p); __VA_ARGS__; } \
    else if ((src_).dtype == TDK_F32 && (dst_).dtype == TDK_F16) { using TI = float; using TO = __half; const TI* src = reinterpret_cast<const TI*>((src_).p); TO* dst = reinterpret_cast<TO*>((dst_).p); __VA_ARGS__; } \
    else { using TI = __half; using TO = __half; const TI* src = reinterpret_cast<const TI*>((src_).p); TO* dst = reinterpret_cast<TO*>((dst_).p); __VA_ARGS__; } \
  } while (0)

int stage_count(int color_smoothing_passes, int green_eq_local, int green_eq_global) {
  const int passes = color_smoothing_passes > 0 ? color_smoothing_passes : 0;
  return (passes + FMAXP - 1) / FMAXP + (green_eq_local ? 1 : 0) + (green_eq_global ? 1 : 0);  // <= FMAXP passes per launch
}
// fp32 images between the stages: one for fp32 callers (the other side of the ping-pong is the caller's output), up to two for
// binary16 callers (every intermediate stays fp32: the result is the fp32 result rounded ONCE)
int scratch_images(int stages, int dtype) { return dtype == TDK_F16 ? (stages >= 3 ? 2 : stages >= 2 ? 1 : 0) : (stages >= 2 ? 1 : 0); }
}  // namespace

TDK_EXPORT size_t tdk_postprocess_workspace_bytes_ex(int width, int height, int color_smoothing_passes, int green_eq_local, int green_eq_global, int dtype) {
  if (width <= 0 || height <= 0) return 0;
  const int stages = stage_count(color_smoothing_passes, green_eq_local, green_eq_global);
  size_t bytes = 256 + GEQ_BLOCKS * sizeof(float2);  // ratio + partial sums
  bytes += (size_t)scratch_images(stages, dtype) * tdk_align_up((size_t)width * height * 3 * sizeof(float), 256);
  return tdk_align_up(bytes, 256);
}

TDK_EXPORT size_t tdk_postprocess_workspace_bytes(int width, int height, int color_smoothing_passes, int green_eq_local, int green_eq_global) {
  return tdk_postprocess_workspace_bytes_ex(width, height, color_smoothing_passes, green_eq_local, green_eq_global, TDK_F32);
}

TDK_EXPORT int tdk_postprocess_ex(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, uint32_t pattern, int color_smoothing_passes,
                                  int green_eq_local, int green_eq_global, float green_eq_threshold, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(rgb_in && rgb_out && workspace, "tdk_postprocess: null pointer");
  TDK_REQUIRE(rgb_in != rgb_out, "tdk_postprocess: in and out must not alias");
  TDK_REQUIRE(width > 0 && height > 0, "tdk_postprocess: invalid size %dx%d", width, height);
  TDK_REQUIRE(dtype == TDK_F32 || dtype == TDK_F16, "unsupported dtype tag %d", dtype);
  hipStream_t s = tdk_stream(stream);
  const int passes = color_smoothing_passes > 0 ? color_smoothing_passes : 0;
  const int stages = stage_count(color_smoothing_passes, green_eq_local, green_eq_global);
  const size_t elem = dtype == TDK_F16 ? 2 : 4;
  if (stages == 0) {
    TDK_HIP_CALL(hipMemcpyAsync(rgb_out, rgb_in, (size_t)width * height * 3 * elem, hipMemcpyDeviceToDevice, s), "tdk_postprocess(copy)");
    return TDK_OK;
  }
  char* ws = reinterpret_cast<char*>(workspace);
  float* ratio = reinterpret_cast<float*>(ws);
  float2* partial = reinterpret_cast<float2*>(ws + 256);
  char* scratch0 = ws + 256 + GEQ_BLOCKS * sizeof(float2);
  char* scratch1 = scratch0 + tdk_align_up((size_t)width * height * 3 * sizeof(float), 256);
  const int64_t npix = (int64_t)width * height;
  const bool w4 = width % 4 == 0;

  const Img user_out{rgb_out, dtype};
  auto dst_of = [&](int i) -> Img {
    if (i == stages - 1) return user_out;
    if (dtype == TDK_F32) return ((stages - 1 - i) % 2 == 0) ? user_out : Img{scratch0, TDK_F32};  // ping-pong through the caller's output
    return Img{(i % 2 == 0) ? scratch0 : scratch1, TDK_F32};
  };
  Img src_img{const_cast<void*>(rgb_in), dtype};
  int stage = 0;

  for (int left = passes; left > 0; left -= FMAXP, stage++) {
    const Img dst_img = dst_of(stage);
    const int n = left < FMAXP ? left : FMAXP;
    PP_TYPES(src_img, dst_img,
             TDK_LAUNCH("tdk_postprocess(color_smoothing)", (smoothing_fused_kernel<TI, TO>), dim3(tdk_div_up(width, STW), tdk_div_up(height, STH)), dim3(SNT), 0, s,
                        src, dst, width, height, (int)(w4 && src_img.aligned4()), (int)(w4 && dst_img.aligned4()), n));
    src_img = dst_img;
  }
  if (green_eq_global) {
    const Img dst_img = dst_of(stage++);
    const int nb = (int)(tdk_div_up64(npix, 256) < GEQ_BLOCKS ? tdk_div_up64(npix, 256) : GEQ_BLOCKS);
    const bool vec_io = w4 && src_img.aligned4() && dst_img.aligned4();
    PP_TYPES(src_img, dst_img, {
      (void)dst;
      TDK_LAUNCH("tdk_postprocess(green_sums)", green_sums_kernel<TI>, dim3(nb), dim3(256), 0, s, src, width, height, pattern, partial);
    });
    TDK_LAUNCH("tdk_postprocess(green_ratio)", green_ratio_kernel, dim3(1), dim3(256), 0, s, partial, nb, ratio);
    PP_TYPES(src_img, dst_img, {
      if (vec_io) TDK_LAUNCH("tdk_postprocess(green_apply)", (green_apply_kernel<4, TI, TO>), row_grid(width / 4, height), dim3(256), 0, s, src, dst, width, height, pattern, ratio);
      else TDK_LAUNCH("tdk_postprocess(green_apply)", (green_apply_kernel<1, TI, TO>), row_grid(width, height), dim3(256), 0, s, src, dst, width, height, pattern, ratio);
    });
    src_img = dst_img;
  }
  if (green_eq_local) {
    const Img dst_img = dst_of(stage++);
    // postprocess.cu:383: threshold / 100. is evaluated in double and narrowed
    const float thr = (float)((double)green_eq_threshold / 100.0);
    const dim3 lgrid(tdk_div_up(width, GLW), tdk_div_up(height, GLH));
    const bool vec_io = w4 && src_img.aligned4() && dst_img.aligned4();
    PP_TYPES(src_img, dst_img, {
      if (vec_io) TDK_LAUNCH("tdk_postprocess(green_local)", (green_local_kernel<4, TI, TO>), lgrid, dim3(256), 0, s, src, dst, width, height, pattern, thr);
      else TDK_LAUNCH("tdk_postprocess(green_local)", (green_local_kernel<1, TI, TO>), lgrid, dim3(256), 0, s, src, dst, width, height, pattern, thr);
    });
    src_img = dst_img;
  }
  return TDK_OK;
}

TDK_EXPORT int tdk_postprocess(const float* rgb_in, float* rgb_out, void* workspace, int width, int height, uint32_t pattern,
                               int color_smoothing_passes, int green_eq_local, int green_eq_global, float green_eq_threshold,
                               tdk_stream_t stream) {
  return tdk_postprocess_ex(rgb_in, rgb_out, workspace, width, height, pattern, color_smoothing_passes, green_eq_local, green_eq_global, green_eq_threshold,
                            TDK_F32, stream);
}

TDK_EXPORT int tdk_apply_white_balance_ex(const void* bayer_in, void* bayer_out, const float* gains, int width, int height, uint32_t pattern, int dtype,
                                          tdk_stream_t stream) {
  TDK_REQUIRE(bayer_in && bayer_out && gains, "tdk_apply_white_balance: null pointer");
  TDK_REQUIRE(width > 0 && height > 0, "tdk_apply_white_balance: invalid size %dx%d", width, height);
  TDK_DISPATCH_DTYPE(dtype, T, TDK_LAUNCH("tdk_apply_white_balance", white_balance_kernel<T>, row_grid(width, height), dim3(256), 0, tdk_stream(stream),
                                          reinterpret_cast<const T*>(bayer_in), reinterpret_cast<T*>(bayer_out), gains, width, height, pattern));
  return TDK_OK;
}

TDK_EXPORT int tdk_apply_white_balance(const float* bayer_in, float* bayer_out, const float* gains, int width, int height, uint32_t pattern,
                                       tdk_stream_t stream) {
  return tdk_apply_white_balance_ex(bayer_in, bayer_out, gains, width, height, pattern, TDK_F32, stream);
}
