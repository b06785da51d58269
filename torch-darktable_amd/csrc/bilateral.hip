// bilateral.hip -- bilateral-grid local contrast on a luminance plane.
//
// Replaces reference csrc/local_contrast/bilateral.cu:30-404 (BilateralImpl::process: memset,
// splat with 8 global float atomics per pixel, three line-walking blur kernels, slice).
// Semantics kept: grid size from compute_grid_size (:273-299), sample coordinates with the RAW
// sigmas (:71-86), splat weight 1/sigma_s^2, [1 4 6 4 1]/16 along x then y and the
// [-2 -4 0 4 2]/16 derivative along z with zero extension (:132-204), slice
// max(0, L - detail * sigma_r * 4 * trilerp) (:208-228).
//
// MI355X design
//  * small sigma_s (<= 4, the pipeline default 2): the whole op is ONE tile kernel that keeps the grid in
//    LDS (bilateral_tile_kernel below); the four kernels described next serve larger sigma_s, where the
//    grid is small, and in-place calls.
//  * splat: MI355X resolves global float atomics at the memory side (~1.3 TB/s of added bytes
//    chip-wide) and LDS float atomics cost ~150 cycles per wave instruction, so the reference's
//    8 atomics/px scatter is turned into a gather: one thread per grid column sums the pixels
//    that touch it (see splat_gather_kernel).  No atomics, no memset, deterministic.
//  * blur x+y: one kernel; a 64 x 16 cell tile + 2-cell halo of one z-slice is staged in LDS,
//    blurred along x into a second LDS buffer and along y on the way out (coalesced along x;
//    the reference walks each line serially in one thread, uncoalesced for x).
//  * z derivative: one thread per (x, y) column, coalesced along x, register window over z.
//  * slice: trilinear gather from the (L2/MALL-resident) grid, fp32 or fp16 planes.
//  The zero-extended stencil forms below are bit-identical to the reference's edge-case
//  formulas (x + 0 == x); and the gather sums in raster order, so the whole op is bit-identical to the oracle.
#include <stdlib.h>

#include "tdk_color.h"

#ifdef TDK_BIL_TIMING
// experiments: clock64() deltas per phase of one workgroup of the tile kernel (profiles/bilateral_phase_exp.py)
__device__ unsigned long long g_bil_phase_cycles[16];
#define BIL_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x == 1000) { const unsigned long long t_ = clock64(); atomicAdd(&g_bil_phase_cycles[k], t_ - bil_t0); bil_t0 = t_; } } while (0)
extern "C" __attribute__((visibility("default"))) int tdk_debug_bilateral_phase_cycles(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bil_phase_cycles), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_bil_phase_cycles), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
#define BIL_MARK(k)
#endif

namespace {

struct GridDims {
  int sx, sy, sz;
};

// bilateral.cu:273-299 (host float math)
GridDims compute_grid_size(int width, int height, float sigma_s, float sigma_r) {
  float ss = sigma_s;
  if (ss < 0.5f) ss = 0.5f;
  const float L_range = 1.0f;
  auto clampf_h = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
  const float gx = clampf_h(roundf((float)width / ss), 4.0f, 3000.0f);
  const float gy = clampf_h(roundf((float)height / ss), 4.0f, 3000.0f);
  const float gz = clampf_h(roundf(L_range / sigma_r), 4.0f, 50.0f);
  const float s_s = fmaxf((float)height / gy, (float)width / gx);
  const float s_r = L_range / gz;
  GridDims d;
  d.sx = (int)ceilf((float)width / s_s) + 1;
  d.sy = (int)ceilf((float)height / s_s) + 1;
  d.sz = (int)ceilf(L_range / s_r) + 1;
  return d;
}

struct Sample {
  int ix, iy, iz;
  float fx, fy, fz;
};

// bilateral.cu:71-86
__device__ __forceinline__ Sample make_sample(int x, int y, float L, GridDims d, float sigma_s, float sigma_r) {
  const float gx = clampf((float)x / sigma_s, 0.0f, (float)(d.sx - 1));
  const float gy = clampf((float)y / sigma_s, 0.0f, (float)(d.sy - 1));
  const float gz = clampf(L / sigma_r, 0.0f, (float)(d.sz - 1));
  Sample s;
  s.ix = min((int)gx, d.sx - 2);
  s.iy = min((int)gy, d.sy - 2);
  s.iz = min((int)gz, d.sz - 2);
  s.fx = gx - (float)s.ix;
  s.fy = gy - (float)s.iy;
  s.fz = gz - (float)s.iz;
  return s;
}

// ---- splat as a GATHER: one thread per grid column (cx, cy) walks the pixels whose trilinear
// footprint touches it, in raster order, and accumulates its sz cells in a private LDS column
// (plain read-modify-write by the owning thread).  No atomics at all -- LDS float atomics cost
// ~150 cycles per wave instruction and global ones are capped at ~1.3 TB/s on MI355X -- and the
// float sums come out in the same order as a sequential raster-order splat, i.e. bit-identical
// to the oracle and identical run to run (the reference's atomic order is not).
// A pixel contributes to column cx when its base cell is cx (weight 1 - f) or cx - 1 (weight f).
__device__ __forceinline__ float axis_weight(int p, float sigma_s, int size, int cell) {
  const float g = clampf((float)p / sigma_s, 0.0f, (float)(size - 1));
  const int ib = min((int)g, size - 2);
  const float f = g - (float)ib;
  return (ib == cell) ? (1.0f - f) : ((ib == cell - 1) ? f : -1.0f);  // -1: no contribution
}

// MAXC = compile-time bound on the candidate window per axis (2 sigma_s + 5 pixels); columns whose
// window is longer (large sigma_s, or the last column that also collects every clamped pixel)
// take the generic loop.  Zero-weight pixels are skipped: every contribution is >= +0 and the
// sums start at +0, so dropping exact zeros cannot change a bit.
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void splat_gather_kernel(const T* __restrict__ lum, float* __restrict__ grid, int width, int height, GridDims d,
                                                           float sigma_s, float sigma_r) {
  extern __shared__ float colacc[];  // [sz][256]
  const int cx = blockIdx.x * 32 + (threadIdx.x & 31), cy = blockIdx.y * 8 + (threadIdx.x >> 5);
  const int tid = threadIdx.x;
  for (int z = 0; z < d.sz; z++) colacc[z * 256 + tid] = 0.0f;
  if (cx >= d.sx || cy >= d.sy) return;
  const float contrib = 1.0f / (sigma_s * sigma_s);
  // candidate pixel window; the last column also collects every pixel clamped onto it
  const int x_lo = max(0, (int)floorf(sigma_s * (float)(cx - 1)) - 1);
  const int x_hi = (cx == d.sx - 1) ? width - 1 : min(width - 1, (int)ceilf(sigma_s * (float)(cx + 1)) + 1);
  const int y_lo = max(0, (int)floorf(sigma_s * (float)(cy - 1)) - 1);
  const int y_hi = (cy == d.sy - 1) ? height - 1 : min(height - 1, (int)ceilf(sigma_s * (float)(cy + 1)) + 1);

  auto deposit = [&](int x, int y, float wxy) {
    const float L = ld(lum, (size_t)y * width + x);
    const float gz = clampf(L / sigma_r, 0.0f, (float)(d.sz - 1));
    const int iz = min((int)gz, d.sz - 2);
    const float fz = gz - (float)iz;
    colacc[iz * 256 + tid] += wxy * (1.0f - fz) * contrib;
    colacc[(iz + 1) * 256 + tid] += wxy * fz * contrib;
  };

  if (x_hi - x_lo + 1 <= MAXC) {
    float wxs[MAXC];  // x weights are the same for every row: computed once, statically indexed
#pragma unroll
    for (int k = 0; k < MAXC; k++) wxs[k] = (x_lo + k <= x_hi) ? axis_weight(x_lo + k, sigma_s, d.sx, cx) : -1.0f;
    for (int y = y_lo; y <= y_hi; y++) {
      const float wy = axis_weight(y, sigma_s, d.sy, cy);
      if (!(wy > 0.0f)) continue;
#pragma unroll
      for (int k = 0; k < MAXC; k++)
        if (wxs[k] > 0.0f) deposit(x_lo + k, y, wxs[k] * wy);
    }
  } else {
    for (int y = y_lo; y <= y_hi; y++) {
      const float wy = axis_weight(y, sigma_s, d.sy, cy);
      if (!(wy > 0.0f)) continue;
      for (int x = x_lo; x <= x_hi; x++) {
        const float wx = axis_weight(x, sigma_s, d.sx, cx);
        if (wx > 0.0f) deposit(x, y, wx * wy);
      }
    }
  }
  const size_t plane = (size_t)d.sx * d.sy;
  float* g = grid + (size_t)cy * d.sx + cx;
  for (int z = 0; z < d.sz; z++) g[z * plane] = colacc[z * 256 + tid];
}

// ---- blur along x then y for one z-slice tile
constexpr int BTW = 64, BTH = 16;
constexpr int BLW = BTW + 4, BLH = BTH + 4;

__global__ __launch_bounds__(256) void blur_xy_kernel(const float* __restrict__ gin, float* __restrict__ gout, GridDims d) {
  __shared__ float a[BLH][BLW + 1];
  __shared__ float b[BLH][BTW + 1];
  const int x0 = blockIdx.x * BTW, y0 = blockIdx.y * BTH, z = blockIdx.z;
  const float* src = gin + (size_t)z * d.sx * d.sy;
  float* dst = gout + (size_t)z * d.sx * d.sy;
  for (int i = threadIdx.x; i < BLW * BLH; i += 256) {
    const int r = i / BLW, c = i - r * BLW;
    const int gx = x0 - 2 + c, gy = y0 - 2 + r;
    a[r][c] = (gx >= 0 && gy >= 0 && gx < d.sx && gy < d.sy) ? src[(size_t)gy * d.sx + gx] : 0.0f;
  }
  __syncthreads();
  const float w0 = 6.0f / 16.0f, w1 = 4.0f / 16.0f, w2 = 1.0f / 16.0f;
  // x pass on all BLH rows (rows outside the grid are zero and stay zero)
  for (int i = threadIdx.x; i < BTW * BLH; i += 256) {
    const int r = i / BTW, c = i - r * BTW;
    const float* p = &a[r][c + 2];
    // outside the grid along y the reference has no data: keep exact zeros
    const int gy = y0 - 2 + r;
    b[r][c] = (gy >= 0 && gy < d.sy) ? (p[0] * w0 + w1 * (p[1] + p[-1]) + w2 * (p[2] + p[-2])) : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BTW * BTH; i += 256) {
    const int r = i / BTW, c = i - r * BTW;
    const int gx = x0 + c, gy = y0 + r;
    if (gx < d.sx && gy < d.sy)
      dst[(size_t)gy * d.sx + gx] = b[r + 2][c] * w0 + w1 * (b[r + 3][c] + b[r + 1][c]) + w2 * (b[r + 4][c] + b[r][c]);
  }
}

// ---- derivative along z (bilateral.cu:171-204)
__global__ __launch_bounds__(256) void blur_z_kernel(const float* __restrict__ gin, float* __restrict__ gout, GridDims d) {
  const size_t plane = (size_t)d.sx * d.sy;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane) return;
  const float w1 = 4.0f / 16.0f, w2 = 2.0f / 16.0f;
  float m2 = 0.0f, m1 = 0.0f;                       // x[z-2], x[z-1]
  float c0 = gin[i];                                 // x[z]
  float p1 = (d.sz > 1) ? gin[i + plane] : 0.0f;     // x[z+1]
  for (int z = 0; z < d.sz; z++) {
    const float p2 = (z + 2 < d.sz) ? gin[i + (size_t)(z + 2) * plane] : 0.0f;
    gout[i + (size_t)z * plane] = w1 * (p1 - m1) + w2 * (p2 - m2);
    m2 = m1; m1 = c0; c0 = p1; p1 = p2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void slice_kernel(const T* __restrict__ lum, const float* __restrict__ grid, T* __restrict__ out, int width,
                                                    int height, GridDims d, float sigma_s, float sigma_r, float detail) {
  const int64_t n = (int64_t)width * height;
  const float norm = -detail * sigma_r * 4.0f;
  const size_t oy = d.sx, oz = (size_t)d.sx * d.sy;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / width), x = (int)(i - (int64_t)y * width);
    const float L = ld(lum, (size_t)i);
    const Sample s = make_sample(x, y, L, d, sigma_s, sigma_r);
    const float ax = 1.0f - s.fx, ay = 1.0f - s.fy, az = 1.0f - s.fz, bx = s.fx, by = s.fy, bz = s.fz;
    const float* g = grid + s.ix + oy * s.iy + oz * s.iz;
    const float Ldiff = g[0] * ax * ay * az + g[1] * bx * ay * az + g[oy] * ax * by * az + g[oy + 1] * bx * by * az + g[oz] * ax * ay * bz +
                        g[oz + 1] * bx * ay * bz + g[oz + oy] * ax * by * bz + g[oz + oy + 1] * bx * by * bz;
    st(out, (size_t)i, fmaxf(0.0f, L + norm * Ldiff));
  }
}

// Fused epilogue of Bilateral.process_rgb / process_log_rgb (reference local_contrast.py:109-125):
// slice the grid at the pixel's (fp32) luminance and put the new lightness straight back into the
// RGB pixel, so the filtered luminance plane is never written to HBM.
template <typename T, bool LOG, int VEC>
__global__ __launch_bounds__(256) void slice_modify_kernel(const float* __restrict__ lum, const float* __restrict__ grid, const T* __restrict__ rgb,
                                                           T* __restrict__ out, int width, int height, GridDims d, float sigma_s, float sigma_r,
                                                           float detail) {
  const int64_t n = (int64_t)width * height / VEC;  // VEC == 4 requires width % 4 == 0
  const float norm = -detail * sigma_r * 4.0f;
  const size_t oy = d.sx, oz = (size_t)d.sx * d.sy;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < n; gi += (int64_t)gridDim.x * 256) {
    const int64_t i0 = gi * VEC;
    const int y = (int)(i0 / width), x0 = (int)(i0 - (int64_t)y * width);
    float v[3 * VEC], Lv[VEC];
    if constexpr (VEC == 4) {
      rgb4_io<T>::load(rgb, (size_t)gi, v);
      s4_io<float>::load(lum, (size_t)gi, Lv);
    } else {
      v[0] = ld(rgb, (size_t)i0 * 3); v[1] = ld(rgb, (size_t)i0 * 3 + 1); v[2] = ld(rgb, (size_t)i0 * 3 + 2);
      Lv[0] = lum[i0];
    }
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      const float L = Lv[k];
      const Sample s = make_sample(x0 + k, y, L, d, sigma_s, sigma_r);
      const float ax = 1.0f - s.fx, ay = 1.0f - s.fy, az = 1.0f - s.fz, bx = s.fx, by = s.fy, bz = s.fz;
      const float* g = grid + s.ix + oy * s.iy + oz * s.iz;
      const float Ldiff = g[0] * ax * ay * az + g[1] * bx * ay * az + g[oy] * ax * by * az + g[oy + 1] * bx * by * az + g[oz] * ax * ay * bz +
                          g[oz + 1] * bx * ay * bz + g[oz + oy] * ax * by * bz + g[oz + oy + 1] * bx * by * bz;
      const float Lnew = fmaxf(0.0f, L + norm * Ldiff);
      const f3 c = mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2]);
      const f3 r = LOG ? cA::modify_log_luminance(c, Lnew) : cA::modify_luminance(c, Lnew);
      v[3 * k] = r.x; v[3 * k + 1] = r.y; v[3 * k + 2] = r.z;
    }
    if constexpr (VEC == 4) rgb4_io<T>::store(out, (size_t)gi, v);
    else { st(out, (size_t)i0 * 3, v[0]); st(out, (size_t)i0 * 3 + 1, v[1]); st(out, (size_t)i0 * 3 + 2, v[2]); }
  }
}

// The same for the Lab hand-over chain: (L, a, b) in, RGB out (MODE 3 of the tile kernel below; this one serves large sigma_s).
template <typename T>
__global__ __launch_bounds__(256) void slice_lab_kernel(const float* __restrict__ lum, const float* __restrict__ grid, const float* __restrict__ ab,
                                                        T* __restrict__ out, int width, int height, GridDims d, float sigma_s, float sigma_r, float detail) {
  const int64_t n = (int64_t)width * height;
  const float norm = -detail * sigma_r * 4.0f;
  const size_t oy = d.sx, oz = (size_t)d.sx * d.sy;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / width), x = (int)(i - (int64_t)y * width);
    const float L = lum[i];
    const Sample s = make_sample(x, y, L, d, sigma_s, sigma_r);
    const float ax = 1.0f - s.fx, ay = 1.0f - s.fy, az = 1.0f - s.fz, bx = s.fx, by = s.fy, bz = s.fz;
    const float* g = grid + s.ix + oy * s.iy + oz * s.iz;
    const float Ldiff = g[0] * ax * ay * az + g[1] * bx * ay * az + g[oy] * ax * by * az + g[oy + 1] * bx * by * az + g[oz] * ax * ay * bz +
                        g[oz + 1] * bx * ay * bz + g[oz + oy] * ax * by * bz + g[oz + oy + 1] * bx * by * bz;
    const float Lnew = fmaxf(0.0f, L + norm * Ldiff);
    const f3 r = clip3(cA::lab_to_rgb(mk3(fmaxf(0.0f, fminf(1.0f, Lnew)), ab[2 * i], ab[2 * i + 1])));
    st(out, (size_t)i * 3, r.x); st(out, (size_t)i * 3 + 1, r.y); st(out, (size_t)i * 3 + 2, r.z);
  }
}

// ---- the whole op in one tile kernel (small sigma_s) -------------------------------------------
// For small sigma_s the grid is as large as the image (12 MP, sigma_s 2, sigma_r 0.2: 2049 x 1537 x 6
// floats = 76 MB) and the four-kernel path moves it through HBM five times.  Here one workgroup owns
// a FTW x FTH pixel tile: it builds, in LDS, exactly the grid cells its pixels slice from (+2 cells
// of blur halo), from the luminance of the tile (+ the pixels that splat into those cells), runs the
// three blurs in LDS and slices straight into the output.  The grid never exists in HBM.
// Every cell value is computed by the same expressions, in the same order, as in the kernels above
// (raster-order gather, x then y then z), so the result is bit-identical to the four-kernel path.
// Requires that no pixel is clamped onto the last grid column/row (host-checked): then a cell only
// collects pixels within sigma_s of it and the tile's pixel halo is bounded.
//
// Everything that depends on the tile COLUMN only or on the tile ROW only (the cell / pixel ranges, the sample
// coordinates x / sigma_s, and for every cell the run of pixels that splat into it with their weights) is computed
// once per launch by bilateral_axis_tables_kernel into a small table in the workspace; a tile workgroup copies its
// two records into LDS.  (Rounds 1-2 recomputed them in every workgroup: a serial chain of IEEE divisions,
// floor / ceil and scalar moves that all 8 waves executed, 18 % of a workgroup's life and 15 % of its instructions.)
constexpr int FTW = 64, FTH = 32, FNT = 512;
constexpr int TAB_HDR = 4;  // words: c_lo, nc, nmax, 1 spare
constexpr int TAB_W = 8;    // weights kept per cell: the pixels with a positive weight on a cell span < 2 sigma_s <= 8 (sigma_s <= 4)

struct AxisTile {
  int c_lo;  // first cell kept in LDS (2 below the first cell sliced; may be negative = outside the grid)
  int nc;    // cells kept
  int p_lo;  // first pixel whose splat can reach those cells
  int np;    // pixels
};

// Shared by host (LDS sizing) and device: plain IEEE float ops, so both sides agree exactly.
__host__ __device__ inline AxisTile axis_tile(int p0, int tile, int size_px, int size_cells, float sigma_s) {
  const int p1 = ((p0 + tile < size_px) ? p0 + tile : size_px) - 1;
  const float top = (float)(size_cells - 1);
  const float g0 = fminf(fmaxf((float)p0 / sigma_s, 0.0f), top), g1 = fminf(fmaxf((float)p1 / sigma_s, 0.0f), top);
  const int c0 = ((int)g0 < size_cells - 2) ? (int)g0 : size_cells - 2;
  const int c1 = (((int)g1 < size_cells - 2) ? (int)g1 : size_cells - 2) + 1;
  AxisTile t;
  t.c_lo = c0 - 2;
  t.nc = c1 - c0 + 5;
  const int lo = (int)floorf(sigma_s * (float)(t.c_lo - 1)) - 1;
  const int hi = (int)ceilf(sigma_s * (float)(c1 + 3)) + 1;
  t.p_lo = lo > 0 ? lo : 0;
  t.np = ((hi < size_px - 1) ? hi : size_px - 1) - t.p_lo + 1;
  return t;
}

struct TileLds {
  int rs;      // LDS grid row stride in cells (odd: a wave walking rows or columns never bank-conflicts)
  int plane;   // floats per z-slice of the LDS grid (>= rs * max nc_y, multiple of 64: bank == column)
  int usize;   // floats in the sample-tile / blur-temp union
  int lw, lh;  // pixels per axis of the sample window of a tile (lw a multiple of 4): the LDS sample tile is lh rows of lw
  int ncx, ncy;  // max cells per axis
  int hx, hy;  // the sample window of the tile at (x0, y0) starts at max(0, x0 - hx), max(0, y0 - hy); hx a multiple of 4
  // launch constants the tile kernel would otherwise compute with IEEE divisions in every wave
  float inv_lw, inv_qw, inv_rs, inv_ncy;  // 1 / lw, 1 / (lw / 4), 1 / rs, 1 / ncy (for fast_div)
  float rc_r, contrib, norm;              // 1 / sigma_r, 1 / sigma_s^2, -detail * sigma_r * 4
};

// first pixel and pixel count of a tile's sample window along one axis
__host__ __device__ inline int win_lo(int p0, int halo) { return p0 > halo ? p0 - halo : 0; }
__host__ __device__ inline int win_np(int lo, int lp, int size_px) { return lo + lp < size_px ? lp : size_px - lo; }

// words of one axis record: header, sample coordinates of the lp pixels, then per cell the first pixel of its run
// (relative to the window start; -1: cell outside the grid), then TAB_W weights per cell, k-major
__host__ __device__ inline int tab_rec(int lp, int nc) { return TAB_HDR + lp + nc * (1 + TAB_W); }

// One 64-thread workgroup per tile column, then per tile row.  For a cell, the pixels with a positive weight
// (axis_weight above) are consecutive; the record keeps the first one and nmax weights, nmax = the longest run of
// the record, shorter runs padded with zero weights (a zero weight adds +0 to a non-negative sum: no bit changes)
// and shifted down where the padding would leave the pixel window.
__global__ __launch_bounds__(64) void bilateral_axis_tables_kernel(int* __restrict__ tab, int width, int height, GridDims d, float sigma_s, int tiles_x,
                                                                  TileLds L) {
  __shared__ int s_nmax;
  const bool is_x = (int)blockIdx.x < tiles_x;
  const int ti = is_x ? (int)blockIdx.x : (int)blockIdx.x - tiles_x;
  const int size_px = is_x ? width : height, size_cells = is_x ? d.sx : d.sy, tile = is_x ? FTW : FTH;
  const int lp = is_x ? L.lw : L.lh, ncm = is_x ? L.ncx : L.ncy, halo = is_x ? L.hx : L.hy;
  int* rec = tab + (is_x ? ti * tab_rec(L.lw, L.ncx) : tiles_x * tab_rec(L.lw, L.ncx) + ti * tab_rec(L.lh, L.ncy));
  const AxisTile t = axis_tile(ti * tile, tile, size_px, size_cells, sigma_s);
  const int p_lo = win_lo(ti * tile, halo), np = win_np(p_lo, lp, size_px);  // contains [t.p_lo, t.p_lo + t.np) (plan_tiles)
  float* gs = reinterpret_cast<float*>(rec + TAB_HDR);
  int* start = rec + TAB_HDR + lp;
  float* wt = reinterpret_cast<float*>(start + ncm);
  const int lane = threadIdx.x;
  if (lane == 0) s_nmax = 1;
  __syncthreads();
  for (int i = lane; i < np; i += 64) gs[i] = clampf((float)(p_lo + i) / sigma_s, 0.0f, (float)(size_cells - 1));  // make_sample's gx / gy
  for (int l = lane; l < t.nc; l += 64) {
    const int cell = t.c_lo + l;
    int first = -1;
    if (cell >= 0 && cell < size_cells) {
      const int a = max(p_lo, (int)floorf(sigma_s * (float)(cell - 1)) - 1);
      const int b = min(p_lo + np - 1, (int)ceilf(sigma_s * (float)(cell + 1)) + 1);
      int k0 = -1, k1 = -1;
      for (int p = a; p <= b; p++)
        if (axis_weight(p, sigma_s, size_cells, cell) > 0.0f) { if (k0 < 0) k0 = p; k1 = p; }
      first = k0 < 0 ? a : k0;
      if (k0 >= 0) atomicMax(&s_nmax, k1 - k0 + 1);
    }
    start[l] = first;
  }
  __syncthreads();
  const int nmax = min(s_nmax, min(TAB_W, np));
  for (int l = lane; l < t.nc; l += 64) {
    const int cell = t.c_lo + l;
    int first = start[l];
    if (first >= 0) {
      first = min(first, p_lo + np - nmax);
      start[l] = first - p_lo;
    }
    for (int k = 0; k < TAB_W; k++) {
      float w = 0.0f;
      if (first >= 0 && k < nmax) w = fmaxf(axis_weight(first + k, sigma_s, size_cells, cell), 0.0f);
      wt[k * ncm + l] = w;  // k-major: the lanes of a wave (consecutive cells) read consecutive words
    }
  }
  if (lane == 0) { rec[0] = t.c_lo; rec[1] = t.nc; rec[2] = nmax; }
}

// i / n for 0 <= i < 2^20, 0 < n < 2^10 with inv = 1.0f / n: (i + 0.5) / n is at least 0.5 / n away
// from an integer, far more than the float rounding error, so the truncation is exact.
__device__ __forceinline__ int fast_div(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

// workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The tile kernel itself lives in tdk_bilateral_tile.h and is compiled twice: bt_exact (the oracle's bits: MODE 0 / 1 / 2) and bt_fast
// (contraction + factored slice for the Lab hand-over chain, MODE 3).
#define TDK_BT_FAST 0
namespace bt_exact {
#include "tdk_bilateral_tile.h"
}  // namespace bt_exact
#undef TDK_BT_FAST
#define TDK_BT_FAST 1
#pragma clang fp contract(fast)
namespace bt_fast {
#include "tdk_bilateral_tile.h"
}  // namespace bt_fast
#pragma clang fp contract(off)
#undef TDK_BT_FAST

constexpr size_t FUSED_LDS_LIMIT = 80 * 1024;  // two workgroups per CU

// Decide whether the tile kernel applies and size its LDS.  Returns false -> four-kernel path.
// sigma_s <= 4: the pixels with a positive weight on one cell lie within sigma_s of it on either side, at most 7 <= TAB_W.
static bool plan_tiles(int width, int height, const GridDims& d, float sigma_s, float sigma_r, float detail, TileLds* L, size_t* lds_bytes) {
  if (!(sigma_s >= 1.0f && sigma_s <= 4.0f)) return false;
  // no pixel may be clamped onto the last column / row (those columns collect far-away pixels)
  if ((float)(width - 1) / sigma_s > (float)(d.sx - 1) || (float)(height - 1) / sigma_s > (float)(d.sy - 1)) return false;
  // One sample-window shape for all tiles: it starts hx (hy) pixels before the tile and is lw (lh) pixels long, the
  // smallest values that contain the pixels axis_tile() asks for in every tile column (row); hx and lw in whole
  // 4-pixel groups so that a window row loads as 16-B groups.
  int ncx = 0, ncy = 0, hx = 0, hy = 0, lw = 0, lh = 0;
  for (int x0 = 0; x0 < width; x0 += FTW) {
    const AxisTile t = axis_tile(x0, FTW, width, d.sx, sigma_s);
    ncx = t.nc > ncx ? t.nc : ncx; hx = x0 - t.p_lo > hx ? x0 - t.p_lo : hx;
  }
  hx = (hx + 3) & ~3;
  for (int x0 = 0; x0 < width; x0 += FTW) {
    const AxisTile t = axis_tile(x0, FTW, width, d.sx, sigma_s);
    const int n = t.p_lo + t.np - win_lo(x0, hx);
    lw = n > lw ? n : lw;
  }
  lw = (lw + 3) & ~3;
  for (int y0 = 0; y0 < height; y0 += FTH) {
    const AxisTile t = axis_tile(y0, FTH, height, d.sy, sigma_s);
    ncy = t.nc > ncy ? t.nc : ncy; hy = y0 - t.p_lo > hy ? y0 - t.p_lo : hy;
  }
  for (int y0 = 0; y0 < height; y0 += FTH) {
    const AxisTile t = axis_tile(y0, FTH, height, d.sy, sigma_s);
    const int n = t.p_lo + t.np - win_lo(y0, hy);
    lh = n > lh ? n : lh;
  }
  L->rs = ncx | 1;
  L->plane = (int)tdk_align_up((size_t)L->rs * ncy, 64);
  const int lt = lw * lh + (ncx + ncy) * (1 + TAB_W), bt = d.sz * L->plane;  // sample tile + the two table records | blur temp
  L->usize = (int)tdk_align_up((size_t)(lt > bt ? lt : bt), 64);
  L->lw = lw; L->lh = lh;
  L->ncx = ncx; L->ncy = ncy;
  L->hx = hx; L->hy = hy;
  L->inv_lw = 1.0f / (float)lw; L->inv_qw = 1.0f / (float)(lw / 4); L->inv_rs = 1.0f / (float)L->rs; L->inv_ncy = 1.0f / (float)ncy;
  L->rc_r = 1.0f / sigma_r; L->contrib = 1.0f / (sigma_s * sigma_s); L->norm = -detail * sigma_r * 4.0f;
  *lds_bytes = ((size_t)d.sz * L->plane + L->usize + (size_t)lw + (size_t)lh) * sizeof(float);
  if (lw > 2 * FNT || lh > 2 * FNT || ncx * (1 + TAB_W) > 2 * FNT || ncy * (1 + TAB_W) > 2 * FNT) return false;  // record copies of the tile kernel
  return *lds_bytes <= FUSED_LDS_LIMIT;
}

// bytes of the axis tables (0 when the tile kernel does not apply for this geometry); they are the FIRST thing in a
// workspace of either layout, so one tdk_bilateral_prepare serves the plane and the RGB entry points alike
static size_t tile_table_bytes(int width, int height, float sigma_s, float sigma_r) {
  const GridDims d = compute_grid_size(width, height, sigma_s, sigma_r);
  TileLds L;
  size_t lds_bytes = 0;
  if (!plan_tiles(width, height, d, sigma_s, sigma_r, 0.0f, &L, &lds_bytes)) return 0;
  const size_t words = (size_t)tdk_div_up(width, FTW) * tab_rec(L.lw, L.ncx) + (size_t)tdk_div_up(height, FTH) * tab_rec(L.lh, L.ncy);
  return tdk_align_up(words * sizeof(int), 256);
}

static int build_tables(int* tab, int width, int height, const GridDims& d, float sigma_s, const TileLds& L, hipStream_t s) {
  const int tiles_x = tdk_div_up(width, FTW), tiles_y = tdk_div_up(height, FTH);
  TDK_LAUNCH("tdk_bilateral(tables)", bilateral_axis_tables_kernel, dim3(tiles_x + tiles_y), dim3(64), 0, s, tab, width, height, d, sigma_s, tiles_x, L);
  return TDK_OK;
}

template <typename TL, typename T, int MODE>
int launch_tiles(const TL* lum, const T* rgb, T* out, int* tab, int width, int height, const GridDims& d, float sigma_s, float sigma_r, float detail,
                 const TileLds& L, size_t lds_bytes, bool vec, bool prepared, hipStream_t s) {
  const int tiles_x = tdk_div_up(width, FTW), tiles_y = tdk_div_up(height, FTH), ntiles = tiles_x * tiles_y;
  const dim3 grid(8 * tdk_div_up(ntiles, 8));
#ifdef TDK_EXPERIMENTS
  if (const char* e = getenv("TDK_BIL_LDS_PAD")) lds_bytes += (size_t)atoi(e);  // fewer resident workgroups per CU (co-residency experiment)
#endif
  if (!prepared) {  // the tables depend on the geometry and the sigmas only: a caller that keeps its workspace builds them once (tdk_bilateral_prepare)
    const int rc = build_tables(tab, width, height, d, sigma_s, L, s);
    if (rc != TDK_OK) return rc;
  }
#define TDK_BT(NS, VECV)                                                                                                                          \
  do {                                                                                                                                            \
    TDK_MAX_LDS_ONCE((NS::bilateral_tile_kernel<TL, T, MODE, VECV>), "tdk_bilateral(hipFuncSetAttribute)");                                       \
    TDK_LAUNCH("tdk_bilateral(tiles)", (NS::bilateral_tile_kernel<TL, T, MODE, VECV>), grid, dim3(FNT), lds_bytes, s, lum, rgb, out, tab, width, height, d, \
               sigma_r, tiles_x, ntiles, L);                                                                                                      \
  } while (0)
  if constexpr (MODE == 3) { if (vec) TDK_BT(bt_fast, 4); else TDK_BT(bt_fast, 1); }
  else { if (vec) TDK_BT(bt_exact, 4); else TDK_BT(bt_exact, 1); }
#undef TDK_BT
  return TDK_OK;
}

// workspace layout: axis tables of the tile kernel | grid | tmp | [fp32 luminance plane (rgb entry points)]
static size_t grid_floats(const GridDims& d) { return tdk_align_up((size_t)d.sx * d.sy * d.sz, 64); }
static size_t plane_workspace_bytes(const GridDims& d, size_t tables) { return tables + tdk_align_up(2 * grid_floats(d) * sizeof(float), 256); }
static size_t rgb_workspace_bytes(const GridDims& d, int width, int height, size_t tables) {
  return tables + tdk_align_up((2 * grid_floats(d) + (size_t)width * height) * sizeof(float), 256);
}

// splat -> blur x,y -> z derivative; leaves the final grid in `grid`
template <typename T>
int build_grid(const T* in, float* grid, float* tmp, int width, int height, const GridDims& d, float sigma_s, float sigma_r, hipStream_t s) {
  const size_t splat_lds = (size_t)d.sz * 256 * sizeof(float);
  const dim3 sgrid(tdk_div_up(d.sx, 32), tdk_div_up(d.sy, 8));
  if (2.0f * sigma_s + 5.0f <= 10.0f) {
    TDK_MAX_LDS_ONCE((splat_gather_kernel<T, 10>), "tdk_bilateral(hipFuncSetAttribute)");
    TDK_LAUNCH("tdk_bilateral(splat)", (splat_gather_kernel<T, 10>), sgrid, dim3(256), splat_lds, s, in, grid, width, height, d, sigma_s, sigma_r);
  } else {
    TDK_MAX_LDS_ONCE((splat_gather_kernel<T, 24>), "tdk_bilateral(hipFuncSetAttribute)");
    TDK_LAUNCH("tdk_bilateral(splat)", (splat_gather_kernel<T, 24>), sgrid, dim3(256), splat_lds, s, in, grid, width, height, d, sigma_s, sigma_r);
  }
  TDK_LAUNCH("tdk_bilateral(blur_xy)", blur_xy_kernel, dim3(tdk_div_up(d.sx, BTW), tdk_div_up(d.sy, BTH), d.sz), dim3(256), 0, s, grid, tmp, d);
  TDK_LAUNCH("tdk_bilateral(blur_z)", blur_z_kernel, dim3((unsigned)tdk_div_up64((int64_t)d.sx * d.sy, 256)), dim3(256), 0, s, tmp, grid, d);
  return TDK_OK;
}

inline unsigned stream_blocks(int64_t npix) {
  int64_t b = tdk_div_up64(npix, 256);
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

template <typename T>
int launch(const void* lum_in, void* lum_out, void* workspace, int width, int height, float sigma_s, float sigma_r, float detail, unsigned flags, hipStream_t s) {
  const GridDims d = compute_grid_size(width, height, sigma_s, sigma_r);
  const size_t tables = tile_table_bytes(width, height, sigma_s, sigma_r);
  int* tab = reinterpret_cast<int*>(workspace);
  float* grid = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + tables);
  float* tmp = grid + grid_floats(d);
  const T* in = reinterpret_cast<const T*>(lum_in);
  TileLds L;
  size_t lds_bytes = 0;
  const bool tiles = !(flags & TDK_BILATERAL_GENERAL_PATH) && lum_in != lum_out  // tiles read a pixel halo: not in place
                     && plan_tiles(width, height, d, sigma_s, sigma_r, detail, &L, &lds_bytes);
  if (tiles) {
    const bool vec = (width % 4) == 0 && tdk_aligned(lum_in, 16) && tdk_aligned(lum_out, 16);
    return launch_tiles<T, T, 0>(in, nullptr, reinterpret_cast<T*>(lum_out), tab, width, height, d, sigma_s, sigma_r, detail, L, lds_bytes, vec,
                                 (flags & TDK_BILATERAL_PREPARED) != 0, s);
  }
  const int rc = build_grid<T>(in, grid, tmp, width, height, d, sigma_s, sigma_r, s);
  if (rc != TDK_OK) return rc;
  TDK_LAUNCH("tdk_bilateral(slice)", slice_kernel<T>, dim3(stream_blocks((int64_t)width * height)), dim3(256), 0, s, in, grid, reinterpret_cast<T*>(lum_out),
             width, height, d, sigma_s, sigma_r, detail);
  return TDK_OK;
}

template <typename T>
int launch_rgb(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, float sigma_s, float sigma_r, float detail, int log_mode, float eps,
               int dtype, unsigned flags, hipStream_t s, const float* lum_in = nullptr) {
  const GridDims d = compute_grid_size(width, height, sigma_s, sigma_r);
  const size_t tables = tile_table_bytes(width, height, sigma_s, sigma_r);
  int* tab = reinterpret_cast<int*>(workspace);
  float* grid = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + tables);
  float* tmp = grid + grid_floats(d);
  const float* plane = lum_in;
  int rc = TDK_OK;
  if (!plane) {  // the producer of rgb_in did not hand its luminance plane over: extract it
    float* mine = tmp + grid_floats(d);
    rc = tdk_compute_luminance(rgb_in, mine, (int64_t)width * height, log_mode, eps, dtype, TDK_F32, reinterpret_cast<tdk_stream_t>(s));
    if (rc != TDK_OK) return rc;
    plane = mine;
  }
  const bool vec = (width % 4) == 0 && tdk_aligned(rgb_in, 16) && tdk_aligned(rgb_out, 16) && tdk_aligned(plane, 16);
  TileLds L;
  size_t lds_bytes = 0;
  if (!(flags & TDK_BILATERAL_GENERAL_PATH) && plan_tiles(width, height, d, sigma_s, sigma_r, detail, &L, &lds_bytes)) {
    const bool prepared = (flags & TDK_BILATERAL_PREPARED) != 0;
    if (log_mode) return launch_tiles<float, T, 2>(plane, reinterpret_cast<const T*>(rgb_in), reinterpret_cast<T*>(rgb_out), tab, width, height, d, sigma_s, sigma_r, detail, L, lds_bytes, vec, prepared, s);
    return launch_tiles<float, T, 1>(plane, reinterpret_cast<const T*>(rgb_in), reinterpret_cast<T*>(rgb_out), tab, width, height, d, sigma_s, sigma_r, detail, L, lds_bytes, vec, prepared, s);
  }
  rc = build_grid<float>(plane, grid, tmp, width, height, d, sigma_s, sigma_r, s);
  if (rc != TDK_OK) return rc;
  const unsigned blocks = stream_blocks((int64_t)width * height / (vec ? 4 : 1));
  const T* rin = reinterpret_cast<const T*>(rgb_in);
  T* rout = reinterpret_cast<T*>(rgb_out);
#define TDK_SM(LOGV, VECV) \
  TDK_LAUNCH("tdk_bilateral(slice+modify)", (slice_modify_kernel<T, LOGV, VECV>), dim3(blocks), dim3(256), 0, s, plane, grid, rin, rout, width, height, d, \
             sigma_s, sigma_r, detail)
  if (log_mode) { if (vec) TDK_SM(true, 4); else TDK_SM(true, 1); }
  else { if (vec) TDK_SM(false, 4); else TDK_SM(false, 1); }
#undef TDK_SM
  return TDK_OK;
}

template <typename T>
int launch_lab(const float* lum, const float* ab, void* rgb_out, void* workspace, int width, int height, float sigma_s, float sigma_r, float detail, unsigned flags,
               hipStream_t s) {
  const GridDims d = compute_grid_size(width, height, sigma_s, sigma_r);
  const size_t tables = tile_table_bytes(width, height, sigma_s, sigma_r);
  int* tab = reinterpret_cast<int*>(workspace);
  float* grid = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + tables);
  float* tmp = grid + grid_floats(d);
  TileLds L;
  size_t lds_bytes = 0;
  if (!(flags & TDK_BILATERAL_GENERAL_PATH) && plan_tiles(width, height, d, sigma_s, sigma_r, detail, &L, &lds_bytes)) {
    const bool vec = (width % 4) == 0 && tdk_aligned(rgb_out, 16) && tdk_aligned(lum, 16) && tdk_aligned(ab, 16);
    return launch_tiles<float, T, 3>(lum, reinterpret_cast<const T*>(ab), reinterpret_cast<T*>(rgb_out), tab, width, height, d, sigma_s, sigma_r, detail, L, lds_bytes,
                                     vec, (flags & TDK_BILATERAL_PREPARED) != 0, s);
  }
  const int rc = build_grid<float>(lum, grid, tmp, width, height, d, sigma_s, sigma_r, s);
  if (rc != TDK_OK) return rc;
  TDK_LAUNCH("tdk_bilateral(slice+lab)", slice_lab_kernel<T>, dim3(stream_blocks((int64_t)width * height)), dim3(256), 0, s, lum, grid, ab, reinterpret_cast<T*>(rgb_out),
             width, height, d, sigma_s, sigma_r, detail);
  return TDK_OK;
}

}  // namespace

TDK_EXPORT int tdk_bilateral_lab(const float* lum_in, const float* ab_in, void* rgb_out, void* workspace, int width, int height, float sigma_s, float sigma_r,
                                 float detail, int out_dtype, unsigned flags, tdk_stream_t stream) {
  TDK_REQUIRE(lum_in && ab_in && rgb_out && workspace, "tdk_bilateral_lab: null pointer");
  TDK_REQUIRE(width > 0 && height > 0, "Invalid dimensions");
  TDK_REQUIRE(sigma_s > 0.0f && sigma_r > 0.0f, "tdk_bilateral_lab: sigmas must be positive");
  TDK_REQUIRE((flags & ~(TDK_BILATERAL_PREPARED | TDK_BILATERAL_GENERAL_PATH)) == 0, "tdk_bilateral_lab: unknown flags 0x%x", flags);
  TDK_DISPATCH_DTYPE(out_dtype, T, return launch_lab<T>(lum_in, ab_in, rgb_out, workspace, width, height, sigma_s, sigma_r, detail, flags, tdk_stream(stream)));
  return TDK_OK;
}

TDK_EXPORT int tdk_bilateral_grid_size(int width, int height, float sigma_s, float sigma_r, int size_xyz[3]) {
  TDK_REQUIRE(width > 0 && height > 0 && sigma_r > 0.0f && size_xyz, "tdk_bilateral_grid_size: invalid arguments");
  const GridDims d = compute_grid_size(width, height, sigma_s, sigma_r);
  size_xyz[0] = d.sx; size_xyz[1] = d.sy; size_xyz[2] = d.sz;
  return TDK_OK;
}

TDK_EXPORT size_t tdk_bilateral_workspace_bytes(int width, int height, float sigma_s, float sigma_r) {
  if (width <= 0 || height <= 0 || !(sigma_r > 0.0f)) return 0;
  const GridDims d = compute_grid_size(width, height, sigma_s, sigma_r);
  return plane_workspace_bytes(d, tile_table_bytes(width, height, sigma_s, sigma_r));
}

TDK_EXPORT size_t tdk_bilateral_rgb_workspace_bytes(int width, int height, float sigma_s, float sigma_r) {
  if (width <= 0 || height <= 0 || !(sigma_r > 0.0f)) return 0;
  const GridDims d = compute_grid_size(width, height, sigma_s, sigma_r);
  return rgb_workspace_bytes(d, width, height, tile_table_bytes(width, height, sigma_s, sigma_r));
}

TDK_EXPORT int tdk_bilateral_prepare(void* workspace, int width, int height, float sigma_s, float sigma_r, tdk_stream_t stream) {
  TDK_REQUIRE(workspace, "tdk_bilateral_prepare: null pointer");
  TDK_REQUIRE(width > 0 && height > 0, "Invalid dimensions");
  TDK_REQUIRE(sigma_s > 0.0f && sigma_r > 0.0f, "tdk_bilateral_prepare: sigmas must be positive");
  const GridDims d = compute_grid_size(width, height, sigma_s, sigma_r);
  TileLds L;
  size_t lds_bytes = 0;
  if (!plan_tiles(width, height, d, sigma_s, sigma_r, 0.0f, &L, &lds_bytes)) return TDK_OK;  // this geometry runs the four-kernel path: nothing to prepare
  return build_tables(reinterpret_cast<int*>(workspace), width, height, d, sigma_s, L, tdk_stream(stream));
}

TDK_EXPORT int tdk_bilateral_ex(const void* lum_in, void* lum_out, void* workspace, int width, int height, float sigma_s, float sigma_r,
                                float detail, int dtype, unsigned flags, tdk_stream_t stream) {
  TDK_REQUIRE(lum_in && lum_out && workspace, "tdk_bilateral: null pointer");
  TDK_REQUIRE(width > 0 && height > 0, "Invalid dimensions");
  TDK_REQUIRE(sigma_s > 0.0f && sigma_r > 0.0f, "tdk_bilateral: sigmas must be positive");
  TDK_REQUIRE((flags & ~(TDK_BILATERAL_PREPARED | TDK_BILATERAL_GENERAL_PATH)) == 0, "tdk_bilateral: unknown flags 0x%x", flags);
  TDK_DISPATCH_DTYPE(dtype, T, return launch<T>(lum_in, lum_out, workspace, width, height, sigma_s, sigma_r, detail, flags, tdk_stream(stream)));
  return TDK_OK;
}

TDK_EXPORT int tdk_bilateral(const void* lum_in, void* lum_out, void* workspace, int width, int height, float sigma_s, float sigma_r,
                             float detail, int dtype, tdk_stream_t stream) {
  return tdk_bilateral_ex(lum_in, lum_out, workspace, width, height, sigma_s, sigma_r, detail, dtype, 0u, stream);
}

TDK_EXPORT int tdk_bilateral_rgb_ex(const void* rgb_in, const float* lum_in, void* rgb_out, void* workspace, int width, int height, float sigma_s,
                                    float sigma_r, float detail, int log_mode, float eps, int dtype, unsigned flags, tdk_stream_t stream) {
  TDK_REQUIRE(rgb_in && rgb_out && workspace, "tdk_bilateral_rgb: null pointer");
  TDK_REQUIRE(width > 0 && height > 0, "Invalid dimensions");
  TDK_REQUIRE(sigma_s > 0.0f && sigma_r > 0.0f, "tdk_bilateral_rgb: sigmas must be positive");
  TDK_REQUIRE(!log_mode || eps > 0.0f, "Epsilon must be positive");
  TDK_REQUIRE(!lum_in || tdk_aligned(lum_in, 16), "tdk_bilateral_rgb: luminance plane must be 16-byte aligned");
  TDK_REQUIRE((flags & ~(TDK_BILATERAL_PREPARED | TDK_BILATERAL_GENERAL_PATH)) == 0, "tdk_bilateral_rgb: unknown flags 0x%x", flags);
  TDK_DISPATCH_DTYPE(dtype, T, return launch_rgb<T>(rgb_in, rgb_out, workspace, width, height, sigma_s, sigma_r, detail, log_mode, eps, dtype, flags, tdk_stream(stream), lum_in));
  return TDK_OK;
}

TDK_EXPORT int tdk_bilateral_rgb(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, float sigma_s, float sigma_r, float detail,
                                 int log_mode, float eps, int dtype, tdk_stream_t stream) {
  return tdk_bilateral_rgb_ex(rgb_in, nullptr, rgb_out, workspace, width, height, sigma_s, sigma_r, detail, log_mode, eps, dtype, 0u, stream);
}

TDK_EXPORT int tdk_bilateral_rgb_lum(const void* rgb_in, const float* lum_in, void* rgb_out, void* workspace, int width, int height, float sigma_s,
                                     float sigma_r, float detail, int log_mode, float eps, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(lum_in, "tdk_bilateral_rgb_lum: null pointer");
  return tdk_bilateral_rgb_ex(rgb_in, lum_in, rgb_out, workspace, width, height, sigma_s, sigma_r, detail, log_mode, eps, dtype, 0u, stream);
}
