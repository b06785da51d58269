// tdk_bilateral_tile.h -- the bilateral tile kernel (small sigma_s: the whole op in one kernel, grid in LDS).
// Included TWICE by bilateral.hip, inside two namespaces (no include guard on purpose):
//   bt_exact (TDK_BT_FAST 0, no FMA contraction): every cell and every sliced value by the oracle's expressions in the oracle's
//            order -- bit-identical to oracle/src/bilateral.c and to the four-kernel path.  MODE 0 / 1 / 2.
//   bt_fast  (TDK_BT_FAST 1, contraction allowed): the same algorithm for the Lab hand-over chain (MODE 3), whose parity is by
//            the colour operators' tolerance anyway: sample coordinate L * (1 / sigma_r) instead of the exact quotient, splat
//            contributions and blurs as fused multiply-adds, the trilinear slice factored into seven lerps (14 instructions
//            instead of the 31 of the eight-term sum of triple products).  Differences ~1e-6 of the lightness.
// Shares with bilateral.hip: GridDims, TileLds, AxisTile, tab_rec, win_lo / win_np, fast_div, lds_barrier, FTW / FTH / FNT, TAB_*.

// The NX candidate pixels per row of one cell column, nmy rows: raster order, each pixel adds its two z
// contributions to the column (read both cells, then write both: one LDS round trip per pixel).
template <int NX>
__device__ __forceinline__ void splat_column(float* __restrict__ acc, const float* __restrict__ urow, const float* __restrict__ wxp, int wxs_stride,
                                             const float* __restrict__ wyp, int wys_stride, int nmy, int LWS, int PS, int sz, float contrib) {
  float wxs[NX];
#pragma unroll
  for (int k = 0; k < NX; k++) wxs[k] = wxp[k * wxs_stride];
  for (int j = 0; j < nmy; j++, urow += LWS) {
    const float wy = wyp[j * wys_stride];
    float gzv[NX];
#pragma unroll
    for (int k = 0; k < NX; k++) gzv[k] = urow[k];
#pragma unroll
    for (int k = 0; k < NX; k++) {
      const float wxy = wxs[k] * wy;
      const float gz = gzv[k];
      const int iz = min((int)gz, sz - 2);
      const float fz = gz - (float)iz;
      float* p0 = acc + iz * PS;
      const float a0 = p0[0], a1 = p0[PS];
#if TDK_BT_FAST
      const float wc = wxy * contrib;
      p0[0] = __builtin_fmaf(wc, 1.0f - fz, a0);
      p0[PS] = __builtin_fmaf(wc, fz, a1);
#else
      p0[0] = a0 + wxy * (1.0f - fz) * contrib;
      p0[PS] = a1 + wxy * fz * contrib;
#endif
    }
  }
}

// gz = clamp(L / sigma_r, 0, ztop) of N samples.  x / c for a divisor c that is constant over the launch, rc = 1.0f / c
// (correctly rounded, host): q = RN(x * rc), r = x - q * c exactly (fma), q' = RN(q + r * rc) is the correctly rounded quotient
// (Markstein) as long as nothing under- or overflows -- 3 instructions instead of the ~12 of v_div_scale / v_rcp / Newton /
// v_div_fixup; ONE range test covers all N samples (wave-uniform, so that the compiler keeps a real, never taken branch to
// the plain division instead of computing both).
template <int N>
__device__ __forceinline__ void sample_gz(const float (&v)[N], float (&g)[N], float sigma_r, float rc_r, float ztop) {
#if TDK_BT_FAST
#pragma unroll
  for (int k = 0; k < N; k++) g[k] = clampf(v[k] * rc_r, 0.0f, ztop);  // (the splat and the slice are continuous in gz: a cell boundary crossed by one ulp changes nothing)
  return;
#endif
  float hi = fabsf(v[0]);
  float lo = (v[0] == 0.0f) ? 1.0f : hi;  // zeros are safe: take them out of the lower bound
#pragma unroll
  for (int k = 1; k < N; k++) {
    const float a = fabsf(v[k]);
    hi = fmaxf(hi, a);  // (a NaN sample fails the test below through lo or hi: fmaxf / fminf drop it, so test it separately)
    lo = fminf(lo, (v[k] == 0.0f) ? 1.0f : a);
  }
  bool safe = lo >= 0x1p-40f && hi <= 0x1p40f && sigma_r >= 0x1p-20f && sigma_r <= 0x1p20f;
#pragma unroll
  for (int k = 0; k < N; k++) safe = safe && (v[k] == v[k]);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!safe) != 0, 0)) {
#pragma unroll
    for (int k = 0; k < N; k++) g[k] = clampf(v[k] / sigma_r, 0.0f, ztop);
    return;
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    const float q = v[k] * rc_r;
    const float r = __builtin_fmaf(-q, sigma_r, v[k]);
    g[k] = clampf(__builtin_fmaf(r, rc_r, q), 0.0f, ztop);
  }
}

// MODE 0: luminance plane in (TL == T) -> filtered plane out; 1 / 2: fp32 plane + RGB in -> RGB out (linear / log);
// 3: fp32 plane + the pixels' Lab chroma (a, b: two floats per pixel, passed through the `rgb` pointer) in -> RGB out: the Lab
// hand-over chain (color.hip: lum_lab_extract) -- modify_luminance without its RGB -> Lab half
template <typename TL, typename T, int MODE, int VEC>
#ifndef TDK_BIL_WPE
#define TDK_BIL_WPE 8  // waves per SIMD the register budget is set for (experiments: co-residency with other frames' kernels)
#endif
__global__ __launch_bounds__(FNT) __attribute__((amdgpu_waves_per_eu(TDK_BIL_WPE, TDK_BIL_WPE))) void bilateral_tile_kernel(
    const TL* __restrict__ lum, const T* __restrict__ rgb, T* __restrict__ out, const int* __restrict__ tab, int width, int height, GridDims d,
    float sigma_r, int tiles_x, int ntiles, TileLds L) {
  extern __shared__ float smem[];
#ifdef TDK_BIL_TIMING
  unsigned long long bil_t0 = clock64();
#endif
  float* A = smem;                      // [sz][plane] grid, cell (lx, ly) at ly * RS + lx
  float* U = A + d.sz * L.plane;        // z sample coordinate of every pixel of the sample window (lh rows of lw), then blur temp
  float* gxs = U + L.usize;             // x sample coordinate of pixel column px_lo + i
  float* gys = gxs + L.lw;
  int* TX = reinterpret_cast<int*>(U + L.lw * L.lh);  // (tail of U, dead before the blur) x record: start[ncx] then weights[ncx][TAB_W]
  int* TY = TX + L.ncx * (1 + TAB_W);

  // consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD a contiguous run of tiles
  const int chunk = gridDim.x >> 3;
  const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int x0 = txi * FTW, y0 = tyi * FTH;
  const int recx = tab_rec(L.lw, L.ncx), recy = tab_rec(L.lh, L.ncy);
  const int* rx = tab + txi * recx;
  const int* ry = tab + tiles_x * recx + tyi * recy;
  const int px_lo = win_lo(x0, L.hx), py_lo = win_lo(y0, L.hy);
  const int npx = win_np(px_lo, L.lw, width), npy = win_np(py_lo, L.lh, height);
  const int tid = threadIdx.x;
  const int PS = L.plane, RS = L.rs, LWS = L.lw;
  const float ztop = (float)(d.sz - 1), rc_r = L.rc_r;

  // ---- set-up: the tile's luminance samples and its two table records into LDS.  All global loads are issued before
  // the first one is waited for; their addresses need nothing but the kernel arguments.
  {
    // each record part is at most 2 * FNT words (plan_tiles): two guarded copies, no loop
    auto copy2 = [&](int* dst, const int* from, int n) {
      if (tid < n) dst[tid] = from[tid];
      if (tid + FNT < n) dst[tid + FNT] = from[tid + FNT];
    };
    auto copy_records = [&]() {
      copy2(reinterpret_cast<int*>(gxs), rx + TAB_HDR, L.lw);
      copy2(reinterpret_cast<int*>(gys), ry + TAB_HDR, L.lh);
      copy2(TX, rx + TAB_HDR + L.lw, L.ncx * (1 + TAB_W));
      copy2(TY, ry + TAB_HDR + L.lh, L.ncy * (1 + TAB_W));
    };
    const TL* src = lum + (size_t)py_lo * width + px_lo;
    if constexpr (VEC == 4) {
      // 4 samples per load: the window starts on a multiple of 4 pixels, the rows are 16-B aligned (host-checked)
      const int qw = L.lw >> 2, total = qw * npy;
      constexpr int NB = 2;
      for (int base = tid; base - tid < total; base += NB * FNT) {  // uniform trip count: every thread helps copy the records
        float v[NB][4];
        int at[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
          const int q = base + k * FNT;
          const int r = fast_div(q, L.inv_qw), c = (q - r * qw) * 4;
          at[k] = (q < total && c < npx) ? r * LWS + c : -1;
          if (at[k] >= 0) s4_io<TL>::load(src + (size_t)r * width + c, 0, v[k]);
        }
        if (base == tid) copy_records();  // the records ride behind the first batch of samples
#pragma unroll
        for (int k = 0; k < NB; k++) {
          if (at[k] >= 0) {
            float g[4];
            sample_gz<4>(v[k], g, sigma_r, rc_r, ztop);  // make_sample's gz
            *reinterpret_cast<float4*>(U + at[k]) = make_float4(g[0], g[1], g[2], g[3]);
          }
        }
      }
    } else {
      const int total = LWS * npy;
      constexpr int NB = 7;
      for (int base = tid; base - tid < total; base += NB * FNT) {  // uniform trip count: every thread helps copy the records
        float v[NB];
        bool on[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
          const int i = base + k * FNT;
          const int r = fast_div(i, L.inv_lw), c = i - r * LWS;
          on[k] = i < total && c < npx;
          v[k] = on[k] ? ld(src, (size_t)r * width + c) : 0.0f;
        }
        if (base == tid) copy_records();
        float g[NB];
        sample_gz<NB>(v, g, sigma_r, rc_r, ztop);
#pragma unroll
        for (int k = 0; k < NB; k++)
          if (on[k]) U[base + k * FNT] = g[k];
      }
    }
  }
  AxisTile ax, ay;  // only the cell ranges; the pixel windows are px_lo / npx, py_lo / npy
  ax.c_lo = rx[0]; ax.nc = rx[1]; ay.c_lo = ry[0]; ay.nc = ry[1];
  const int nmx = rx[2], nmy = ry[2];
  BIL_MARK(8);
  lds_barrier();
  BIL_MARK(0);

  // ---- splat (gather, raster order per column; same expressions as splat_gather_kernel).  A thread owns the sz cells of
  // its column: it clears them and accumulates in place.
  const float contrib = L.contrib;
  {
    const int ncol = RS * ay.nc;
    const float* WX = reinterpret_cast<const float*>(TX + L.ncx);
    const float* WY = reinterpret_cast<const float*>(TY + L.ncy);
    for (int c = tid; c < ncol; c += FNT) {
      const int ly = fast_div(c, L.inv_rs), lx = c - ly * RS;
      if (lx >= ax.nc) continue;  // padding column of the odd row stride
      const int xa = TX[lx], ya = TY[ly];
      float* acc = A + c;
      for (int z = 0; z < d.sz; z++) acc[z * PS] = 0.0f;
      if (xa < 0 || ya < 0) continue;  // cell outside the grid: stays zero
      const float* urow = U + ya * LWS + xa;
      const float* wxp = WX + lx;
      const float* wyp = WY + ly;
      switch (nmx) {
        case 1: splat_column<1>(acc, urow, wxp, L.ncx, wyp, L.ncy, nmy, LWS, PS, d.sz, contrib); break;
        case 2: splat_column<2>(acc, urow, wxp, L.ncx, wyp, L.ncy, nmy, LWS, PS, d.sz, contrib); break;
        case 3: splat_column<3>(acc, urow, wxp, L.ncx, wyp, L.ncy, nmy, LWS, PS, d.sz, contrib); break;
        case 4: splat_column<4>(acc, urow, wxp, L.ncx, wyp, L.ncy, nmy, LWS, PS, d.sz, contrib); break;
        case 5: splat_column<5>(acc, urow, wxp, L.ncx, wyp, L.ncy, nmy, LWS, PS, d.sz, contrib); break;
        case 6: splat_column<6>(acc, urow, wxp, L.ncx, wyp, L.ncy, nmy, LWS, PS, d.sz, contrib); break;
        case 7: splat_column<7>(acc, urow, wxp, L.ncx, wyp, L.ncy, nmy, LWS, PS, d.sz, contrib); break;
        default: splat_column<8>(acc, urow, wxp, L.ncx, wyp, L.ncy, nmy, LWS, PS, d.sz, contrib); break;
      }
    }
  }
  lds_barrier();
  BIL_MARK(1);

  // ---- blur x: A -> U, one thread per (z, row) with a register window, CH cells per step: the reads of a step are
  // independent and issued together (a cell-by-cell walk pays one LDS latency per cell: it was 20 % of the workgroup's
  // life).  Cells beyond the LDS tile read as zero; they only feed cells nobody slices.
  const float w0 = 6.0f / 16.0f, w1 = 4.0f / 16.0f, w2 = 1.0f / 16.0f;
  {
    constexpr int CH = 13;
    for (int row = tid; row < d.sz * L.ncy; row += FNT) {
      const int z = fast_div(row, L.inv_ncy), ly = row - z * L.ncy;
      if (ly >= ay.nc) continue;
      const float* p = A + z * PS + ly * RS;
      float* q = U + z * PS + ly * RS;
      float m2 = 0.0f, m1 = 0.0f, c0 = p[0], p1 = p[1];  // nc >= 6
      for (int lx0 = 0; lx0 < ax.nc; lx0 += CH) {
        float nx[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) nx[k] = (lx0 + k + 2 < ax.nc) ? p[lx0 + k + 2] : 0.0f;
#pragma unroll
        for (int k = 0; k < CH; k++) {
          if (lx0 + k < ax.nc) q[lx0 + k] = c0 * w0 + w1 * (p1 + m1) + w2 * (nx[k] + m2);
          m2 = m1; m1 = c0; c0 = p1; p1 = nx[k];
        }
      }
    }
  }
  lds_barrier();
  BIL_MARK(2);
  // ---- blur y: U -> A, one thread per (z, column)
  {
    constexpr int CH = 11;
    for (int cc = tid; cc < d.sz * RS; cc += FNT) {
      const int z = fast_div(cc, L.inv_rs), lx = cc - z * RS;
      if (lx >= ax.nc) continue;
      const float* p = U + z * PS + lx;
      float* q = A + z * PS + lx;
      float m2 = 0.0f, m1 = 0.0f, c0 = p[0], p1 = p[RS];
      for (int ly0 = 0; ly0 < ay.nc; ly0 += CH) {
        float nx[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) nx[k] = (ly0 + k + 2 < ay.nc) ? p[(ly0 + k + 2) * RS] : 0.0f;
#pragma unroll
        for (int k = 0; k < CH; k++) {
          if (ly0 + k < ay.nc) q[(ly0 + k) * RS] = c0 * w0 + w1 * (p1 + m1) + w2 * (nx[k] + m2);
          m2 = m1; m1 = c0; c0 = p1; p1 = nx[k];
        }
      }
    }
  }
  lds_barrier();
  BIL_MARK(3);
  // ---- z derivative, in place, one thread per column: the whole column in registers when it is short
  {
    const float v1 = 4.0f / 16.0f, v2 = 2.0f / 16.0f;
    constexpr int ZR = 8;
    for (int c = tid; c < RS * ay.nc; c += FNT) {
      float* p = A + c;
      if (d.sz <= ZR) {
        float v[ZR + 4];
        v[0] = v[1] = 0.0f;
#pragma unroll
        for (int z = 0; z < ZR + 2; z++) v[z + 2] = (z < d.sz) ? p[z * PS] : 0.0f;
#pragma unroll
        for (int z = 0; z < ZR; z++)
          if (z < d.sz) p[z * PS] = v1 * (v[z + 3] - v[z + 1]) + v2 * (v[z + 4] - v[z]);
      } else {
        float m2 = 0.0f, m1 = 0.0f, c0 = p[0];
        float p1 = (d.sz > 1) ? p[PS] : 0.0f;
        for (int z = 0; z < d.sz; z++) {
          const float p2 = (z + 2 < d.sz) ? p[(z + 2) * PS] : 0.0f;
          p[z * PS] = v1 * (p1 - m1) + v2 * (p2 - m2);
          m2 = m1; m1 = c0; c0 = p1; p1 = p2;
        }
      }
    }
  }
  lds_barrier();
  BIL_MARK(4);

  // ---- slice (+ put the new lightness back into the pixel)
  const float norm = L.norm;
  constexpr int GW = FTW / VEC;
  for (int g = tid; g < GW * FTH; g += FNT) {
    const int py = g / GW, y = y0 + py, x = x0 + (g - py * GW) * VEC;
    if (x >= width || y >= height) continue;
    const size_t i0 = (size_t)y * width + x;
    float Lv[VEC], o[MODE == 0 ? VEC : 3 * VEC], c2[MODE == 3 ? 2 * VEC : 1];
    if constexpr (VEC == 4) {
      s4_io<TL>::load(lum, i0 >> 2, Lv);
      if constexpr (MODE == 3) {
        const float* ab = reinterpret_cast<const float*>(rgb);
        s4_io<float>::load(ab, i0 >> 1, c2);
        s4_io<float>::load(ab, (i0 >> 1) + 1, c2 + 4);
      } else if constexpr (MODE != 0) {
        rgb4_io<T>::load(rgb, i0 >> 2, o);
      }
    } else {
      Lv[0] = ld(lum, i0);
      if constexpr (MODE == 3) { const float* ab = reinterpret_cast<const float*>(rgb); c2[0] = ab[2 * i0]; c2[1] = ab[2 * i0 + 1]; }
      else if constexpr (MODE != 0) { o[0] = ld(rgb, i0 * 3); o[1] = ld(rgb, i0 * 3 + 1); o[2] = ld(rgb, i0 * 3 + 2); }
    }
    const float gy = gys[y - py_lo];
    const int iy = min((int)gy, d.sy - 2);
    const float by = gy - (float)iy, ayw = 1.0f - by;
    const float* grow = A + (iy - ay.c_lo) * RS - ax.c_lo;
    float gzv[VEC];
    sample_gz<VEC>(Lv, gzv, sigma_r, rc_r, ztop);
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      const float Lp = Lv[k];
      const float gx = gxs[x + k - px_lo];
      const int ix = min((int)gx, d.sx - 2);
      const float bx = gx - (float)ix, axw = 1.0f - bx;
      const float gz = gzv[k];
      const int iz = min((int)gz, d.sz - 2);
      const float bz = gz - (float)iz, azw = 1.0f - bz;
      const int oy = RS, oz = PS;
      const float* gp = grow + iz * PS + ix;
#if TDK_BT_FAST
      auto lerp2 = [](float a, float b, float t) { return __builtin_fmaf(t, b - a, a); };
      const float z0 = lerp2(lerp2(gp[0], gp[1], bx), lerp2(gp[oy], gp[oy + 1], bx), by);
      const float z1 = lerp2(lerp2(gp[oz], gp[oz + 1], bx), lerp2(gp[oz + oy], gp[oz + oy + 1], bx), by);
      const float Ldiff = lerp2(z0, z1, bz);
      (void)axw; (void)ayw; (void)azw;
#else
      const float Ldiff = gp[0] * axw * ayw * azw + gp[1] * bx * ayw * azw + gp[oy] * axw * by * azw + gp[oy + 1] * bx * by * azw +
                          gp[oz] * axw * ayw * bz + gp[oz + 1] * bx * ayw * bz + gp[oz + oy] * axw * by * bz + gp[oz + oy + 1] * bx * by * bz;
#endif
      const float Lnew = fmaxf(0.0f, Lp + norm * Ldiff);
      if constexpr (MODE == 0) {
        o[k] = Lnew;
      } else if constexpr (MODE == 3) {
        const f3 r = clip3(cA::lab_to_rgb(mk3(fmaxf(0.0f, fminf(1.0f, Lnew)), c2[2 * k], c2[2 * k + 1])));  // the second half of modify_luminance
        o[3 * k] = r.x; o[3 * k + 1] = r.y; o[3 * k + 2] = r.z;
      } else {
        const f3 c = mk3(o[3 * k], o[3 * k + 1], o[3 * k + 2]);
        const f3 r = (MODE == 2) ? cA::modify_log_luminance(c, Lnew) : cA::modify_luminance(c, Lnew);
        o[3 * k] = r.x; o[3 * k + 1] = r.y; o[3 * k + 2] = r.z;
      }
    }
    if constexpr (MODE == 0) {
      if constexpr (VEC == 4) s4_io<T>::store(out, i0 >> 2, o);
      else st(out, i0, o[0]);
    } else {
      if constexpr (VEC == 4) rgb4_io<T>::store(out, i0 >> 2, o);
      else { st(out, i0 * 3, o[0]); st(out, i0 * 3 + 1, o[1]); st(out, i0 * 3 + 2, o[2]); }
    }
  }
  BIL_MARK(5);
}
