// tdk_stencils.h -- PPG stencil formulas shared by the PPG kernel and RCD's border rings.
// Same term order as the oracle (oracle/src/stencils.h) so results are bit-identical.
#pragma once

#include "tdk_common.h"

// 3x3 same-colour average for the outer ring (reference csrc/debayer/ppg.cu:342-389).
// `rd(x, y)` returns the raw sample (no clamp); called only for in-image (x, y).
template <typename RD>
__device__ __forceinline__ f3 border_average(RD rd, int x, int y, int width, int height, uint32_t pattern) {
  float sum[3] = {0.0f, 0.0f, 0.0f};
  int cnt[3] = {0, 0, 0};
  for (int j = y - 1; j <= y + 1; j++)
    for (int i = x - 1; i <= x + 1; i++)
      if (j >= 0 && i >= 0 && j < height && i < width) {
        const int f = cfa_color(j, i, pattern);
        sum[f] += fmaxf(0.0f, rd(i, j));
        cnt[f]++;
      }
  const float own = fmaxf(0.0f, rd(x, y));
  f3 o;
  o.x = cnt[0] > 0 ? sum[0] / (float)cnt[0] : own;
  // the reference adds the (always empty) code-3 bucket: (sum[1] + 0) / (count[1] + 0)
  o.y = cnt[1] > 0 ? (sum[1] + 0.0f) / (float)cnt[1] : own;
  o.z = cnt[2] > 0 ? sum[2] / (float)cnt[2] : own;
  const int f = cfa_color(y, x, pattern);
  if (f == 0) o.x = own;
  else if (f == 2) o.z = own;
  else o.y = own;
  return o;
}

// Green at a red/blue site from 7 samples along x (h[3] = centre) and along y
// (reference ppg.cu:184-221 == rcd.cu:345-383).
__device__ __forceinline__ float ppg_green(const float h[7], const float v[7]) {
  const float pc = h[3];
  const float guessx = (h[2] + pc + h[4]) * 2.0f - h[5] - h[1];
  const float diffx = (fabsf(h[1] - pc) + fabsf(h[5] - pc) + fabsf(h[2] - h[4])) * 3.0f + (fabsf(h[6] - h[4]) + fabsf(h[0] - h[2])) * 2.0f;
  const float guessy = (v[2] + pc + v[4]) * 2.0f - v[5] - v[1];
  const float diffy = (fabsf(v[1] - pc) + fabsf(v[5] - pc) + fabsf(v[2] - v[4])) * 3.0f + (fabsf(v[6] - v[4]) + fabsf(v[0] - v[2])) * 2.0f;
  if (diffx > diffy) {
    const float m = fminf(v[2], v[4]), M = fmaxf(v[2], v[4]);
    return fmaxf(fminf(guessy * 0.25f, M), m);
  }
  const float m = fminf(h[2], h[4]), M = fmaxf(h[2], h[4]);
  return fmaxf(fminf(guessx * 0.25f, M), m);
}

// Red/blue fill (reference ppg.cu:289-335 == rcd.cu:444-490).  `nb(dx, dy)` returns the RGB
// neighbour; `c` = CFA colour of the centre; red_in_row = fc(row, col + 1) == 0.
template <typename NB> __device__ __forceinline__ f3 ppg_redblue(NB nb, f3 col, int c, bool red_in_row) {
  if (c == 1 || c == 3) {
    const f3 nt = nb(0, -1), nbm = nb(0, 1), nl = nb(-1, 0), nr = nb(1, 0);
    if (red_in_row) {
      col.z = (nt.z + nbm.z + 2.0f * col.y - nt.y - nbm.y) * 0.5f;
      col.x = (nl.x + nr.x + 2.0f * col.y - nl.y - nr.y) * 0.5f;
    } else {
      col.x = (nt.x + nbm.x + 2.0f * col.y - nt.y - nbm.y) * 0.5f;
      col.z = (nl.z + nr.z + 2.0f * col.y - nl.y - nr.y) * 0.5f;
    }
  } else {
    const f3 ntl = nb(-1, -1), ntr = nb(1, -1), nbl = nb(-1, 1), nbr = nb(1, 1);
    if (c == 0) {  // red site: fill blue
      const float diff1 = fabsf(ntl.z - nbr.z) + fabsf(ntl.y - col.y) + fabsf(nbr.y - col.y);
      const float guess1 = ntl.z + nbr.z + 2.0f * col.y - ntl.y - nbr.y;
      const float diff2 = fabsf(ntr.z - nbl.z) + fabsf(ntr.y - col.y) + fabsf(nbl.y - col.y);
      const float guess2 = ntr.z + nbl.z + 2.0f * col.y - ntr.y - nbl.y;
      if (diff1 > diff2) col.z = guess2 * 0.5f;
      else if (diff1 < diff2) col.z = guess1 * 0.5f;
      else col.z = (guess1 + guess2) * 0.25f;
    } else {  // blue site: fill red
      const float diff1 = fabsf(ntl.x - nbr.x) + fabsf(ntl.y - col.y) + fabsf(nbr.y - col.y);
      const float guess1 = ntl.x + nbr.x + 2.0f * col.y - ntl.y - nbr.y;
      const float diff2 = fabsf(ntr.x - nbl.x) + fabsf(ntr.y - col.y) + fabsf(nbl.y - col.y);
      const float guess2 = ntr.x + nbl.x + 2.0f * col.y - ntr.y - nbl.y;
      if (diff1 > diff2) col.x = guess2 * 0.5f;
      else if (diff1 < diff2) col.x = guess1 * 0.5f;
      else col.x = (guess1 + guess2) * 0.25f;
    }
  }
  return col;
}

// Store four consecutive RGB pixels of row `y` starting at column `x` (x % 4 == 0).
template <typename T>
__device__ __forceinline__ void store_rgb4(T* out, int x, int y, int width, int vec_ok, const float px[12]) {
  const size_t p = (size_t)y * width + x;
  if (vec_ok) {
    rgb4_io<T>::store(out, p >> 2, px);
  } else {
    for (int k = 0; k < 4 && x + k < width; k++) {
      st(out, (p + k) * 3 + 0, px[3 * k]);
      st(out, (p + k) * 3 + 1, px[3 * k + 1]);
      st(out, (p + k) * 3 + 2, px[3 * k + 2]);
    }
  }
}
