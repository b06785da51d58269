/*
 * oracle/src/stencils.h -- PPG stencil formulas shared by ppg.c and rcd.c (RCD's border
 * rings reuse them).  CPU oracle, test infrastructure only.
 */
#ifndef TDK_ORACLE_STENCILS_H
#define TDK_ORACLE_STENCILS_H

#include "common.h"

void tdk_border_interpolate(const float* in, float* rgb, int width, int height, uint32_t pattern, int border);

/* Green at a red/blue site from the 7 samples along x (hx[3] = centre) and along y.
 * reference ppg.cu:184-221 == rcd.cu:345-383. */
static inline float tdk_ppg_green(const float hx[7], const float vy[7]) {
  const float pc = hx[3];
  const float guessx = (hx[2] + pc + hx[4]) * 2.0f - hx[5] - hx[1];
  const float diffx = (fabsf(hx[1] - pc) + fabsf(hx[5] - pc) + fabsf(hx[2] - hx[4])) * 3.0f +
                      (fabsf(hx[6] - hx[4]) + fabsf(hx[0] - hx[2])) * 2.0f;
  const float guessy = (vy[2] + pc + vy[4]) * 2.0f - vy[5] - vy[1];
  const float diffy = (fabsf(vy[1] - pc) + fabsf(vy[5] - pc) + fabsf(vy[2] - vy[4])) * 3.0f +
                      (fabsf(vy[6] - vy[4]) + fabsf(vy[0] - vy[2])) * 2.0f;
  if (diffx > diffy) {
    const float m = fminf(vy[2], vy[4]), M = fmaxf(vy[2], vy[4]);
    return fmaxf(fminf(guessy * 0.25f, M), m);
  }
  const float m = fminf(hx[2], hx[4]), M = fmaxf(hx[2], hx[4]);
  return fmaxf(fminf(guessx * 0.25f, M), m);
}

/* Red/blue fill from the 3x3 RGB neighbourhood nb[row][col][channel]; `col` holds the
 * centre pixel on entry and the filled pixel on exit.  c = CFA colour of the centre,
 * red_in_row = fc(row, col+1) == 0.  reference ppg.cu:289-335 == rcd.cu:444-490. */
static inline void tdk_ppg_redblue(float nb[3][3][3], int c, int red_in_row, float col[3]) {
  if (c == 1 || c == 3) {
    const float* nt = nb[0][1];
    const float* nbm = nb[2][1];
    const float* nl = nb[1][0];
    const float* nr = nb[1][2];
    if (red_in_row) {
      col[2] = (nt[2] + nbm[2] + 2.0f * col[1] - nt[1] - nbm[1]) * 0.5f;
      col[0] = (nl[0] + nr[0] + 2.0f * col[1] - nl[1] - nr[1]) * 0.5f;
    } else {
      col[0] = (nt[0] + nbm[0] + 2.0f * col[1] - nt[1] - nbm[1]) * 0.5f;
      col[2] = (nl[2] + nr[2] + 2.0f * col[1] - nl[1] - nr[1]) * 0.5f;
    }
  } else {
    const float* ntl = nb[0][0];
    const float* ntr = nb[0][2];
    const float* nbl = nb[2][0];
    const float* nbr = nb[2][2];
    const int k = (c == 0) ? 2 : 0; /* red site fills blue, blue site fills red */
    const float diff1 = fabsf(ntl[k] - nbr[k]) + fabsf(ntl[1] - col[1]) + fabsf(nbr[1] - col[1]);
    const float guess1 = ntl[k] + nbr[k] + 2.0f * col[1] - ntl[1] - nbr[1];
    const float diff2 = fabsf(ntr[k] - nbl[k]) + fabsf(ntr[1] - col[1]) + fabsf(nbl[1] - col[1]);
    const float guess2 = ntr[k] + nbl[k] + 2.0f * col[1] - ntr[1] - nbl[1];
    if (diff1 > diff2) col[k] = guess2 * 0.5f;
    else if (diff1 < diff2) col[k] = guess1 * 0.5f;
    else col[k] = (guess1 + guess2) * 0.25f;
  }
}

#endif
