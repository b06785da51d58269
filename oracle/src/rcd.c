/*
 * oracle/src/rcd.c -- Ratio Corrected Demosaic (CPU oracle, test infrastructure only).
 *
 * LITERAL restatement of reference csrc/debayer/rcd.cu, first call on a fresh workspace:
 * the eight H*W scratch planes are zero-initialised (rcd.cu:585-594), the planes are
 * shared exactly as the reference shares them (VP_diff = v_diff then p_diff,
 * HQ_diff = h_diff then q_diff, lpf_PQ = lpf then PQ_dir; rcd.cu:637-660) and the
 * half-density planes are addressed with the same flat `idx / 2` arithmetic
 * (rcd.cu:101,157,175-181,199-207).  That reproduces, as a pure function of the input,
 * everything the first call reads: zeros outside each step's write region and, in the
 * p/q planes, the same-call v_diff/h_diff values left in slots step 4.1 does not write.
 *
 * Launch order (rcd.cu:616-668): border_interpolate(3) -> border green (32) ->
 * border red/blue (16, in place) -> populate -> 1.1 -> 1.2 -> 2.1 -> 3.1 -> 4.1 -> 4.2
 * -> 5.1 -> 5.2 -> write_output(margin 7).  Width must be even (idx/2 packing).
 */
#include "common.h"
#include "stencils.h"

/* rcd.cu:285-385 */
static void border_green(const float* in, float* out, int w, int h, uint32_t pattern, int border) {
#pragma omp parallel for schedule(static)
  for (int y = 3; y < h - 3; y++) {
    for (int x = 3; x < w - 3; x++) {
      if (x >= border && x < w - border && y >= border && y < h - border) continue;
      const int c = cfa_color(y, x, pattern);
      float hx[7], vy[7];
      for (int d = -3; d <= 3; d++) {
        const int xx = x + d, yy = y + d;
        hx[d + 3] = (xx >= 0 && xx < w) ? fmaxf(0.0f, in[(size_t)y * w + xx]) : 0.0f;
        vy[d + 3] = (yy >= 0 && yy < h) ? fmaxf(0.0f, in[(size_t)yy * w + x]) : 0.0f;
      }
      float col[3] = {0.0f, 0.0f, 0.0f};
      col[c == 0 ? 0 : (c == 2 ? 2 : 1)] = hx[3];
      if (c == 0 || c == 2) col[1] = tdk_ppg_green(hx, vy);
      float* dst = out + ((size_t)y * w + x) * 3;
      for (int k = 0; k < 3; k++) dst[k] = fmaxf(col[k], 0.0f);
    }
  }
}

/* rcd.cu:387-493; the reference runs it in place -- it only reads native R/B and green
 * and only writes non-native R/B, so a read-from-copy pass is equivalent. */
static void border_redblue(float* out, int w, int h, uint32_t pattern, int border) {
  const size_t n3 = (size_t)w * h * 3;
  float* src = (float*)malloc(n3 * sizeof(float));
  memcpy(src, out, n3 * sizeof(float));
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      if (x >= border && x < w - border && y >= border && y < h - border) continue;
      float nb[3][3][3];
      for (int j = -1; j <= 1; j++)
        for (int i = -1; i <= 1; i++) {
          const int xx = x + i, yy = y + j;
          const int ok = xx >= 0 && yy >= 0 && xx < w && yy < h;
          for (int k = 0; k < 3; k++) nb[j + 1][i + 1][k] = ok ? fmaxf(0.0f, src[((size_t)yy * w + xx) * 3 + k]) : 0.0f;
        }
      float col[3] = {nb[1][1][0], nb[1][1][1], nb[1][1][2]};
      if (y > 0 && x > 0 && x < w - 1 && y < h - 1)
        tdk_ppg_redblue(nb, cfa_color(y, x, pattern), cfa_color(y, x + 1, pattern) == 0, col);
      float* dst = out + ((size_t)y * w + x) * 3;
      for (int k = 0; k < 3; k++) dst[k] = fmaxf(col[k], 0.0f);
    }
  }
  free(src);
}

/* dump_pq / dump_p / dump_q (each H*W floats, may be NULL): the lpf_PQ, VP_diff and HQ_diff planes
 * as they stand after step 4.2, in the reference's flat slot layout -- for the known-answer test of
 * the idx/2 slot aliasing (tests/test_oracle_kat.py). */
static int rcd_impl(const float* in, float* out, int w, int h, uint32_t pattern, float* dump_pq, float* dump_p, float* dump_q) {
  if (w & 1) return 1;
  const size_t n = (size_t)w * h;
  float* planes = (float*)calloc(n * 8, sizeof(float));
  float *cfa = planes, *rgb0 = planes + n, *rgb1 = planes + 2 * n, *rgb2 = planes + 3 * n;
  float *VH_dir = planes + 4 * n, *VP = planes + 5 * n, *HQ = planes + 6 * n, *lpfPQ = planes + 7 * n;
  float* rgb[3] = {rgb0, rgb1, rgb2};
  const int w2 = 2 * w, w3 = 3 * w, w4 = 4 * w;

  memset(out, 0, n * 3 * sizeof(float)); /* output_buffer_ = zeros (rcd.cu:586) */
  tdk_border_interpolate(in, out, w, h, pattern, 3);
  border_green(in, out, w, h, pattern, 32);
  border_redblue(out, w, h, pattern, 16);

  /* populate, scale = 1 (rcd.cu:30-46) */
#pragma omp parallel for schedule(static)
  for (int row = 0; row < h; row++)
    for (int col = 0; col < w; col++) {
      const int idx = row * w + col;
      const float val = 1.0f * fmaxf(0.0f, in[idx]);
      cfa[idx] = rgb[cfa_color(row, col, pattern)][idx] = val;
    }

  /* step 1.1 (rcd.cu:63-75) */
#pragma omp parallel for schedule(static)
  for (int row = 3; row <= h - 4; row++)
    for (int col = 3; col <= w - 4; col++) {
      const int idx = row * w + col;
      VP[idx] = f_sq(cfa[idx - w3] - 3.0f * cfa[idx - w2] - cfa[idx - w] + 6.0f * cfa[idx] - cfa[idx + w] - 3.0f * cfa[idx + w2] + cfa[idx + w3]);
      HQ[idx] = f_sq(cfa[idx - 3] - 3.0f * cfa[idx - 2] - cfa[idx - 1] + 6.0f * cfa[idx] - cfa[idx + 1] - 3.0f * cfa[idx + 2] + cfa[idx + 3]);
    }

  /* step 1.2 (rcd.cu:78-90) */
#pragma omp parallel for schedule(static)
  for (int row = 2; row <= h - 3; row++)
    for (int col = 2; col <= w - 3; col++) {
      const int idx = row * w + col;
      const float eps = 1e-10f;
      const float V_Stat = fmaxf(eps, VP[idx - w] + VP[idx] + VP[idx + w]);
      const float H_Stat = fmaxf(eps, HQ[idx - 1] + HQ[idx] + HQ[idx + 1]);
      VH_dir[idx] = V_Stat / (V_Stat + H_Stat);
    }

  /* step 2.1 (rcd.cu:93-104) */
#pragma omp parallel for schedule(static)
  for (int row = 2; row <= h - 2; row++)
    for (int col = 2 + (cfa_color(row, 0, pattern) & 1); col <= w - 2; col += 2) {
      const int idx = row * w + col;
      lpfPQ[idx / 2] = cfa[idx] + 0.5f * (cfa[idx - w] + cfa[idx + w] + cfa[idx - 1] + cfa[idx + 1]) +
                       0.25f * (cfa[idx - w - 1] + cfa[idx - w + 1] + cfa[idx + w - 1] + cfa[idx + w + 1]);
    }

  /* step 3.1 (rcd.cu:107-146) */
#pragma omp parallel for schedule(static)
  for (int row = 4; row <= h - 5; row++)
    for (int col = 4 + (cfa_color(row, 0, pattern) & 1); col <= w - 5; col += 2) {
      const int idx = row * w + col;
      const int lidx = idx / 2;
      const float eps = 1e-5f;
      const float* lpf = lpfPQ;
      const float VH_c = VH_dir[idx];
      const float VH_n = 0.25f * (VH_dir[idx - w - 1] + VH_dir[idx - w + 1] + VH_dir[idx + w - 1] + VH_dir[idx + w + 1]);
      const float VH_Disc = (fabsf(0.5f - VH_c) < fabsf(0.5f - VH_n)) ? VH_n : VH_c;
      const float cfai = cfa[idx];
      const float N_Grad = eps + fabsf(cfa[idx - w] - cfa[idx + w]) + fabsf(cfai - cfa[idx - w2]) + fabsf(cfa[idx - w] - cfa[idx - w3]) + fabsf(cfa[idx - w2] - cfa[idx - w4]);
      const float S_Grad = eps + fabsf(cfa[idx + w] - cfa[idx - w]) + fabsf(cfai - cfa[idx + w2]) + fabsf(cfa[idx + w] - cfa[idx + w3]) + fabsf(cfa[idx + w2] - cfa[idx + w4]);
      const float W_Grad = eps + fabsf(cfa[idx - 1] - cfa[idx + 1]) + fabsf(cfai - cfa[idx - 2]) + fabsf(cfa[idx - 1] - cfa[idx - 3]) + fabsf(cfa[idx - 2] - cfa[idx - 4]);
      const float E_Grad = eps + fabsf(cfa[idx + 1] - cfa[idx - 1]) + fabsf(cfai - cfa[idx + 2]) + fabsf(cfa[idx + 1] - cfa[idx + 3]) + fabsf(cfa[idx + 2] - cfa[idx + 4]);
      const float lpfi = lpf[lidx];
      const float N_Est = cfa[idx - w] * (lpfi + lpfi) / (eps + lpfi + lpf[lidx - w]);
      const float S_Est = cfa[idx + w] * (lpfi + lpfi) / (eps + lpfi + lpf[lidx + w]);
      const float W_Est = cfa[idx - 1] * (lpfi + lpfi) / (eps + lpfi + lpf[lidx - 1]);
      const float E_Est = cfa[idx + 1] * (lpfi + lpfi) / (eps + lpfi + lpf[lidx + 1]);
      const float V_Est = (S_Grad * N_Est + N_Grad * S_Est) / (N_Grad + S_Grad);
      const float H_Est = (W_Grad * E_Est + E_Grad * W_Est) / (E_Grad + W_Grad);
      rgb1[idx] = f_mix(V_Est, H_Est, VH_Disc);
    }

  /* step 4.1 (rcd.cu:149-163): every odd column of every row (its pattern argument is
   * unused); reuses VP/HQ as p_diff/q_diff at flat slot idx/2.  It only reads cfa, so the
   * row-parallel order is immaterial. */
#pragma omp parallel for schedule(static)
  for (int row = 3; row <= h - 4; row++)
    for (int col = 3; col <= w - 4; col += 2) {
      const int idx = row * w + col;
      const int idx2 = idx / 2;
      VP[idx2] = f_sq((cfa[idx - w3 - 3] - cfa[idx - w - 1] - cfa[idx + w + 1] + cfa[idx + w3 + 3]) - 3.0f * (cfa[idx - w2 - 2] + cfa[idx + w2 + 2]) + 6.0f * cfa[idx]);
      HQ[idx2] = f_sq((cfa[idx - w3 + 3] - cfa[idx - w + 1] - cfa[idx + w - 1] + cfa[idx + w3 - 3]) - 3.0f * (cfa[idx - w2 + 2] + cfa[idx + w2 - 2]) + 6.0f * cfa[idx]);
    }

  /* step 4.2 (rcd.cu:166-182): reads p/q slots, writes PQ_dir slots of lpf_PQ.  The
   * reference reads p_diff/q_diff (VP/HQ) and writes a different plane, so any order. */
#pragma omp parallel for schedule(static)
  for (int row = 2; row <= h - 3; row++)
    for (int col = 2 + (cfa_color(row, 0, pattern) & 1); col <= w - 3; col += 2) {
      const int idx = row * w + col;
      const int idx2 = idx / 2, idx3 = (idx - w - 1) / 2, idx4 = (idx + w - 1) / 2;
      const float eps = 1e-10f;
      const float P_Stat = fmaxf(eps, VP[idx3] + VP[idx2] + VP[idx4 + 1]);
      const float Q_Stat = fmaxf(eps, HQ[idx3 + 1] + HQ[idx2] + HQ[idx4]);
      lpfPQ[idx2] = P_Stat / (P_Stat + Q_Stat);
    }

  if (dump_pq) memcpy(dump_pq, lpfPQ, n * sizeof(float));
  if (dump_p) memcpy(dump_p, VP, n * sizeof(float));
  if (dump_q) memcpy(dump_q, HQ, n * sizeof(float));

  /* step 5.1 (rcd.cu:185-224): reads and writes rgb0/rgb2 but never the same site class */
#pragma omp parallel for schedule(static)
  for (int row = 4; row <= h - 4; row++)
    for (int col = 4 + (cfa_color(row, 0, pattern) & 1); col <= w - 4; col += 2) {
      float* rgbc = rgb[2 - cfa_color(row, col, pattern)];
      const float* PQ_dir = lpfPQ;
      const int idx = row * w + col;
      const int pqidx = idx / 2, pqidx2 = (idx - w - 1) / 2, pqidx3 = (idx + w - 1) / 2;
      const float eps = 1e-5f;
      const float PQ_c = PQ_dir[pqidx];
      const float PQ_n = 0.25f * (PQ_dir[pqidx2] + PQ_dir[pqidx2 + 1] + PQ_dir[pqidx3] + PQ_dir[pqidx3 + 1]);
      const float PQ_Disc = (fabsf(0.5f - PQ_c) < fabsf(0.5f - PQ_n)) ? PQ_n : PQ_c;
      const float NW_Grad = eps + fabsf(rgbc[idx - w - 1] - rgbc[idx + w + 1]) + fabsf(rgbc[idx - w - 1] - rgbc[idx - w3 - 3]) + fabsf(rgb1[idx] - rgb1[idx - w2 - 2]);
      const float NE_Grad = eps + fabsf(rgbc[idx - w + 1] - rgbc[idx + w - 1]) + fabsf(rgbc[idx - w + 1] - rgbc[idx - w3 + 3]) + fabsf(rgb1[idx] - rgb1[idx - w2 + 2]);
      const float SW_Grad = eps + fabsf(rgbc[idx - w + 1] - rgbc[idx + w - 1]) + fabsf(rgbc[idx + w - 1] - rgbc[idx + w3 - 3]) + fabsf(rgb1[idx] - rgb1[idx + w2 - 2]);
      const float SE_Grad = eps + fabsf(rgbc[idx - w - 1] - rgbc[idx + w + 1]) + fabsf(rgbc[idx + w + 1] - rgbc[idx + w3 + 3]) + fabsf(rgb1[idx] - rgb1[idx + w2 + 2]);
      const float NW_Est = rgbc[idx - w - 1] - rgb1[idx - w - 1];
      const float NE_Est = rgbc[idx - w + 1] - rgb1[idx - w + 1];
      const float SW_Est = rgbc[idx + w - 1] - rgb1[idx + w - 1];
      const float SE_Est = rgbc[idx + w + 1] - rgb1[idx + w + 1];
      const float P_Est = (NW_Grad * SE_Est + SE_Grad * NW_Est) / (NW_Grad + SE_Grad);
      const float Q_Est = (NE_Grad * SW_Est + SW_Grad * NE_Est) / (NE_Grad + SW_Grad);
      rgbc[idx] = rgb1[idx] + f_mix(P_Est, Q_Est, PQ_Disc);
    }

  /* step 5.2 (rcd.cu:227-282): green sites; reads R/B sites only */
#pragma omp parallel for schedule(static)
  for (int row = 4; row <= h - 4; row++)
    for (int col = 4 + (cfa_color(row, 1, pattern) & 1); col <= w - 4; col += 2) {
      const int idx = row * w + col;
      const float eps = 1e-5f;
      const float VH_c = VH_dir[idx];
      const float VH_n = 0.25f * (VH_dir[idx - w - 1] + VH_dir[idx - w + 1] + VH_dir[idx + w - 1] + VH_dir[idx + w + 1]);
      const float VH_Disc = (fabsf(0.5f - VH_c) < fabsf(0.5f - VH_n)) ? VH_n : VH_c;
      const float g = rgb1[idx];
      const float N1 = eps + fabsf(g - rgb1[idx - w2]);
      const float S1 = eps + fabsf(g - rgb1[idx + w2]);
      const float W1 = eps + fabsf(g - rgb1[idx - 2]);
      const float E1 = eps + fabsf(g - rgb1[idx + 2]);
      const float gN = rgb1[idx - w], gS = rgb1[idx + w], gW = rgb1[idx - 1], gE = rgb1[idx + 1];
      for (int c = 0; c <= 2; c += 2) {
        float* rgbc = rgb[c];
        const float SNabs = fabsf(rgbc[idx - w] - rgbc[idx + w]);
        const float EWabs = fabsf(rgbc[idx - 1] - rgbc[idx + 1]);
        const float N_Grad = N1 + SNabs + fabsf(rgbc[idx - w] - rgbc[idx - w3]);
        const float S_Grad = S1 + SNabs + fabsf(rgbc[idx + w] - rgbc[idx + w3]);
        const float W_Grad = W1 + EWabs + fabsf(rgbc[idx - 1] - rgbc[idx - 3]);
        const float E_Grad = E1 + EWabs + fabsf(rgbc[idx + 1] - rgbc[idx + 3]);
        const float N_Est = rgbc[idx - w] - gN;
        const float S_Est = rgbc[idx + w] - gS;
        const float W_Est = rgbc[idx - 1] - gW;
        const float E_Est = rgbc[idx + 1] - gE;
        const float V_Est = (N_Grad * S_Est + S_Grad * N_Est) / (N_Grad + S_Grad);
        const float H_Est = (E_Grad * W_Est + W_Grad * E_Est) / (E_Grad + W_Grad);
        rgbc[idx] = rgb1[idx] + f_mix(V_Est, H_Est, VH_Disc);
      }
    }

  /* write_output, scale = 1, margin 7 (rcd.cu:49-60) */
#pragma omp parallel for schedule(static)
  for (int row = 7; row < h - 7; row++)
    for (int col = 7; col < w - 7; col++) {
      const size_t idx = (size_t)row * w + col;
      out[idx * 3 + 0] = fmaxf(1.0f * rgb0[idx], 0.0f);
      out[idx * 3 + 1] = fmaxf(1.0f * rgb1[idx], 0.0f);
      out[idx * 3 + 2] = fmaxf(1.0f * rgb2[idx], 0.0f);
    }

  free(planes);
  return 0;
}

TDK_API int oracle_rcd(const float* in, float* out, int w, int h, uint32_t pattern) { return rcd_impl(in, out, w, h, pattern, NULL, NULL, NULL); }

TDK_API int oracle_rcd_planes(const float* in, float* out, int w, int h, uint32_t pattern, float* pq, float* p_diff, float* q_diff) {
  return rcd_impl(in, out, w, h, pattern, pq, p_diff, q_diff);
}
