/*
 * oracle/src/jpeg.c -- sequential JPEG encoder (CPU oracle, test infrastructure only).
 *
 * What it stands in for: the reference's Jpeg.encode is a wrapper over NVIDIA's nvjpeg (csrc/jpeg_encoder.cu:104-180:
 * nvjpegEncoderParamsSetQuality / SetOptimizedHuffman(1) / SetSamplingFactors / SetEncoding, nvjpegEncodeImage,
 * nvjpegEncodeRetrieveBitstream).  nvjpeg is a closed third-party library (CUDA toolkit, version unpinned: setup.py:33) and
 * the reference holds no JPEG bytes, so PARITY WITH nvjpeg IS UNPINNED.  What is restated here is the published algorithm
 * behind those calls -- ITU-T T.81: Annex A (FDCT, quantisation, zig-zag), Annex F.1.2 (sequential Huffman coding),
 * Annex G.1.2 (progressive, spectral selection only), Annex K.1 / K.2 (example quantisation tables, optimal code lengths),
 * Annex B (markers), JFIF 1.02 (APP0, YCbCr) and the IJG quality scaling every encoder including nvjpeg uses
 * (quality < 50: 5000 / q, else 200 - 2 q).  The output is pinned by an independent decoder instead: tests decode it with
 * libjpeg (Pillow) and compare it with libjpeg's own encoding of the same image at the same settings.
 *
 * Arithmetic contract shared with the device encoder (csrc/jpeg.hip), which must produce the same bytes:
 *   Y  = min(255, rint((0.299 R + 0.587 G) + 0.114 B))            all in float, no contraction
 *   Cb = min(255, rint(128 + ((-0.168736 R - 0.331264 G) + 0.5 B)))
 *   Cr = min(255, rint(128 + ((0.5 R - 0.418688 G) - 0.081312 B)))
 *   4:2:2 chroma sample = (c[2 j] + c[2 j + 1]) * 0.5 (exact), edge pixels replicated to whole MCUs
 *   FDCT: the Arai-Agui-Nakajima 8-point flow graph on rows, then on columns, float; its output scale
 *   8 * aan[u] * aan[v] is folded into the divisor: coefficient = clamp(rint(d * rq), -1023 (AC) .. 1023), rq = float(1 / (q aan aan 8))
 *   Huffman tables: optimal lengths per scan by the procedure of K.2, ties broken towards the larger symbol value.
 *   Progressive: SOF2, one interleaved DC scan (Ss = Se = 0, Ah = Al = 0), then one AC scan 1..63 per component; every block
 *   closes its own band (EOBRUN is always 1, symbol 0x00), so blocks are coded independently of each other.
 */
#include "common.h"

static const uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
/* T.81 Table K.1 / K.2 (natural order) */
static const uint8_t Q_LUMA[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                   69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                                   81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92,  95,  98,  112, 100, 103, 99};
static const uint8_t Q_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                     99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const double AAN[8] = {1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379};

static void scaled_table(const uint8_t* base, int quality, uint8_t* q) {
  if (quality < 1) quality = 1;
  if (quality > 100) quality = 100;
  const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
  for (int i = 0; i < 64; i++) {
    int v = (base[i] * scale + 50) / 100;
    q[i] = (uint8_t)(v < 1 ? 1 : v > 255 ? 255 : v);
  }
}

static void fdct8(float* d, int s) { /* one 8-point pass over d[0], d[s], ... (AAN flow graph) */
  const float t0 = d[0] + d[7 * s], t7 = d[0] - d[7 * s], t1 = d[s] + d[6 * s], t6 = d[s] - d[6 * s];
  const float t2 = d[2 * s] + d[5 * s], t5 = d[2 * s] - d[5 * s], t3 = d[3 * s] + d[4 * s], t4 = d[3 * s] - d[4 * s];
  const float e0 = t0 + t3, e3 = t0 - t3, e1 = t1 + t2, e2 = t1 - t2;
  d[0] = e0 + e1;
  d[4 * s] = e0 - e1;
  const float z1 = (e2 + e3) * 0.707106781f;
  d[2 * s] = e3 + z1;
  d[6 * s] = e3 - z1;
  const float o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
  const float z5 = (o0 - o2) * 0.382683433f;
  const float z2 = 0.541196100f * o0 + z5, z4 = 1.306562965f * o2 + z5, z3 = o1 * 0.707106781f;
  const float z11 = t7 + z3, z13 = t7 - z3;
  d[5 * s] = z13 + z2;
  d[3 * s] = z13 - z2;
  d[s] = z11 + z4;
  d[7 * s] = z11 - z4;
}

/* ---- optimal Huffman code lengths (T.81 K.2, figures K.1 - K.4); freq[256] = symbol counts */
typedef struct {
  uint8_t bits[17];    /* bits[l] = number of codes of length l */
  uint8_t vals[256];   /* symbols in code order */
  int nvals;
  uint16_t code[256];
  uint8_t len[256];
} HuffTable;

static void optimal_table(const uint32_t* counts, HuffTable* t) {
  int64_t freq[257];
  int codesize[257], others[257];
  for (int i = 0; i < 256; i++) freq[i] = counts[i];
  freq[256] = 1; /* reserves the all-ones code */
  for (int i = 0; i < 257; i++) { codesize[i] = 0; others[i] = -1; }
  for (;;) {
    int c1 = -1, c2 = -1;
    int64_t v = INT64_MAX;
    for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
    v = INT64_MAX;
    for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
    if (c2 < 0) break;
    freq[c1] += freq[c2];
    freq[c2] = 0;
    codesize[c1]++;
    while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
    others[c1] = c2;
    codesize[c2]++;
    while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
  }
  int bits[64] = {0};
  for (int i = 0; i <= 256; i++) if (codesize[i]) bits[codesize[i] < 63 ? codesize[i] : 63]++;
  for (int i = 63; i > 16; i--) {
    while (bits[i] > 0) {
      int j = i - 2;
      while (bits[j] == 0) j--;
      bits[i] -= 2;
      bits[i - 1]++;
      bits[j + 1] += 2;
      bits[j]--;
    }
  }
  int i = 16;
  while (bits[i] == 0) i--;
  bits[i]--; /* the reserved code point */
  memset(t, 0, sizeof *t);
  for (int l = 1; l <= 16; l++) t->bits[l] = (uint8_t)bits[l];
  int n = 0;
  for (int l = 1; l <= 63; l++)
    for (int s = 0; s < 256; s++)
      if (codesize[s] == l) t->vals[n++] = (uint8_t)s;
  t->nvals = n;
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    for (int j = 0; j < t->bits[l]; j++, k++) { t->code[t->vals[k]] = (uint16_t)code++; t->len[t->vals[k]] = (uint8_t)l; }
    code <<= 1;
  }
}

/* ---- byte sink with marker helpers and the entropy-coded bit writer (byte stuffing inline) */
typedef struct {
  uint8_t* p;
  size_t n, cap;
  uint32_t acc;
  int nb;
  int overflow;
} Sink;
static void put8(Sink* s, int v) { if (s->n < s->cap) s->p[s->n] = (uint8_t)v; else s->overflow = 1; s->n++; }
static void put16(Sink* s, int v) { put8(s, v >> 8); put8(s, v & 255); }
static void put_bits(Sink* s, uint32_t code, int len) {
  for (int i = len - 1; i >= 0; i--) {
    s->acc = (s->acc << 1) | ((code >> i) & 1u);
    if (++s->nb == 8) {
      put8(s, (int)s->acc);
      if (s->acc == 0xff) put8(s, 0);
      s->acc = 0;
      s->nb = 0;
    }
  }
}
static void flush_bits(Sink* s) { if (s->nb) put_bits(s, (1u << (8 - s->nb)) - 1u, 8 - s->nb); }

static int nbits_of(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { n++; a >>= 1; } return n; }

/* one block's band [ss, se] in zig-zag order: counts symbols (sink == NULL) or writes them */
static void code_block(const int16_t* zz, int ss, int se, int pred_dc, uint32_t* dc_hist, uint32_t* ac_hist, const HuffTable* dct,
                       const HuffTable* act, Sink* sink) {
  if (ss == 0) {
    const int diff = zz[0] - pred_dc, s = nbits_of(diff);
    if (sink) { put_bits(sink, dct->code[s], dct->len[s]); if (s) put_bits(sink, (uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << s) - 1u), s); }
    else dc_hist[s]++;
    ss = 1;
  }
  if (se == 0) return;
  int run = 0;
  for (int k = ss; k <= se; k++) {
    const int v = zz[k];
    if (v == 0) { run++; continue; }
    while (run > 15) { if (sink) put_bits(sink, act->code[0xf0], act->len[0xf0]); else ac_hist[0xf0]++; run -= 16; }
    const int s = nbits_of(v), sym = (run << 4) | s;
    if (sink) { put_bits(sink, act->code[sym], act->len[sym]); put_bits(sink, (uint32_t)(v < 0 ? v - 1 : v) & ((1u << s) - 1u), s); }
    else ac_hist[sym]++;
    run = 0;
  }
  if (run > 0) { if (sink) put_bits(sink, act->code[0], act->len[0]); else ac_hist[0]++; }
}

static void put_dht(Sink* s, int cls, int id, const HuffTable* t) {
  put16(s, 0xffc4);
  put16(s, 2 + 1 + 16 + t->nvals);
  put8(s, (cls << 4) | id);
  for (int l = 1; l <= 16; l++) put8(s, t->bits[l]);
  for (int i = 0; i < t->nvals; i++) put8(s, t->vals[i]);
}

typedef struct {
  int ncomp, hs[3], nbx[3], nby, nbx_real[3], nby_real, nmcux, nmcuy;
  int16_t* coef[3]; /* per component: nbx * nby blocks of 64 zig-zag coefficients */
} Frame;

/* one scan: comps[0..ns), band [ss, se]; interleaved when ns > 1 */
static void encode_scan(const Frame* f, int ns, const int* comps, int ss, int se, Sink* out) {
  uint32_t dch[2][256], ach[2][256];
  HuffTable dct[2], act[2];
  memset(dch, 0, sizeof dch);
  memset(ach, 0, sizeof ach);
  for (int pass = 0; pass < 2; pass++) {
    Sink* sink = pass ? out : NULL;
    int pred[3] = {0, 0, 0};
    if (ns > 1) {
      for (int my = 0; my < f->nmcuy; my++)
        for (int mx = 0; mx < f->nmcux; mx++)
          for (int i = 0; i < ns; i++) {
            const int c = comps[i], tb = c ? 1 : 0;
            for (int j = 0; j < f->hs[c]; j++) {
              const int16_t* zz = f->coef[c] + ((size_t)my * f->nbx[c] + (size_t)mx * f->hs[c] + j) * 64;
              code_block(zz, ss, se, pred[c], dch[tb], ach[tb], &dct[tb], &act[tb], sink);
              pred[c] = zz[0];
            }
          }
    } else {
      const int c = comps[0], tb = c ? 1 : 0;
      for (int by = 0; by < f->nby_real; by++)
        for (int bx = 0; bx < f->nbx_real[c]; bx++) {
          const int16_t* zz = f->coef[c] + ((size_t)by * f->nbx[c] + bx) * 64;
          code_block(zz, ss, se, pred[c], dch[tb], ach[tb], &dct[tb], &act[tb], sink);
          pred[c] = zz[0];
        }
    }
    if (pass == 0) {
      int used[2] = {0, 0};
      for (int i = 0; i < ns; i++) used[comps[i] ? 1 : 0] = 1;
      for (int tb = 0; tb < 2; tb++) {
        if (!used[tb]) continue;
        if (ss == 0) { optimal_table(dch[tb], &dct[tb]); put_dht(out, 0, tb, &dct[tb]); }
        if (se > 0) { optimal_table(ach[tb], &act[tb]); put_dht(out, 1, tb, &act[tb]); }
      }
      put16(out, 0xffda);
      put16(out, 6 + 2 * ns);
      put8(out, ns);
      for (int i = 0; i < ns; i++) { put8(out, comps[i] + 1); put8(out, comps[i] ? 0x11 : 0x00); }
      put8(out, ss);
      put8(out, se);
      put8(out, 0);
    }
  }
  flush_bits(out);
}

/* input_format: 0 BGR planar, 1 RGB planar, 2 BGR interleaved, 3 RGB interleaved (reference csrc/jpeg_encoder.h enum order);
 * subsampling: 0 = 4:4:4, 1 = 4:2:2, 2 = gray.  Returns the stream length (also when it exceeds `cap`: nothing is written beyond it). */
TDK_API int64_t oracle_jpeg_encode(const uint8_t* img, int w, int h, int input_format, int quality, int subsampling, int progressive,
                                   uint8_t* out, int64_t cap, int16_t* coef_out) {
  Frame f;
  memset(&f, 0, sizeof f);
  f.ncomp = subsampling == 2 ? 1 : 3;
  f.hs[0] = subsampling == 1 ? 2 : 1;
  f.hs[1] = f.hs[2] = 1;
  const int mcuw = 8 * f.hs[0];
  f.nmcux = (w + mcuw - 1) / mcuw;
  f.nmcuy = (h + 7) / 8;
  f.nby = f.nmcuy;
  f.nby_real = (h + 7) / 8;
  for (int c = 0; c < f.ncomp; c++) {
    f.nbx[c] = f.nmcux * f.hs[c];
    const int wc = c == 0 ? w : (w * f.hs[c] + f.hs[0] - 1) / f.hs[0];
    f.nbx_real[c] = (wc + 7) / 8;
    f.coef[c] = (int16_t*)malloc((size_t)f.nbx[c] * f.nby * 64 * sizeof(int16_t));
  }
  uint8_t qt[2][64];
  float rq[2][64];
  scaled_table(Q_LUMA, quality, qt[0]);
  scaled_table(Q_CHROMA, quality, qt[1]);
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 64; i++) rq[t][i] = (float)(1.0 / ((double)qt[t][i] * AAN[i >> 3] * AAN[i & 7] * 8.0));

  const int planar = input_format < 2, bgr = (input_format & 1) == 0;
  const size_t plane = (size_t)w * h;
  const int pw = f.nmcux * mcuw; /* padded width */
  int* ycc = (int*)malloc((size_t)3 * pw * 8 * sizeof(int)); /* one MCU row: Y, Cb, Cr lines */
  for (int my = 0; my < f.nmcuy; my++) {
    for (int r = 0; r < 8; r++) {
      const int y = my * 8 + r < h ? my * 8 + r : h - 1;
      for (int x = 0; x < pw; x++) {
        const int xs = x < w ? x : w - 1;
        float c0, c1, c2;
        if (planar) { c0 = img[(size_t)y * w + xs]; c1 = img[plane + (size_t)y * w + xs]; c2 = img[2 * plane + (size_t)y * w + xs]; }
        else { const uint8_t* p = img + ((size_t)y * w + xs) * 3; c0 = p[0]; c1 = p[1]; c2 = p[2]; }
        const float R = bgr ? c2 : c0, G = c1, B = bgr ? c0 : c2;
        ycc[(0 * 8 + r) * pw + x] = (int)fminf(255.0f, rintf((0.299f * R + 0.587f * G) + 0.114f * B));
        ycc[(1 * 8 + r) * pw + x] = (int)fminf(255.0f, rintf(128.0f + ((-0.168736f * R - 0.331264f * G) + 0.5f * B)));
        ycc[(2 * 8 + r) * pw + x] = (int)fminf(255.0f, rintf(128.0f + ((0.5f * R - 0.418688f * G) - 0.081312f * B)));
      }
    }
    for (int c = 0; c < f.ncomp; c++) {
      const int sub = f.hs[0] / f.hs[c]; /* horizontal decimation of this component */
      for (int bx = 0; bx < f.nbx[c]; bx++) {
        float d[64];
        for (int r = 0; r < 8; r++)
          for (int x = 0; x < 8; x++) {
            const int* line = ycc + (c * 8 + r) * pw;
            const int sx = (bx * 8 + x) * sub;
            d[r * 8 + x] = sub == 2 ? (float)(line[sx] + line[sx + 1]) * 0.5f - 128.0f : (float)line[sx] - 128.0f;
          }
        for (int r = 0; r < 8; r++) fdct8(d + 8 * r, 1);
        for (int x = 0; x < 8; x++) fdct8(d + x, 8);
        int16_t* zz = f.coef[c] + ((size_t)my * f.nbx[c] + bx) * 64;
        const float* q = rq[c ? 1 : 0];
        for (int k = 0; k < 64; k++) {
          const float v = rintf(d[ZZ[k]] * q[ZZ[k]]);
          zz[k] = (int16_t)fminf(1023.0f, fmaxf(k ? -1023.0f : -1024.0f, v));
        }
      }
    }
  }
  free(ycc);
  if (coef_out) { /* test hook: component planes back to back */
    size_t o = 0;
    for (int c = 0; c < f.ncomp; c++) { const size_t n = (size_t)f.nbx[c] * f.nby * 64; memcpy(coef_out + o, f.coef[c], n * sizeof(int16_t)); o += n; }
  }

  Sink s = {out, 0, cap > 0 ? (size_t)cap : 0, 0, 0, 0};
  put16(&s, 0xffd8);
  put16(&s, 0xffe0); put16(&s, 16);
  put8(&s, 'J'); put8(&s, 'F'); put8(&s, 'I'); put8(&s, 'F'); put8(&s, 0);
  put16(&s, 0x0101); put8(&s, 0); put16(&s, 1); put16(&s, 1); put8(&s, 0); put8(&s, 0);
  for (int t = 0; t < (f.ncomp == 1 ? 1 : 2); t++) {
    put16(&s, 0xffdb); put16(&s, 67); put8(&s, t);
    for (int k = 0; k < 64; k++) put8(&s, qt[t][ZZ[k]]);
  }
  put16(&s, progressive ? 0xffc2 : 0xffc0);
  put16(&s, 8 + 3 * f.ncomp); put8(&s, 8); put16(&s, h); put16(&s, w); put8(&s, f.ncomp);
  for (int c = 0; c < f.ncomp; c++) { put8(&s, c + 1); put8(&s, (f.hs[c] << 4) | 1); put8(&s, c ? 1 : 0); }
  const int all[3] = {0, 1, 2};
  if (!progressive) encode_scan(&f, f.ncomp, all, 0, 63, &s);
  else {
    encode_scan(&f, f.ncomp, all, 0, 0, &s);
    for (int c = 0; c < f.ncomp; c++) encode_scan(&f, 1, &c, 1, 63, &s);
  }
  put16(&s, 0xffd9);
  for (int c = 0; c < f.ncomp; c++) free(f.coef[c]);
  return (int64_t)s.n;
}
