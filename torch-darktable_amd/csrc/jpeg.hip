// jpeg.hip -- device JPEG encoder: uint8 RGB/BGR image in HBM -> JFIF byte stream.
//
// Replaces the reference's nvjpeg wrapper (csrc/jpeg_encoder.cu:104-180: nvjpegEncodeImage + nvjpegEncodeRetrieveBitstream with
// SetQuality, SetOptimizedHuffman(1), SetSamplingFactors(444 / 422 / GRAY), baseline or progressive encoding).  nvjpeg is a closed
// library; what is implemented is the published algorithm behind those calls (ITU-T T.81 Annex A, B, F.1.2, G.1.2, K.1, K.2; JFIF
// 1.02; IJG quality scaling), with the arithmetic contract written out in oracle/src/jpeg.c, whose bytes this file reproduces.
//
// MI355X design: five passes over data that stays in HBM, no per-symbol host work.
//  1. jpeg_fdct_kernel: a workgroup stages an 8-row strip of 16 - 256 MCUs (coalesced dword reads of the interleaved or planar
//     image, colour conversion on the way) into LDS as int16 Y / Cb / Cr samples laid out sample-major ([y][x][block]: the DCT
//     phase reads conflict-free), then ONE THREAD PER 8x8 BLOCK runs the Arai-Agui-Nakajima flow graph on 64 registers (rows,
//     columns), quantises by a reciprocal table held in SGPRs (kernel argument, wave-uniform per component) and stores the block's
//     64 int16 coefficients in zig-zag order as one 128-byte line.
//  2. jpeg_code_kernel<HIST>: one thread per block of the scan walks its 64 coefficients (a fully unrolled loop over registers) and
//     counts the Huffman symbols into LDS histograms; the host turns the four histograms into optimal code lengths (K.2) -- the one
//     host round trip per scan, 4 KB each way.
//  3. jpeg_code_kernel<LEN>: the same walk adds up the block's bit length; a workgroup sum per 256 blocks, one single-workgroup
//     prefix sum over those (jpeg_scan_kernel), and
//  4. jpeg_code_kernel<WRITE>: the same walk a third time shifts the codes into a 64-bit accumulator and stores big-endian dwords at
//     the block's bit offset (atomicOr on the first and last dword it shares with its neighbours, plain stores in between).
//  5. byte stuffing (0xFF -> 0xFF 0x00) as count / prefix sum / scatter over 4 KB chunks, straight into the final stream behind
//     the headers the host has written there.
// The coefficients (128 B per block, groups of 64 blocks stored transposed so that a wave's accesses are contiguous segments: 50 MB for
// a 12 MP 4:2:2 frame) are read three times -- cheaper than keeping symbol lists.
// Progressive = SOF2 with one interleaved DC scan and one AC scan (1..63) per component; every block closes its own band
// (EOBRUN = 1), so the block walks stay independent.
#include "tdk_common.h"

#include <string.h>

#include <type_traits>

#include <vector>

namespace {

// natural index of zig-zag position k (a function so that device code can use it: after unrolling every call folds to a constant)
__host__ __device__ constexpr int zz_of(int k) {
  constexpr int t[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return t[k];
}
// T.81 tables K.1 / K.2, natural order
constexpr uint8_t Q_LUMA[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                                81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92,  95,  98,  112, 100, 103, 99};
constexpr uint8_t Q_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                  99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
constexpr double AAN[8] = {1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379};

struct Geo {
  int w, h, ncomp, hs0;        // hs0: horizontal sampling factor of Y (2 for 4:2:2), chroma is 1
  int nmcux, nmcuy;
  int nbx[3], nbx_real[3];     // blocks per row of a component plane: padded to whole MCUs / the ones a non-interleaved scan codes
  int nby_real;
  long long coff[3];           // first block of the component's plane in the coefficient array
  long long nblocks;           // all planes
  int planar, bgr;
};
struct Quant { float rq[2][64]; };   // 1 / (q * aan[v] * aan[u] * 8), natural order; [0] luma, [1] chroma
struct ScanDesc {
  int ns, comp0, ss, se;       // ns components starting with comp0 (ns > 1: all of them, interleaved)
  long long nscan;             // blocks the scan codes
  int chunk;                   // scan positions per workgroup of the coding passes: 256, or 64 MCUs of an interleaved scan (256 at 4:2:2, 192 at 4:4:4)
};

// Coefficient store: block idx keeps its 64 zig-zag int16 as 32 dwords, dword j of block idx at ((idx / 64) * 32 + j) * 64 + idx % 64
// -- groups of 64 blocks, transposed.  A wave = 64 blocks of a scan (runs of consecutive blocks of one component) then reads or
// writes dword j of all its blocks as a few contiguous segments instead of 64 different 128-byte lines.
__host__ __device__ __forceinline__ size_t coef_word(long long idx, int j) { return ((size_t)(idx >> 6) * 32 + j) * 64 + (size_t)(idx & 63); }

// ------------------------------------------------------------------ pass 1: colour conversion + FDCT + quantisation
__device__ __forceinline__ void fdct8(float& d0, float& d1, float& d2, float& d3, float& d4, float& d5, float& d6, float& d7) {
  const float t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6, t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
  const float e0 = t0 + t3, e3 = t0 - t3, e1 = t1 + t2, e2 = t1 - t2;
  d0 = e0 + e1;
  d4 = e0 - e1;
  const float z1 = (e2 + e3) * 0.707106781f;
  d2 = e3 + z1;
  d6 = e3 - z1;
  const float o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
  const float z5 = (o0 - o2) * 0.382683433f;
  const float z2 = 0.541196100f * o0 + z5, z4 = 1.306562965f * o2 + z5, z3 = o1 * 0.707106781f;
  const float z11 = t7 + z3, z13 = t7 - z3;
  d5 = z13 + z2;
  d3 = z13 - z2;
  d1 = z11 + z4;
  d7 = z11 - z4;
}

__device__ __forceinline__ void ycc_of(float c0, float c1, float c2, bool bgr, int& y, int& cb, int& cr) {
  const float R = bgr ? c2 : c0, G = c1, B = bgr ? c0 : c2;
  y = (int)fminf(255.0f, rintf((0.299f * R + 0.587f * G) + 0.114f * B));
  cb = (int)fminf(255.0f, rintf(128.0f + ((-0.168736f * R - 0.331264f * G) + 0.5f * B)));
  cr = (int)fminf(255.0f, rintf(128.0f + ((0.5f * R - 0.418688f * G) - 0.081312f * B)));
}

// SUB: 0 = 4:4:4, 1 = 4:2:2, 2 = gray.  TPX pixels of an 8-row strip per workgroup.
template <int SUB> struct Tile {
  static constexpr int TPX = SUB == 0 ? 512 : SUB == 1 ? 1024 : 2048;
  static constexpr int NYB = TPX / 8;                              // Y blocks
  static constexpr int NCB = SUB == 0 ? TPX / 8 : SUB == 1 ? TPX / 16 : 0;  // blocks per chroma component
  // Y samples are bytes; chroma samples bytes too, except 4:2:2 where a site holds the SUM of its two pixels (<= 510): int16.  25 KB
  // per workgroup at 4:2:2 (was 39 KB with int16 planes and wider pitches): six workgroups = 24 waves per CU instead of four.
  using CT = typename std::conditional<SUB == 1, int16_t, uint8_t>::type;
  static constexpr int PY = NYB + 4, PC = NCB + 4;                 // plane pitches (entries per sample position)
  static constexpr int Y_BYTES = 64 * PY, C_BYTES = NCB ? 64 * PC * (int)sizeof(CT) : 0;
  static constexpr int LDS_BYTES = Y_BYTES + 2 * C_BYTES;
  static_assert(NYB + 2 * NCB <= 256, "one thread per block");
  static_assert(Y_BYTES % 4 == 0, "the chroma planes start aligned");
};

template <int SUB>
__global__ __launch_bounds__(256) void jpeg_fdct_kernel(const uint8_t* __restrict__ img, uint32_t* __restrict__ coef, Geo g, Quant qt) {
  using T = Tile<SUB>;
  using CT = typename T::CT;
  extern __shared__ __align__(16) uint8_t jl[];
  uint8_t* sY = jl;
  CT* sCb = reinterpret_cast<CT*>(jl + T::Y_BYTES);
  CT* sCr = sCb + 64 * T::PC;
  const int tile = blockIdx.x, my = blockIdx.y;
  const int x_tile = tile * T::TPX;
  const size_t plane = (size_t)g.w * g.h;
  const bool bgr = g.bgr != 0;

  // ---- stage: groups of four pixels
#pragma unroll 4
  for (int i = threadIdx.x; i < 8 * (T::TPX / 4); i += 256) {
    const int r = i / (T::TPX / 4), gx = i - r * (T::TPX / 4);
    const int px0 = x_tile + gx * 4;
    const int y = min(my * 8 + r, g.h - 1);
    int yv[4], cbv[4], crv[4];
    if (g.planar) {
      const uint8_t* p0 = img + (size_t)y * g.w;
      if (px0 + 3 < g.w && ((((uintptr_t)p0 + px0) | plane) & 3) == 0) {
        const uint32_t a = *reinterpret_cast<const uint32_t*>(p0 + px0), b = *reinterpret_cast<const uint32_t*>(p0 + plane + px0),
                       c = *reinterpret_cast<const uint32_t*>(p0 + 2 * plane + px0);
#pragma unroll
        for (int k = 0; k < 4; k++) ycc_of((float)((a >> (8 * k)) & 255u), (float)((b >> (8 * k)) & 255u), (float)((c >> (8 * k)) & 255u), bgr, yv[k], cbv[k], crv[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int xs = min(px0 + k, g.w - 1);
          ycc_of((float)p0[xs], (float)p0[plane + xs], (float)p0[2 * plane + xs], bgr, yv[k], cbv[k], crv[k]);
        }
      }
    } else {
      const uint8_t* p0 = img + (size_t)y * g.w * 3;
      if (px0 + 3 < g.w && (((uintptr_t)p0 + (size_t)px0 * 3) & 3) == 0) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p0 + (size_t)px0 * 3);
        const uint32_t a = q[0], b = q[1], c = q[2];
        ycc_of((float)(a & 255u), (float)((a >> 8) & 255u), (float)((a >> 16) & 255u), bgr, yv[0], cbv[0], crv[0]);
        ycc_of((float)(a >> 24), (float)(b & 255u), (float)((b >> 8) & 255u), bgr, yv[1], cbv[1], crv[1]);
        ycc_of((float)((b >> 16) & 255u), (float)(b >> 24), (float)(c & 255u), bgr, yv[2], cbv[2], crv[2]);
        ycc_of((float)((c >> 8) & 255u), (float)((c >> 16) & 255u), (float)(c >> 24), bgr, yv[3], cbv[3], crv[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint8_t* p = p0 + (size_t)min(px0 + k, g.w - 1) * 3;
          ycc_of((float)p[0], (float)p[1], (float)p[2], bgr, yv[k], cbv[k], crv[k]);
        }
      }
    }
    const int lx = gx * 4;  // pixel column inside the tile
#pragma unroll
    for (int k = 0; k < 4; k++) sY[(r * 8 + ((lx + k) & 7)) * T::PY + ((lx + k) >> 3)] = (uint8_t)yv[k];
    if constexpr (SUB == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sCb[(r * 8 + ((lx + k) & 7)) * T::PC + ((lx + k) >> 3)] = (CT)cbv[k];
        sCr[(r * 8 + ((lx + k) & 7)) * T::PC + ((lx + k) >> 3)] = (CT)crv[k];
      }
    } else if constexpr (SUB == 1) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int cx = lx / 2 + k;
        sCb[(r * 8 + (cx & 7)) * T::PC + (cx >> 3)] = (CT)(cbv[2 * k] + cbv[2 * k + 1]);
        sCr[(r * 8 + (cx & 7)) * T::PC + (cx >> 3)] = (CT)(crv[2 * k] + crv[2 * k + 1]);
      }
    }
  }
  __syncthreads();

  // ---- one thread per block
  const int t = threadIdx.x;
  int c, lb;  // component, block inside the tile's share of that component
  if (t < T::NYB) { c = 0; lb = t; }
  else if (t < T::NYB + T::NCB) { c = 1; lb = t - T::NYB; }
  else if (t < T::NYB + 2 * T::NCB) { c = 2; lb = t - T::NYB - T::NCB; }
  else return;
  const int per_tile = c == 0 ? T::NYB : T::NCB;
  const int bx = tile * per_tile + lb;
  if (bx >= g.nbx[c]) return;
  const bool halved = SUB == 1 && c != 0;
  float d[64];
  if (c == 0) {
#pragma unroll
    for (int i = 0; i < 64; i++) d[i] = (float)sY[i * T::PY + lb] - 128.0f;
  } else {
    const CT* src = c == 1 ? sCb : sCr;
#pragma unroll
    for (int i = 0; i < 64; i++) {
      const float s = (float)src[i * T::PC + lb];
      d[i] = halved ? s * 0.5f - 128.0f : s - 128.0f;
    }
  }
#pragma unroll
  for (int r = 0; r < 8; r++) fdct8(d[8 * r], d[8 * r + 1], d[8 * r + 2], d[8 * r + 3], d[8 * r + 4], d[8 * r + 5], d[8 * r + 6], d[8 * r + 7]);
#pragma unroll
  for (int x = 0; x < 8; x++) fdct8(d[x], d[8 + x], d[16 + x], d[24 + x], d[32 + x], d[40 + x], d[48 + x], d[56 + x]);
  const float* rq = qt.rq[__builtin_amdgcn_readfirstlane(c ? 1 : 0)];  // a wave holds blocks of one component
  uint32_t packed[32];
#pragma unroll
  for (int k = 0; k < 64; k++) {
    const float v = fminf(1023.0f, fmaxf(k ? -1023.0f : -1024.0f, rintf(d[zz_of(k)] * rq[zz_of(k)])));
    const uint32_t u = (uint32_t)(int)v & 0xffffu;
    if (k & 1) packed[k >> 1] |= u << 16;
    else packed[k >> 1] = u;
  }
  const long long idx = g.coff[c] + (long long)my * g.nbx[c] + bx;
#pragma unroll
  for (int j = 0; j < 32; j++) coef[coef_word(idx, j)] = packed[j];
}

// ------------------------------------------------------------------ passes 2 - 4: the block walk
enum { HIST = 0, LEN = 1, WRITE = 2 };

// Bit sink of one block.  The dwords go either into the workgroup's LDS staging buffer (the usual case: LDS atomics on the dword a
// block shares with its neighbour, plain LDS stores in between) or, for a workgroup whose segment does not fit it, straight into
// the global stream (global atomicOr on the shared dwords: two per block, the price the staging buffer avoids -- 786 K atomics per
// 12 MP frame made the pass 37 us instead of 22, profiles/r05/experiments/jpeg_write_atomics.txt).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct BitWriter {
  uint32_t* gbuf;      // big-endian dwords of the unstuffed scan (global path)
  lds_u32* lbuf;       // the workgroup's staging buffer (LDS path), dword 0 = the dword that holds the workgroup's first bit
  bool to_lds;
  uint64_t acc;
  int n;               // valid low bits of acc
  size_t word;
  bool shared;         // the next dword to flush is shared with the previous block
  __device__ __forceinline__ void start(uint64_t bitpos) {
    word = (size_t)(bitpos >> 5);
    n = (int)(bitpos & 31);
    acc = 0;
    shared = n != 0;
  }
  __device__ __forceinline__ void flush(uint32_t wv, bool atomic) {
    if (to_lds) {
      if (atomic) __hip_atomic_fetch_or(lbuf + word, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else lbuf[word] = wv;
    } else {
      if (atomic) atomicOr(gbuf + word, wv);
      else gbuf[word] = wv;
    }
  }
  __device__ __forceinline__ void put(uint32_t code, int len) {
    acc = (acc << len) | code;
    n += len;
    if (n >= 32) {
      flush((uint32_t)(acc >> (n - 32)), shared);
      shared = false;
      word++;
      n -= 32;
      acc &= (1ull << n) - 1ull;
    }
  }
  __device__ __forceinline__ void finish() {
    if (n > 0) flush((uint32_t)(acc << (32 - n)), true);
  }
};

__device__ __forceinline__ int nbits_of(int v) { return 32 - __clz(v < 0 ? -v : v); }

// scan position -> block index in the coefficient array (and the block that predicts its DC, or -1)
__device__ __forceinline__ void locate(const Geo& g, const ScanDesc& sc, long long s, int& c, long long& idx, long long& prev) {
  if (sc.ns > 1) {
    const int bpm = g.hs0 + 2;
    const long long mcu = s / bpm;
    const int j = (int)(s - mcu * bpm);
    const long long my = mcu / g.nmcux;
    const int mx = (int)(mcu - my * g.nmcux);
    int bx;
    if (j < g.hs0) { c = 0; bx = mx * g.hs0 + j; }
    else { c = j - g.hs0 + 1; bx = mx; }
    idx = g.coff[c] + my * g.nbx[c] + bx;
    prev = idx == g.coff[c] ? -1 : idx - 1;   // planes are padded to whole MCUs: the scan order of a component is its raster order
  } else {
    c = sc.comp0;
    const int nr = g.nbx_real[c];
    const long long by = s / nr;
    const int bx = (int)(s - by * nr);
    idx = g.coff[c] + by * g.nbx[c] + bx;
    prev = s == 0 ? -1 : bx > 0 ? idx - 1 : g.coff[c] + (by - 1) * g.nbx[c] + nr - 1;
  }
}

// tabs: [dc0, dc1, ac0, ac1][256] = code << 8 | length.  hist: the same four tables of counts.
template <int MODE>
__global__ __launch_bounds__(256) void jpeg_code_kernel(const uint32_t* __restrict__ coef, Geo g, ScanDesc sc, uint32_t* __restrict__ hist,
                                                         const uint32_t* __restrict__ tabs, uint16_t* __restrict__ lens,
                                                         uint32_t* __restrict__ wgsum, const uint64_t* __restrict__ wgoff, uint32_t* __restrict__ raw) {
  // HIST: eight copies of the four histograms, a lane counts into copy (lane & 7); the copies start 1025 words apart, so the same
  // symbol in different copies sits in different banks (most lanes of a wave emit one of a handful of symbols: with one copy the
  // LDS atomics of a wave instruction serialise on those few addresses)
  constexpr int COPIES = MODE == HIST ? 8 : 1, CSTRIDE = 1025;
  __shared__ uint32_t lt[MODE == HIST ? COPIES * CSTRIDE : 1024];
  __shared__ uint32_t wsum[4];
  constexpr int LCAP = 4096;                         // dwords of the WRITE pass's staging buffer (a 12 MP frame at quality 94: ~260 per workgroup)
  __shared__ uint32_t seg[MODE == WRITE ? LCAP : 1];
  __shared__ uint32_t pex[MODE == WRITE ? 256 : 1];
  if (MODE == HIST) for (int i = threadIdx.x; i < COPIES * CSTRIDE; i += 256) lt[i] = 0u;
  else for (int i = threadIdx.x; i < 1024; i += 256) lt[i] = tabs[i];
  __syncthreads();
  const int copy_base = MODE == HIST ? (int)(threadIdx.x & (COPIES - 1)) * CSTRIDE : 0;

  // HIST runs a fixed grid over all chunks of 256 blocks (fewer workgroups = fewer same-address global atomics when the LDS
  // histograms are flushed); LEN and WRITE run one workgroup per chunk (their prefix sums are per chunk)
  // Lane -> scan position inside the chunk.  An interleaved scan alternates the components (Y0 Y1 Cb Cr per MCU): taken in scan order
  // the lanes of a wave would pick their coefficient dwords from three planes in runs of one or two blocks.  Instead the chunk's 64
  // MCUs are taken component by component -- lanes 0 .. 64 hs - 1 the Y blocks, the next 64 Cb, the next 64 Cr -- so that a wave reads
  // 64 CONSECUTIVE blocks of one plane (a full 256-byte row of the transposed store per load).  Lengths and offsets stay indexed by
  // scan position; the WRITE pass's prefix sum runs over the chunk in scan order through LDS.
  const int CH = sc.chunk;
  int ls = threadIdx.x;      // local scan position
  if (sc.ns > 1) {
    const int bpm = g.hs0 + 2, ny = 64 * g.hs0, t = threadIdx.x;
    if (t < ny) ls = (t / g.hs0) * bpm + (t % g.hs0);
    else if (t < ny + 128) ls = ((t - ny) & 63) * bpm + g.hs0 + ((t - ny) >> 6);
    else ls = CH;            // idle lane (4:4:4: 192 positions per chunk)
  }
  const long long nchunks = (sc.nscan + CH - 1) / CH;
  for (long long chunk = blockIdx.x; chunk < nchunks; chunk += (MODE == HIST ? (long long)gridDim.x : nchunks)) {
  const long long s = chunk * CH + ls;
  const bool live = ls < CH && s < sc.nscan;
  uint32_t nbits_total = 0;
  BitWriter bw;
  size_t w0 = 0, nw = 0;       // WRITE: first dword of the workgroup's segment in the stream, dwords it touches
  unsigned long long o0 = 0, o1 = 0;
  if (MODE == WRITE) {
    // exclusive prefix of this workgroup's block lengths, in scan order: thread i scans position i of the chunk
    const long long si = chunk * CH + threadIdx.x;
    const uint32_t len_i = ((int)threadIdx.x < CH && si < sc.nscan) ? lens[si] : 0u;
    uint32_t incl = len_i;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o, 64);
      if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    o0 = wgoff[blockIdx.x];
    o1 = wgoff[blockIdx.x + 1];
    if (chunk == nchunks - 1) o1 = (o1 + 7) & ~7ull;   // the scan's last block pads the last byte with ones
    w0 = (size_t)(o0 >> 5);
    nw = (size_t)((o1 + 31) >> 5) - w0;
    bw.to_lds = nw <= (size_t)LCAP;
    bw.gbuf = raw;
    bw.lbuf = (lds_u32*)seg;
    if (bw.to_lds) {
      for (size_t i = threadIdx.x; i < nw; i += 256) seg[i] = 0u;
    } else {
      // the segment is larger than the staging buffer (noise at quality 100): blocks write the stream directly.  The dwords only this
      // workgroup touches are cleared here; the two it may share with its neighbours were cleared by jpeg_scan_kernel.
      const size_t end = (o1 & 31) ? nw - 1 : nw;   // a last dword that ends on the boundary is this workgroup's alone
      for (size_t i = 1 + threadIdx.x; i < end; i += 256) raw[w0 + i] = 0u;
      __threadfence();
    }
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < (int)(threadIdx.x >> 6); k++) before += wsum[k];
    pex[threadIdx.x] = before + incl - len_i;   // exclusive prefix of scan position threadIdx.x
    __syncthreads();
    const unsigned long long bitpos = o0 + (live ? pex[ls] : 0u);
    bw.start(bw.to_lds ? bitpos - ((unsigned long long)w0 << 5) : bitpos);
  }

  if (live) {
    int c;
    long long idx, prev;
    locate(g, sc, s, c, idx, prev);
    const int tb = c ? 1 : 0;
    uint32_t w[32];
#pragma unroll
    for (int j = 0; j < 32; j++) w[j] = coef[coef_word(idx, j)];

    auto emit = [&](int table, int sym, uint32_t extra, int nextra) {
      if (MODE == HIST) atomicAdd(&lt[copy_base + table * 256 + sym], 1u);
      else {
        const uint32_t e = lt[table * 256 + sym];
        if (MODE == LEN) nbits_total += (e & 255u) + nextra;
        else bw.put(((e >> 8) << nextra) | extra, (int)(e & 255u) + nextra);   // code and amplitude bits in one go: <= 16 + 11 bits
      }
    };

    if (sc.ss == 0) {
      const int dc = (int)(int16_t)(w[0] & 0xffffu);
      const int pred = prev < 0 ? 0 : (int)(int16_t)(coef[coef_word(prev, 0)] & 0xffffu);
      const int diff = dc - pred, sz = nbits_of(diff);
      emit(tb, sz, (uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << sz) - 1u), sz);
    }
    if (sc.se > 0) {
      int run = 0;
#pragma unroll
      for (int k = 1; k < 64; k++) {
        const int v = (int)(int16_t)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu));
        if (v == 0) run++;
        else {
          while (run > 15) { emit(2 + tb, 0xf0, 0u, 0); run -= 16; }
          const int sz = nbits_of(v);
          emit(2 + tb, (run << 4) | sz, (uint32_t)(v < 0 ? v - 1 : v) & ((1u << sz) - 1u), sz);
          run = 0;
        }
      }
      if (run > 0) emit(2 + tb, 0, 0u, 0);
    }
  }

  if (MODE == HIST) continue;
  if (MODE == LEN) {
    if (live) lens[s] = (uint16_t)nbits_total;
    uint32_t v = nbits_total;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) wgsum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  } else {
    if (live) {
      if (s == sc.nscan - 1) {  // close the scan: pad the last byte with ones
        const int used = (int)((bw.word * 32 + bw.n) & 7);
        if (used) bw.put((1u << (8 - used)) - 1u, 8 - used);
      }
      bw.finish();
    }
    if (bw.to_lds) {  // the staged segment to the stream: coalesced stores, an atomic only where a neighbouring workgroup shares the dword
      __syncthreads();
      for (size_t i = threadIdx.x; i < nw; i += 256) {
        const uint32_t v = seg[i];
        const bool shared_dword = (i == 0 && (o0 & 31)) || (i == nw - 1 && (o1 & 31));
        if (shared_dword) atomicOr(raw + w0 + i, v);
        else raw[w0 + i] = v;
      }
    }
  }
  }  // chunks
  if (MODE == HIST) {
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) {
      uint32_t v = 0;
#pragma unroll
      for (int c = 0; c < COPIES; c++) v += lt[c * CSTRIDE + i];
      if (v) atomicAdd(hist + i, v);
    }
  }
}

// device-side scalars of a scan
struct Scal {
  unsigned long long total_bits;
  unsigned int nbytes, nchunks, total_ff, overflow;
  unsigned long long seg_len;
};

// exclusive prefix sum of n uint32 (n from the host, or from *n_dev) into uint64 out[0..n]; one workgroup of 1024.
// what: 0 = block lengths -> total_bits / nbytes / nchunks, 1 = 0xFF counts -> total_ff / seg_len
// what = 1 also closes the stream behind this segment with EOI (the markers of a following scan overwrite it)
__global__ __launch_bounds__(1024) void jpeg_scan_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n_host,
                                                          Scal* __restrict__ sc, int what, uint8_t* __restrict__ segment, unsigned long long room,
                                                          uint32_t* __restrict__ clear) {
  __shared__ uint64_t wtot[16];
  __shared__ uint64_t carry_s;
  const uint32_t n = what == 0 ? n_host : sc->nchunks;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint64_t v = i < n ? in[i] : 0;
    uint64_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t up = __shfl_up(incl, o, 64);
      if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t before = carry_s;
    for (int k = 0; k < (int)(threadIdx.x >> 6); k++) before += wtot[k];
    if (i < n) {
      out[i] = before + incl - v;
      if (what == 0) clear[(before + incl - v) >> 5] = 0u;   // the dword two workgroups of the WRITE pass may share
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[n] = carry_s;
    if (what == 0) {
      clear[carry_s >> 5] = 0u;   // the dword of the padding bits
      sc->total_bits = carry_s;
      sc->overflow = 0u;
      sc->nbytes = (unsigned int)((carry_s + 7) >> 3);
      sc->nchunks = (sc->nbytes + 4095u) >> 12;
    } else {
      sc->total_ff = (unsigned int)carry_s;
      const unsigned long long seg = (unsigned long long)sc->nbytes + carry_s;
      sc->seg_len = seg;
      if (seg + 2 <= room) { segment[seg] = 0xff; segment[seg + 1] = 0xd9; }
      else sc->overflow = 1u;
    }
  }
}

// byte k of the unstuffed scan: big-endian inside its dword
__device__ __forceinline__ uint32_t be_byte(uint32_t word, int k) { return (word >> (24 - 8 * k)) & 255u; }

// SCATTER = false: ffcnt[chunk] = number of 0xFF bytes in the 4 KB chunk; true: write the chunk's bytes, stuffed, to `stream`.
template <bool SCATTER>
__global__ __launch_bounds__(256) void jpeg_stuff_kernel(const uint32_t* __restrict__ raw, const Scal* __restrict__ sc, uint32_t* __restrict__ ffcnt,
                                                          const uint64_t* __restrict__ ffoff, uint8_t* __restrict__ stream, unsigned long long base,
                                                          unsigned long long cap, Scal* __restrict__ sc_out) {
  __shared__ uint32_t wsum[4];
  const uint32_t nbytes = sc->nbytes, nchunks = sc->nchunks;
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint32_t b0 = chunk * 4096u + threadIdx.x * 16u;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (b0 < nbytes) v = reinterpret_cast<const uint4*>(raw)[b0 >> 4];   // raw is zeroed one dword beyond nbytes; the tail of the uint4 is masked below
    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) cnt += (b0 + k < nbytes && be_byte(wd[k >> 2], k & 3) == 255u) ? 1u : 0u;
    if (!SCATTER) {
      uint32_t t = cnt;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
      __syncthreads();
      if (threadIdx.x == 0) ffcnt[chunk] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      __syncthreads();
    } else {
      uint32_t incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += up;
      }
      if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
      __syncthreads();
      uint32_t before = 0;
      for (int k = 0; k < (int)(threadIdx.x >> 6); k++) before += wsum[k];
      __syncthreads();
      unsigned long long pos = base + b0 + ffoff[chunk] + before + incl - cnt;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if (b0 + k < nbytes) {
          const uint32_t by = be_byte(wd[k >> 2], k & 3);
          if (pos + 2 <= cap) { stream[pos] = (uint8_t)by; if (by == 255u) stream[pos + 1] = 0; }
          else sc_out->overflow = 1u;
          pos += by == 255u ? 2 : 1;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ host: tables, markers, orchestration
struct HuffTable {
  uint8_t bits[17];
  uint8_t vals[256];
  int nvals;
  uint32_t packed[256];  // code << 8 | length
};

// optimal code lengths: T.81 K.2 (figures K.1 - K.4); ties towards the larger symbol value
void optimal_table(const uint32_t* counts, HuffTable& t) {
  long long freq[257];
  int codesize[257], others[257];
  for (int i = 0; i < 256; i++) freq[i] = counts[i];
  freq[256] = 1;
  for (int i = 0; i < 257; i++) { codesize[i] = 0; others[i] = -1; }
  for (;;) {
    int c1 = -1, c2 = -1;
    long long v = INT64_MAX;
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
  bits[i]--;
  memset(&t, 0, sizeof t);
  for (int l = 1; l <= 16; l++) t.bits[l] = (uint8_t)bits[l];
  int n = 0;
  for (int l = 1; l <= 63; l++)
    for (int s = 0; s < 256; s++)
      if (codesize[s] == l) t.vals[n++] = (uint8_t)s;
  t.nvals = n;
  uint32_t code = 0;
  int k = 0;
  for (int l = 1; l <= 16; l++) {
    for (int j = 0; j < t.bits[l]; j++, k++) t.packed[t.vals[k]] = (code++ << 8) | (uint32_t)l;
    code <<= 1;
  }
}

struct Bytes {
  std::vector<uint8_t> v;
  void u8(int x) { v.push_back((uint8_t)x); }
  void u16(int x) { u8(x >> 8); u8(x & 255); }
};

void put_dht(Bytes& b, int cls, int id, const HuffTable& t) {
  b.u16(0xffc4);
  b.u16(2 + 1 + 16 + t.nvals);
  b.u8((cls << 4) | id);
  for (int l = 1; l <= 16; l++) b.u8(t.bits[l]);
  for (int i = 0; i < t.nvals; i++) b.u8(t.vals[i]);
}

void scaled_table(const uint8_t* base, int quality, uint8_t* q) {
  quality = quality < 1 ? 1 : quality > 100 ? 100 : quality;
  const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
  for (int i = 0; i < 64; i++) {
    const int v = (base[i] * scale + 50) / 100;
    q[i] = (uint8_t)(v < 1 ? 1 : v > 255 ? 255 : v);
  }
}

bool make_geo(int w, int h, int input_format, int subsampling, Geo& g) {
  memset(&g, 0, sizeof g);
  g.w = w;
  g.h = h;
  g.ncomp = subsampling == 2 ? 1 : 3;
  g.hs0 = subsampling == 1 ? 2 : 1;
  g.nmcux = tdk_div_up(w, 8 * g.hs0);
  g.nmcuy = tdk_div_up(h, 8);
  g.nby_real = g.nmcuy;
  long long off = 0;
  for (int c = 0; c < g.ncomp; c++) {
    const int hs = c == 0 ? g.hs0 : 1;
    g.nbx[c] = g.nmcux * hs;
    const int wc = c == 0 ? w : (w * hs + g.hs0 - 1) / g.hs0;
    g.nbx_real[c] = tdk_div_up(wc, 8);
    off = (off + 63) & ~63ll;   // a plane starts on a group of the transposed store: 64 consecutive blocks of a wave are one 256-byte row
    g.coff[c] = off;
    off += (long long)g.nbx[c] * g.nmcuy;
  }
  g.nblocks = off;
  g.planar = input_format < 2;
  g.bgr = (input_format & 1) == 0;
  return true;
}

// workspace layout (byte offsets, 256-byte aligned)
struct Layout {
  size_t coef, lens, wgsum, wgoff, hist, tabs, scal, ffcnt, ffoff, raw, stream, total;
  size_t raw_cap, stream_cap, nwg, nchunk_cap;
};
constexpr size_t HEADER_CAP = 8192;      // SOI .. SOF + (DHT x 4 + SOS) per scan, four scans: < 4 KB
constexpr size_t MAX_BLOCK_BYTES = 212;  // 27 bits of DC + 63 x 26 bits of AC

Layout make_layout(const Geo& g) {
  Layout L;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = tdk_align_up(o + bytes, 256); return at; };
  L.nwg = (size_t)tdk_div_up64(g.nblocks, 192) + 1;   // chunks of the coding passes hold 192 or 256 scan positions
  L.raw_cap = tdk_align_up((size_t)g.nblocks * MAX_BLOCK_BYTES + 64, 4096);
  L.nchunk_cap = L.raw_cap / 4096 + 1;
  L.stream_cap = HEADER_CAP + L.raw_cap;
  L.coef = take((size_t)tdk_div_up64(g.nblocks, 64) * 64 * 128);
  L.lens = take((size_t)g.nblocks * 2);
  L.wgsum = take(L.nwg * 4);
  L.wgoff = take((L.nwg + 1) * 8);
  L.hist = take(1024 * 4);
  L.tabs = take(1024 * 4);
  L.scal = take(sizeof(Scal));
  L.ffcnt = take(L.nchunk_cap * 4);
  L.ffoff = take((L.nchunk_cap + 1) * 8);
  L.raw = take(L.raw_cap + 64);
  L.stream = take(L.stream_cap);
  L.total = o;
  return L;
}

template <int SUB> int launch_fdct(const uint8_t* img, uint32_t* coef, const Geo& g, const Quant& q, hipStream_t st) {
  using T = Tile<SUB>;
  const dim3 grid((unsigned)tdk_div_up(g.nmcux * 8 * g.hs0, T::TPX), (unsigned)g.nmcuy);
  TDK_LAUNCH("tdk_jpeg(fdct)", (jpeg_fdct_kernel<SUB>), grid, dim3(256), (size_t)T::LDS_BYTES, st, img, coef, g, q);
  return TDK_OK;
}

// byte positions inside a scan are 32-bit on the device: the worst-case stream of the frame has to stay below 4 GB (about 1.2
// gigapixels at 4:2:2; JPEG itself stops at 65 535 x 65 535)
bool geo_supported(const Geo& g) { return (unsigned long long)g.nblocks * MAX_BLOCK_BYTES < 0xf0000000ull; }

}  // namespace

TDK_EXPORT size_t tdk_jpeg_workspace_bytes(int width, int height, int subsampling) {
  if (width <= 0 || height <= 0 || width > 65535 || height > 65535 || subsampling < 0 || subsampling > 2) return 0;
  Geo g;
  make_geo(width, height, 3, subsampling, g);
  if (!geo_supported(g)) return 0;
  return make_layout(g).total;
}

TDK_EXPORT int tdk_jpeg_encode(const void* image, int width, int height, int input_format, int quality, int subsampling, int progressive,
                               void* workspace, size_t* length, tdk_stream_t stream) {
  TDK_REQUIRE(image && workspace && length, "tdk_jpeg_encode: null pointer");
  TDK_REQUIRE(width > 0 && height > 0 && width <= 65535 && height <= 65535, "tdk_jpeg_encode: image %dx%d outside 1..65535", width, height);
  TDK_REQUIRE(input_format >= 0 && input_format <= 3, "Invalid input format");
  TDK_REQUIRE(subsampling >= 0 && subsampling <= 2, "Invalid subsampling");
  TDK_REQUIRE(tdk_aligned(workspace, 256), "tdk_jpeg_encode: workspace must be 256-byte aligned");
  hipStream_t st = tdk_stream(stream);
  Geo g;
  make_geo(width, height, input_format, subsampling, g);
  TDK_REQUIRE(geo_supported(g), "tdk_jpeg_encode: image %dx%d too large (worst-case stream beyond 4 GB)", width, height);
  const Layout L = make_layout(g);
  uint8_t* ws = reinterpret_cast<uint8_t*>(workspace);
  uint32_t* coef = reinterpret_cast<uint32_t*>(ws + L.coef);
  uint16_t* lens = reinterpret_cast<uint16_t*>(ws + L.lens);
  uint32_t* wgsum = reinterpret_cast<uint32_t*>(ws + L.wgsum);
  uint64_t* wgoff = reinterpret_cast<uint64_t*>(ws + L.wgoff);
  uint32_t* hist = reinterpret_cast<uint32_t*>(ws + L.hist);
  uint32_t* tabs = reinterpret_cast<uint32_t*>(ws + L.tabs);
  Scal* scal = reinterpret_cast<Scal*>(ws + L.scal);
  uint32_t* ffcnt = reinterpret_cast<uint32_t*>(ws + L.ffcnt);
  uint64_t* ffoff = reinterpret_cast<uint64_t*>(ws + L.ffoff);
  uint32_t* raw = reinterpret_cast<uint32_t*>(ws + L.raw);
  uint8_t* out = ws + L.stream;

  uint8_t qt[2][64];
  Quant q;
  scaled_table(Q_LUMA, quality, qt[0]);
  scaled_table(Q_CHROMA, quality, qt[1]);
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 64; i++) q.rq[t][i] = (float)(1.0 / ((double)qt[t][i] * AAN[i >> 3] * AAN[i & 7] * 8.0));

  int rc = subsampling == 0 ? launch_fdct<0>(reinterpret_cast<const uint8_t*>(image), coef, g, q, st)
         : subsampling == 1 ? launch_fdct<1>(reinterpret_cast<const uint8_t*>(image), coef, g, q, st)
                            : launch_fdct<2>(reinterpret_cast<const uint8_t*>(image), coef, g, q, st);
  if (rc != TDK_OK) return rc;

  // frame header
  Bytes hd;
  hd.u16(0xffd8);
  hd.u16(0xffe0); hd.u16(16);
  for (const char ch : {'J', 'F', 'I', 'F', '\0'}) hd.u8(ch);
  hd.u16(0x0101); hd.u8(0); hd.u16(1); hd.u16(1); hd.u8(0); hd.u8(0);
  for (int t = 0; t < (g.ncomp == 1 ? 1 : 2); t++) {
    hd.u16(0xffdb); hd.u16(67); hd.u8(t);
    for (int k = 0; k < 64; k++) hd.u8(qt[t][zz_of(k)]);
  }
  hd.u16(progressive ? 0xffc2 : 0xffc0);
  hd.u16(8 + 3 * g.ncomp); hd.u8(8); hd.u16(height); hd.u16(width); hd.u8(g.ncomp);
  for (int c = 0; c < g.ncomp; c++) { hd.u8(c + 1); hd.u8(((c == 0 ? g.hs0 : 1) << 4) | 1); hd.u8(c ? 1 : 0); }

  // scans
  struct ScanPlan { int ns, comp0, ss, se; };
  std::vector<ScanPlan> plan;
  if (!progressive) plan.push_back({g.ncomp, 0, 0, 63});
  else {
    plan.push_back({g.ncomp, 0, 0, 0});
    for (int c = 0; c < g.ncomp; c++) plan.push_back({1, c, 1, 63});
  }
  std::vector<Bytes> keep;  // host sources of the asynchronous copies stay alive until the last synchronisation
  keep.reserve(plan.size() * 2 + 2);
  std::vector<std::vector<uint32_t>> keep_tabs;
  keep_tabs.reserve(plan.size());
  // an early return (a failed launch or copy) must not free those sources under a copy that is still in flight
  struct DrainOnExit {
    hipStream_t s;
    bool armed;
    ~DrainOnExit() { if (armed) (void)hipStreamSynchronize(s); }
  } drain{st, true};
  size_t pos = 0;
  Bytes pending = hd;  // bytes to put in front of the next scan's entropy-coded segment
  const int persistent = tdk_device_cus() * 4;
  Scal hs;
  for (size_t si = 0; si < plan.size(); si++) {
    const ScanPlan& p = plan[si];
    ScanDesc sc;
    sc.ns = p.ns; sc.comp0 = p.comp0; sc.ss = p.ss; sc.se = p.se;
    if (p.ns > 1) sc.nscan = (long long)g.nmcux * g.nmcuy * (g.hs0 + 2);
    else sc.nscan = (long long)g.nbx_real[p.comp0] * g.nby_real;
    sc.chunk = p.ns > 1 ? 64 * (g.hs0 + 2) : 256;
    const unsigned nwg = (unsigned)tdk_div_up64(sc.nscan, sc.chunk);

    TDK_HIP_CALL(hipMemsetAsync(hist, 0, 4096, st), "tdk_jpeg_encode: memset");
    const unsigned nhist = nwg < (unsigned)(persistent / 2) ? nwg : (unsigned)(persistent / 2);
    TDK_LAUNCH("tdk_jpeg(histogram)", (jpeg_code_kernel<HIST>), dim3(nhist), dim3(256), 0, st, coef, g, sc, hist, tabs, lens, wgsum, wgoff, raw);
    uint32_t hh[1024];
    TDK_HIP_CALL(hipMemcpyAsync(hh, hist, sizeof hh, hipMemcpyDeviceToHost, st), "tdk_jpeg_encode: histogram copy");
    if (si > 0) TDK_HIP_CALL(hipMemcpyAsync(&hs, scal, sizeof hs, hipMemcpyDeviceToHost, st), "tdk_jpeg_encode: length copy");
    TDK_HIP_CALL(hipStreamSynchronize(st), "tdk_jpeg_encode: synchronize");
    if (si > 0) {
      TDK_REQUIRE(!hs.overflow, "tdk_jpeg_encode: stream larger than the workspace");
      pos += (size_t)hs.seg_len;
    }

    // tables + the markers in front of this scan
    HuffTable tb[4];
    keep_tabs.emplace_back(1024, 0u);
    std::vector<uint32_t>& packed = keep_tabs.back();
    bool used[2] = {false, false};
    for (int i = 0; i < p.ns; i++) used[(p.comp0 + i) ? 1 : 0] = true;
    for (int t = 0; t < 2; t++) {
      if (!used[t]) continue;
      if (p.ss == 0) { optimal_table(hh + t * 256, tb[t]); put_dht(pending, 0, t, tb[t]); memcpy(&packed[t * 256], tb[t].packed, 1024); }
      if (p.se > 0) { optimal_table(hh + (2 + t) * 256, tb[2 + t]); put_dht(pending, 1, t, tb[2 + t]); memcpy(&packed[(2 + t) * 256], tb[2 + t].packed, 1024); }
    }
    pending.u16(0xffda);
    pending.u16(6 + 2 * p.ns);
    pending.u8(p.ns);
    for (int i = 0; i < p.ns; i++) { const int c = p.comp0 + i; pending.u8(c + 1); pending.u8(c ? 0x11 : 0x00); }
    pending.u8(p.ss);
    pending.u8(p.se);
    pending.u8(0);
    TDK_REQUIRE(pos + pending.v.size() + 2 <= L.stream_cap, "tdk_jpeg_encode: stream larger than the workspace");
    keep.push_back(pending);
    TDK_HIP_CALL(hipMemcpyAsync(out + pos, keep.back().v.data(), keep.back().v.size(), hipMemcpyHostToDevice, st), "tdk_jpeg_encode: header copy");
    pos += pending.v.size();
    pending.v.clear();
    TDK_HIP_CALL(hipMemcpyAsync(tabs, packed.data(), 4096, hipMemcpyHostToDevice, st), "tdk_jpeg_encode: table copy");

    TDK_LAUNCH("tdk_jpeg(lengths)", (jpeg_code_kernel<LEN>), dim3(nwg), dim3(256), 0, st, coef, g, sc, hist, tabs, lens, wgsum, wgoff, raw);
    TDK_LAUNCH("tdk_jpeg(scan)", jpeg_scan_kernel, dim3(1), dim3(1024), 0, st, wgsum, wgoff, (uint32_t)nwg, scal, 0, out, 0ull, raw);
    TDK_LAUNCH("tdk_jpeg(write)", (jpeg_code_kernel<WRITE>), dim3(nwg), dim3(256), 0, st, coef, g, sc, hist, tabs, lens, wgsum, wgoff, raw);
    TDK_LAUNCH("tdk_jpeg(count ff)", (jpeg_stuff_kernel<false>), dim3((unsigned)persistent), dim3(256), 0, st, raw, scal, ffcnt, ffoff, out,
               (unsigned long long)pos, (unsigned long long)(L.stream_cap - 2), scal);
    TDK_LAUNCH("tdk_jpeg(scan)", jpeg_scan_kernel, dim3(1), dim3(1024), 0, st, ffcnt, ffoff, 0u, scal, 1, out + pos, (unsigned long long)(L.stream_cap - pos), raw);
    TDK_LAUNCH("tdk_jpeg(stuff)", (jpeg_stuff_kernel<true>), dim3((unsigned)persistent), dim3(256), 0, st, raw, scal, ffcnt, ffoff, out,
               (unsigned long long)pos, (unsigned long long)(L.stream_cap - 2), scal);
  }
  TDK_HIP_CALL(hipMemcpyAsync(&hs, scal, sizeof hs, hipMemcpyDeviceToHost, st), "tdk_jpeg_encode: length copy");
  TDK_HIP_CALL(hipStreamSynchronize(st), "tdk_jpeg_encode: synchronize");
  TDK_REQUIRE(!hs.overflow, "tdk_jpeg_encode: stream larger than the workspace");
  pos += (size_t)hs.seg_len;
  *length = pos + 2;  // EOI is already there (jpeg_scan_kernel)
  drain.armed = false;  // the stream was synchronised two lines up
  return TDK_OK;
}

TDK_EXPORT int tdk_jpeg_retrieve(const void* workspace, int width, int height, int subsampling, uint8_t* out_host, size_t length, tdk_stream_t stream) {
  TDK_REQUIRE(workspace && out_host, "tdk_jpeg_retrieve: null pointer");
  TDK_REQUIRE(width > 0 && height > 0 && width <= 65535 && height <= 65535 && subsampling >= 0 && subsampling <= 2, "tdk_jpeg_retrieve: bad geometry");
  Geo g;
  make_geo(width, height, 3, subsampling, g);
  const Layout L = make_layout(g);
  TDK_REQUIRE(length <= L.stream_cap, "tdk_jpeg_retrieve: length %zu beyond the stream buffer", length);
  hipStream_t st = tdk_stream(stream);
  TDK_HIP_CALL(hipMemcpyAsync(out_host, reinterpret_cast<const uint8_t*>(workspace) + L.stream, length, hipMemcpyDeviceToHost, st), "tdk_jpeg_retrieve: copy");
  TDK_HIP_CALL(hipStreamSynchronize(st), "tdk_jpeg_retrieve: synchronize");
  return TDK_OK;
}

// test hook: the quantised coefficient planes of the last tdk_jpeg_encode on this workspace (zig-zag int16, components back to back)
TDK_EXPORT int tdk_jpeg_coefficients(const void* workspace, int width, int height, int subsampling, int16_t* out_host, tdk_stream_t stream) {
  TDK_REQUIRE(workspace && out_host, "tdk_jpeg_coefficients: null pointer");
  TDK_REQUIRE(width > 0 && height > 0 && width <= 65535 && height <= 65535 && subsampling >= 0 && subsampling <= 2, "tdk_jpeg_coefficients: bad geometry");
  Geo g;
  make_geo(width, height, 3, subsampling, g);
  const Layout L = make_layout(g);
  hipStream_t st = tdk_stream(stream);
  const size_t nwords = (size_t)tdk_div_up64(g.nblocks, 64) * 64 * 32;
  std::vector<uint32_t> tmp(nwords);
  TDK_HIP_CALL(hipMemcpyAsync(tmp.data(), reinterpret_cast<const uint8_t*>(workspace) + L.coef, nwords * 4, hipMemcpyDeviceToHost, st), "tdk_jpeg_coefficients: copy");
  TDK_HIP_CALL(hipStreamSynchronize(st), "tdk_jpeg_coefficients: synchronize");
  long long o = 0;
  for (int c = 0; c < g.ncomp; c++)
    for (long long b = 0; b < (long long)g.nbx[c] * g.nmcuy; b++, o++)
      for (int j = 0; j < 32; j++) {
        const uint32_t v = tmp[coef_word(g.coff[c] + b, j)];
        out_host[o * 64 + 2 * j] = (int16_t)(v & 0xffffu);
        out_host[o * 64 + 2 * j + 1] = (int16_t)(v >> 16);
      }
  return TDK_OK;
}
