// Micro-benchmark: does a LONG straight-line VALU body issue as fast as a short loop?  (gfx950)
//   hipcc -O3 --offload-arch=gfx950 -o tests/hip_unit/build/ifetch_bench tests/hip_unit/ifetch_bench.hip
// Bodies of 32 / 256 / 2048 independent VALU instructions (16 register streams) in 4-byte (v_add_f32_e32) and 8-byte
// (v_fmac_f32_e32 with a literal) encodings, at 1 / 2 / 4 waves per SIMD: SIMD cycles per instruction and the implied
// instruction bytes per cycle per CU.  A body that no longer fits the wave's instruction buffer has to stream from the
// instruction cache (shared by two CUs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define ADD_E32(i) "v_add_f32_e32 %" #i ", %16, %" #i "\n\t"
#define FMAC_LIT(i) "v_fmac_f32_e32 %" #i ", 0x3f54db31, %16\n\t"
#define MIX(i) "v_add_f32_e32 %" #i ", %16, %" #i "\n\t" "v_fmac_f32_e32 %" #i ", 0x3f54db31, %16\n\t"
#define X2(a) a a
#define X8(a) X2(X2(X2(a)))
#define X64(a) X8(X8(a))
#define OPERANDS : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(x)

// MODE: encoding 0 = add (4 B), 1 = fmac literal (8 B), 2 = alternating (6 B avg);  LEN: 0 = 32, 1 = 256, 2 = 2048 instructions per loop body
template <int MODE, int LEN> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  float r[16];
  for (int i = 0; i < 16; i++) r[i] = p[threadIdx.x + i];
  const float x = p[threadIdx.x + 20] * 1e-9f;
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) {
      if constexpr (LEN == 0) asm volatile(X2(R16(ADD_E32)) OPERANDS);
      else if constexpr (LEN == 1) asm volatile(X8(X2(R16(ADD_E32))) OPERANDS);
      else asm volatile(X64(X2(R16(ADD_E32))) OPERANDS);
    } else if constexpr (MODE == 1) {
      if constexpr (LEN == 0) asm volatile(X2(R16(FMAC_LIT)) OPERANDS);
      else if constexpr (LEN == 1) asm volatile(X8(X2(R16(FMAC_LIT))) OPERANDS);
      else asm volatile(X64(X2(R16(FMAC_LIT))) OPERANDS);
    } else {
      if constexpr (LEN == 0) asm volatile(R16(MIX) OPERANDS);
      else if constexpr (LEN == 1) asm volatile(X8(R16(MIX)) OPERANDS);
      else asm volatile(X64(R16(MIX)) OPERANDS);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += r[i];
  p[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int LEN> void run(const char* name, float* d) {
  const int n = LEN == 0 ? 32 : (LEN == 1 ? 256 : 2048);
  const double bytes = MODE == 0 ? 4.0 : (MODE == 1 ? 8.0 : 6.0);
  printf("%-26s body %4d instr:", name, n);
  for (int bpc : {1, 2, 4}) {
    const int iters = 2000000 / n, blocks = 256 * bpc;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    bench<MODE, LEN><<<blocks, 256>>>(d, iters / 4);
    (void)hipEventRecord(a);
    bench<MODE, LEN><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * n) / bpc;  // SIMD cycles per instruction
    printf("  %dw: %5.2f cyc/instr (%4.1f B/clk/CU)", bpc, cyc, 4.0 * bytes / cyc);
  }
  printf("\n");
}

int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  run<0, 0>("v_add_f32_e32 (4 B)", d); run<0, 1>("v_add_f32_e32 (4 B)", d); run<0, 2>("v_add_f32_e32 (4 B)", d);
  run<1, 0>("v_fmac literal (8 B)", d); run<1, 1>("v_fmac literal (8 B)", d); run<1, 2>("v_fmac literal (8 B)", d);
  run<2, 0>("add / fmac-literal mix", d); run<2, 1>("add / fmac-literal mix", d); run<2, 2>("add / fmac-literal mix", d);
  return 0;
}
