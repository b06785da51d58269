// Micro-benchmark: SIMD cycles per VALU instruction by encoding, independent instruction streams (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define ADD_E32(i) "v_add_f32_e32 %" #i ", %16, %" #i "\n\t"
#define FMAC_LIT(i) "v_fmac_f32_e32 %" #i ", 0x3f54db31, %16\n\t"
#define FMAMK(i) "v_fmamk_f32 %" #i ", %16, 0x3f54db31, %" #i "\n\t"
#define FMA_VOP3(i) "v_fma_f32 %" #i ", %16, %17, %" #i "\n\t"
#define FMA_SGPR(i) "v_fma_f32 %" #i ", %16, %18, %" #i "\n\t"
#define MUL_LIT(i) "v_mul_f32_e32 %" #i ", 0x3f7ff000, %" #i "\n\t"
#define FMAC_E32(i) "v_fmac_f32_e32 %" #i ", %16, %17\n\t"
#define EXP(i) "v_exp_f32_e32 %" #i ", %" #i "\n\t"
#define RCP(i) "v_rcp_f32_e32 %" #i ", %" #i "\n\t"
#define CND64(i) "v_cndmask_b32_e64 %" #i ", %16, %" #i ", %19\n\t"
#define MAX3(i) "v_max3_f32 %" #i ", %16, %17, %" #i "\n\t"
#define MUL_SGPR_E32(i) "v_mul_f32_e32 %" #i ", %18, %" #i "\n\t"
#define FMAC_SGPR_E32(i) "v_fmac_f32_e32 %" #i ", %18, %16\n\t"
#define ADD_SGPR_E32(i) "v_add_f32_e32 %" #i ", %18, %" #i "\n\t"
#define CVT(i) "v_cvt_f16_f32_e32 %" #i ", %" #i "\n\t"
#define OPERANDS : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(x), "v"(y), "s"(sc), "s"(mask)
template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  float r[16];
  for (int i = 0; i < 16; i++) r[i] = p[threadIdx.x + i];
  const float x = p[threadIdx.x + 20] * 1e-9f, y = 0.999f, sc = 0.999f;
  const unsigned long long mask = 0x5555555555555555ull;
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) asm volatile(R16(ADD_E32) R16(ADD_E32) OPERANDS);
    else if constexpr (MODE == 1) asm volatile(R16(FMAC_LIT) R16(FMAC_LIT) OPERANDS);
    else if constexpr (MODE == 2) asm volatile(R16(FMAMK) R16(FMAMK) OPERANDS);
    else if constexpr (MODE == 3) asm volatile(R16(FMA_VOP3) R16(FMA_VOP3) OPERANDS);
    else if constexpr (MODE == 4) asm volatile(R16(FMA_SGPR) R16(FMA_SGPR) OPERANDS);
    else if constexpr (MODE == 5) asm volatile(R16(MUL_LIT) R16(MUL_LIT) OPERANDS);
    else if constexpr (MODE == 6) asm volatile(R16(FMAC_E32) R16(FMAC_E32) OPERANDS);
    else if constexpr (MODE == 7) asm volatile(R16(EXP) R16(EXP) OPERANDS);
    else if constexpr (MODE == 8) asm volatile(R16(RCP) R16(RCP) OPERANDS);
    else if constexpr (MODE == 9) asm volatile(R16(CND64) R16(CND64) OPERANDS);
    else if constexpr (MODE == 10) asm volatile(R16(MAX3) R16(MAX3) OPERANDS);
    else if constexpr (MODE == 11) asm volatile(R16(CVT) R16(CVT) OPERANDS);
    else if constexpr (MODE == 12) asm volatile(R16(MUL_SGPR_E32) R16(MUL_SGPR_E32) OPERANDS);
    else if constexpr (MODE == 13) asm volatile(R16(FMAC_SGPR_E32) R16(FMAC_SGPR_E32) OPERANDS);
    else if constexpr (MODE == 14) asm volatile(R16(ADD_SGPR_E32) R16(ADD_SGPR_E32) OPERANDS);
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += r[i];
  p[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d) {
  printf("%-34s", name);
  for (int bpc : {1, 2, 4, 8}) {
    const int iters = 40000, blocks = 256 * bpc;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    bench<MODE><<<blocks, 256>>>(d, iters / 4);
    (void)hipEventRecord(a);
    bench<MODE><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("  %dw: %5.2f", bpc, ms * 1e-3 * 2.4e9 / iters / 32.0 / bpc);
  }
  printf("   (SIMD cycles per instruction @2.4 GHz)\n");
}
int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  run<0>("v_add_f32_e32 (4 B)", d);
  run<6>("v_fmac_f32_e32 (4 B)", d);
  run<1>("v_fmac_f32_e32 literal (8 B)", d);
  run<2>("v_fmamk_f32 literal (8 B)", d);
  run<5>("v_mul_f32_e32 literal (8 B)", d);
  run<3>("v_fma_f32 VOP3 vgprs (8 B)", d);
  run<4>("v_fma_f32 VOP3 sgpr (8 B)", d);
  run<12>("v_mul_f32_e32 sgpr src0 (4 B)", d);
  run<13>("v_fmac_f32_e32 sgpr src0 (4 B)", d);
  run<14>("v_add_f32_e32 sgpr src0 (4 B)", d);
  run<10>("v_max3_f32 VOP3 (8 B)", d);
  run<9>("v_cndmask_b32_e64 (8 B)", d);
  run<11>("v_cvt_f16_f32 (4 B)", d);
  run<7>("v_exp_f32", d);
  run<8>("v_rcp_f32", d);
  return 0;
}
