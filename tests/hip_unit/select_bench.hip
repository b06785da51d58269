// Micro-benchmark: what a per-lane select costs on gfx950, by where its lane mask lives.  Independent instruction
// streams (16 destination registers), SIMD cycles per instruction (or per bracketed group) at 1 / 2 / 4 / 8 waves per SIMD, 2.4 GHz assumed.
//   hipcc -O3 --offload-arch=gfx950 -o tests/hip_unit/build/select_bench tests/hip_unit/select_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define ADD(i) "v_add_f32_e32 %" #i ", %16, %" #i "\n\t"
#define CND_VCC_E32(i) "v_cndmask_b32_e32 %" #i ", %16, %" #i ", vcc\n\t"
#define CND_VCC_E64(i) "v_cndmask_b32_e64 %" #i ", %16, %" #i ", vcc\n\t"
#define CND_SGPR(i) "v_cndmask_b32_e64 %" #i ", %16, %" #i ", %19\n\t"
#define CND_SGPR_IMM(i) "v_cndmask_b32_e64 %" #i ", 0, %" #i ", %19\n\t"
#define AND_V(i) "v_and_b32_e32 %" #i ", %17, %" #i "\n\t"
#define CMP_VCC(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\n\t"
#define CMP_SGPR(i) "v_cmp_lt_f32_e64 %20, %16, %" #i "\n\t"
#define CMP_CND_VCC(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\n\tv_cndmask_b32_e32 %" #i ", %17, %" #i ", vcc\n\t"
#define CMP_CND_SGPR(i) "v_cmp_lt_f32_e64 %20, %16, %" #i "\n\tv_cndmask_b32_e64 %" #i ", %17, %" #i ", %20\n\t"
#define CMP_CND_SGPR2(i) "v_cmp_lt_f32_e64 %21, %16, %" #i "\n\tv_cndmask_b32_e64 %" #i ", %17, %" #i ", %21\n\t"
#define ALT16(A, B) A(0) B(1) A(2) B(3) A(4) B(5) A(6) B(7) A(8) B(9) A(10) B(11) A(12) B(13) A(14) B(15)
// two compares first, then their two selects: the form a scheduler would produce
#define PAIR2(i, j) "v_cmp_lt_f32_e64 %20, %16, %" #i "\n\tv_cmp_lt_f32_e64 %21, %16, %" #j "\n\tv_cndmask_b32_e64 %" #i ", %17, %" #i ", %20\n\tv_cndmask_b32_e64 %" #j ", %17, %" #j ", %21\n\t"
#define PAIRS16 PAIR2(0, 1) PAIR2(2, 3) PAIR2(4, 5) PAIR2(6, 7) PAIR2(8, 9) PAIR2(10, 11) PAIR2(12, 13) PAIR2(14, 15)
// the forms hipcc emits: two wait states between a VALU write of VCC / an SGPR pair and the v_cndmask that reads it as its lane mask
#define NOP_VCC(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\n\ts_nop 1\n\tv_cndmask_b32_e32 %" #i ", %17, %" #i ", vcc\n\t"
#define NOP_VCC64(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\n\ts_nop 1\n\tv_cndmask_b32_e64 %" #i ", %17, %" #i ", vcc\n\t"
#define NOP_SGPR(i) "v_cmp_lt_f32_e64 %20, %16, %" #i "\n\ts_nop 1\n\tv_cndmask_b32_e64 %" #i ", %17, %" #i ", %20\n\t"
#define FILL_VCC(i, j) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\n\tv_add_f32_e32 %" #j ", %16, %" #j "\n\tv_mul_f32_e32 %" #j ", %17, %" #j "\n\tv_cndmask_b32_e32 %" #i ", %17, %" #i ", vcc\n\t"
#define FILL_SGPR(i, j) "v_cmp_lt_f32_e64 %20, %16, %" #i "\n\tv_add_f32_e32 %" #j ", %16, %" #j "\n\tv_mul_f32_e32 %" #j ", %17, %" #j "\n\tv_cndmask_b32_e64 %" #i ", %17, %" #i ", %20\n\t"
#define TWO_VCC(i, j) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\n\ts_nop 1\n\tv_cndmask_b32_e32 %" #i ", %17, %" #i ", vcc\n\tv_cndmask_b32_e32 %" #j ", %17, %" #j ", vcc\n\t"
#define TWO_SGPR(i, j) "v_cmp_lt_f32_e64 %20, %16, %" #i "\n\ts_nop 1\n\tv_cndmask_b32_e64 %" #i ", %17, %" #i ", %20\n\tv_cndmask_b32_e64 %" #j ", %17, %" #j ", %20\n\t"
#define P8(OP) OP(0, 1) OP(2, 3) OP(4, 5) OP(6, 7) OP(8, 9) OP(10, 11) OP(12, 13) OP(14, 15)
#define CMP_CND_SGPR_ABS(i) "v_cmp_lt_f32_e64 %20, |%16|, |%" #i "|\n\tv_cndmask_b32_e64 %" #i ", %17, %" #i ", %20\n\t"
#define MAXMIN(i) "v_max_f32_e32 %" #i ", %16, %" #i "\n\t"
#define MED3(i) "v_med3_f32 %" #i ", %16, %17, %" #i "\n\t"
#define OPERANDS : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(x), "v"(y), "s"(sc), "s"(mask), "s"(tmp), "s"(tmp2) : "vcc", "memory"
template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  float r[16];
  for (int i = 0; i < 16; i++) r[i] = p[threadIdx.x + i];
  const float x = p[threadIdx.x + 20] * 1e-9f + 0.5f, y = 0.999f, sc = 0.999f;
  const unsigned long long mask = 0x5555555555555555ull;
  unsigned long long tmp = 0x3333333333333333ull, tmp2 = 0x0f0f0f0f0f0f0f0full;  // an SGPR pair the compare forms overwrite (declared as input: nothing reads it afterwards)
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) asm volatile(R16(ADD) R16(ADD) OPERANDS);
    else if constexpr (MODE == 1) asm volatile("s_mov_b64 vcc, %19\n\t" R16(CND_VCC_E32) R16(CND_VCC_E32) OPERANDS);
    else if constexpr (MODE == 2) asm volatile("s_mov_b64 vcc, %19\n\t" R16(CND_VCC_E64) R16(CND_VCC_E64) OPERANDS);
    else if constexpr (MODE == 3) asm volatile(R16(CND_SGPR) R16(CND_SGPR) OPERANDS);
    else if constexpr (MODE == 4) asm volatile(R16(CND_SGPR_IMM) R16(CND_SGPR_IMM) OPERANDS);
    else if constexpr (MODE == 5) asm volatile(R16(AND_V) R16(AND_V) OPERANDS);
    else if constexpr (MODE == 6) asm volatile(R16(CMP_VCC) R16(CMP_VCC) OPERANDS);
    else if constexpr (MODE == 7) asm volatile(R16(CMP_SGPR) R16(CMP_SGPR) OPERANDS);
    else if constexpr (MODE == 8) asm volatile(R16(CMP_CND_VCC) OPERANDS);
    else if constexpr (MODE == 9) asm volatile(R16(CMP_CND_SGPR) OPERANDS);
    else if constexpr (MODE == 10) asm volatile(R16(CMP_CND_SGPR_ABS) OPERANDS);
    else if constexpr (MODE == 11) asm volatile(R16(MAXMIN) R16(MAXMIN) OPERANDS);
    else if constexpr (MODE == 13) asm volatile(ALT16(CMP_CND_SGPR, CMP_CND_SGPR2) OPERANDS);
    else if constexpr (MODE == 14) asm volatile(PAIRS16 OPERANDS);
    else if constexpr (MODE == 12) asm volatile(R16(MED3) R16(MED3) OPERANDS);
    else if constexpr (MODE == 15) asm volatile(R16(NOP_VCC) OPERANDS);
    else if constexpr (MODE == 16) asm volatile(R16(NOP_SGPR) OPERANDS);
    else if constexpr (MODE == 17) asm volatile(P8(FILL_VCC) P8(FILL_VCC) OPERANDS);
    else if constexpr (MODE == 18) asm volatile(P8(FILL_SGPR) P8(FILL_SGPR) OPERANDS);
    else if constexpr (MODE == 19) asm volatile(P8(TWO_VCC) P8(TWO_VCC) OPERANDS);
    else if constexpr (MODE == 20) asm volatile(P8(TWO_SGPR) P8(TWO_SGPR) OPERANDS);
    else if constexpr (MODE == 21) asm volatile(R16(NOP_VCC64) OPERANDS);
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += r[i];
  p[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, float per = 32.0f) {
  printf("%-58s", name);
  for (int bpc : {1, 2, 4, 8}) {
    const int iters = 20000, blocks = 256 * bpc;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    bench<MODE><<<blocks, 256>>>(d, iters / 4);
    (void)hipEventRecord(a);
    bench<MODE><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("  %dw: %6.2f", bpc, ms * 1e-3 * 2.4e9 / iters / per / bpc);
  }
  printf("\n");
}
int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  run<0>("v_add_f32_e32 (reference)", d);
  run<1>("v_cndmask_b32_e32 ..., vcc", d);
  run<2>("v_cndmask_b32_e64 ..., vcc (VOP3 encoding)", d);
  run<3>("v_cndmask_b32_e64 ..., s[n:n+1]", d);
  run<4>("v_cndmask_b32_e64 0, v, s[n:n+1]", d);
  run<5>("v_and_b32 mask-in-VGPR", d);
  run<6>("v_cmp_lt_f32_e32 vcc", d);
  run<7>("v_cmp_lt_f32_e64 s[n:n+1]", d);
  run<8>("v_cmp_e32 vcc + v_cndmask_e32 vcc (pair)", d);
  run<9>("v_cmp_e64 s[n:n+1] + v_cndmask_e64 s[n:n+1] (pair)", d);
  run<13>("same, two SGPR pairs alternating", d);
  run<14>("same, two compares then two selects", d);
  run<10>("v_cmp_e64 |a|,|b| s[n:n+1] + v_cndmask_e64 (pair)", d);
  run<11>("v_max_f32_e32", d);
  run<12>("v_med3_f32", d);
  printf("-- per GROUP (cycles for everything in the brackets), with the wait states hipcc inserts:\n");
  run<15>("[v_cmp_e32 vcc; s_nop 1; v_cndmask_e32 vcc]", d, 16.0f);
  run<21>("[v_cmp_e32 vcc; s_nop 1; v_cndmask_e64 vcc]", d, 16.0f);
  run<16>("[v_cmp_e64 s; s_nop 1; v_cndmask_e64 s]", d, 16.0f);
  run<17>("[v_cmp_e32 vcc; v_add; v_mul; v_cndmask_e32 vcc]", d, 16.0f);
  run<18>("[v_cmp_e64 s; v_add; v_mul; v_cndmask_e64 s]", d, 16.0f);
  run<19>("[v_cmp_e32 vcc; s_nop 1; 2 x v_cndmask_e32 vcc]", d, 16.0f);
  run<20>("[v_cmp_e64 s; s_nop 1; 2 x v_cndmask_e64 s]", d, 16.0f);
  return 0;
}
