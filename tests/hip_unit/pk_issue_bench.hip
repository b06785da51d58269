// Micro-benchmark: issue cost of packed fp32 VALU instructions on gfx950 (independent streams, 8 register pairs), against
// the plain forms, at 1 / 2 / 4 waves per SIMD.  Prints SIMD cycles per INSTRUCTION and per fp32 operation-lane (a packed
// instruction does two per lane).
//   hipcc -O3 --offload-arch=gfx950 -o tests/hip_unit/build/pk_issue_bench tests/hip_unit/pk_issue_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define X4(a) a a a a
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define PKFMA_SW(i) "v_pk_fma_f32 %" #i ", %8, %" #i ", %9 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]\n\t"
#define PKFMA_S(i) "v_pk_fma_f32 %" #i ", %10, %9, %" #i "\n\t"
#define PKADD(i) "v_pk_add_f32 %" #i ", %8, %" #i "\n\t"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %8, %" #i "\n\t"
#define FMA2(i) "v_fmac_f32_e32 %" #i ", %11, %12\n\t"
#define OPERANDS : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x), "v"(y), "s"(sc), "v"(xs), "v"(ys)

template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  v2f r[8];
  for (int i = 0; i < 8; i++) r[i] = v2f{p[threadIdx.x + i], p[threadIdx.x + i + 8]};
  const v2f x = {p[threadIdx.x + 20] * 1e-9f, p[threadIdx.x + 21] * 1e-9f}, y = {0.999f, 0.998f};
  const float xs = x.x, ys = y.x;
  const v2f sc = {0.999f, 0.997f};
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) asm volatile(X4(X4(R8(PKFMA))) OPERANDS);
    else if constexpr (MODE == 1) asm volatile(X4(X4(R8(PKFMA_SW))) OPERANDS);
    else if constexpr (MODE == 2) asm volatile(X4(X4(R8(PKFMA_S))) OPERANDS);
    else if constexpr (MODE == 3) asm volatile(X4(X4(R8(PKADD))) OPERANDS);
    else if constexpr (MODE == 4) asm volatile(X4(X4(R8(PKMUL))) OPERANDS);
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += r[i].x + r[i].y;
  p[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* d) {
  printf("%-40s", name);
  for (int bpc : {1, 2, 4}) {
    const int n = 128, iters = 20000, blocks = 256 * bpc;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    bench<MODE><<<blocks, 256>>>(d, iters / 4);
    (void)hipEventRecord(a);
    bench<MODE><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * n) / bpc;
    printf("  %dw: %5.2f cyc/instr (%4.2f per fp32 op-wave)", bpc, cyc, cyc / 2);
  }
  printf("\n");
}

int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  run<0>("v_pk_fma_f32 vgprs", d);
  run<1>("v_pk_fma_f32 op_sel swap + neg_hi", d);
  run<2>("v_pk_fma_f32 sgpr-pair src0", d);
  run<3>("v_pk_add_f32", d);
  run<4>("v_pk_mul_f32", d);
  return 0;
}
