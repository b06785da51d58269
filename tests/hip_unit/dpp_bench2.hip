#include <hip/hip_runtime.h>
#include <stdio.h>
#define CND "v_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %2, %3, %2, vcc\n\tv_cndmask_b32 %4, %5, %4, vcc\n\tv_cndmask_b32 %6, %7, %6, vcc\n\t" \
            "v_cndmask_b32 %1, %0, %1, vcc\n\tv_cndmask_b32 %3, %2, %3, vcc\n\tv_cndmask_b32 %5, %4, %5, vcc\n\tv_cndmask_b32 %7, %6, %7, vcc\n\t"
#define CNDD "v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
             "v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
             "v_cndmask_b32_dpp %1, %0, %1, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_cndmask_b32_dpp %3, %2, %3, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
             "v_cndmask_b32_dpp %5, %4, %5, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_cndmask_b32_dpp %7, %6, %7, vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CND64 "v_cndmask_b32_e64 %0, %1, %0, %8\n\tv_cndmask_b32_e64 %2, %3, %2, %8\n\tv_cndmask_b32_e64 %4, %5, %4, %8\n\tv_cndmask_b32_e64 %6, %7, %6, %8\n\t" \
              "v_cndmask_b32_e64 %1, %0, %1, %8\n\tv_cndmask_b32_e64 %3, %2, %3, %8\n\tv_cndmask_b32_e64 %5, %4, %5, %8\n\tv_cndmask_b32_e64 %7, %6, %7, %8\n\t"
#define OPS : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(m) : "vcc"
template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  float a = p[threadIdx.x], b = a * 2, c = a * 3, d = a * 4, e = a * 5, f = a * 6, g = a * 7, h = a * 8;
  const unsigned long long m = 0x5555555555555555ull;
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) asm volatile("s_mov_b64 vcc, %8\n\t" CND OPS);                       // 1 s_mov per 8
    else if constexpr (MODE == 1) asm volatile("s_mov_b64 vcc, %8\n\t" CND CND CND CND OPS);      // 1 s_mov per 32
    else if constexpr (MODE == 2) asm volatile(CND64 CND64 CND64 CND64 OPS);                        // sgpr-pair mask, no vcc
    else if constexpr (MODE == 3) asm volatile("s_mov_b64 vcc, %8\n\t" CNDD CNDD CNDD CNDD OPS);  // dpp, 1 s_mov per 32
    else if constexpr (MODE == 4) asm volatile("s_mov_b64 vcc, %8\n\ts_nop 4\n\t" CND OPS);       // s_mov, nop, 8
    else if constexpr (MODE == 5) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\t" CND OPS);           // VALU writes vcc
  }
  p[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <int MODE> void run(const char* name, float* d, int bpc, int n) {
  const int iters = 20000, blocks = 256 * bpc;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  bench<MODE><<<blocks, 256>>>(d, 10);
  (void)hipEventRecord(a);
  bench<MODE><<<blocks, 256>>>(d, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  printf("%-44s %d waves/SIMD: %6.2f SIMD cycles per cndmask\n", name, bpc, ms * 1e-3 * 2.4e9 / iters / n / bpc);
}
int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  for (int w : {1, 4}) {
    run<0>("s_mov vcc + 8 cndmask", d, w, 8);
    run<1>("s_mov vcc + 32 cndmask", d, w, 32);
    run<2>("32 cndmask_e64 sgpr mask", d, w, 32);
    run<3>("s_mov vcc + 32 cndmask_dpp", d, w, 32);
    run<4>("s_mov vcc + s_nop 4 + 8 cndmask", d, w, 8);
    run<5>("v_cmp vcc + 8 cndmask", d, w, 8);
  }
  return 0;
}
