// Micro-benchmark: issue cost of cross-lane VALU forms on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  float a = p[threadIdx.x], b = a * 2, c = a * 3, d = a * 4, e = a * 5, f = a * 6, g = a * 7, h = a * 8;
  const unsigned long long m = 0x5555555555555555ull;
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) {  // 8 plain v_mov
      asm volatile(REP8("v_mov_b32 %0, %1\n\t") "v_mov_b32 %1, %0" : "+v"(a), "+v"(b));
    } else if constexpr (MODE == 1) {  // 8 v_mov_dpp quad_perm, independent registers
      asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_mov_b32_dpp %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_mov_b32_dpp %5, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_mov_b32_dpp %7, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (MODE == 2) {  // 8 v_cndmask_dpp, vcc set once per 8
      asm volatile("s_mov_b64 vcc, %8\n\t"
                   "v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_cndmask_b32_dpp %1, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_cndmask_b32_dpp %3, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_cndmask_b32_dpp %5, %4, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_cndmask_b32_dpp %7, %6, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(m) : "vcc");
    } else if constexpr (MODE == 3) {  // 8 plain v_cndmask (no dpp), vcc set once per 8
      asm volatile("s_mov_b64 vcc, %8\n\t"
                   "v_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %2, %3, %2, vcc\n\tv_cndmask_b32 %4, %5, %4, vcc\n\tv_cndmask_b32 %6, %7, %6, vcc\n\t"
                   "v_cndmask_b32 %1, %0, %1, vcc\n\tv_cndmask_b32 %3, %2, %3, vcc\n\tv_cndmask_b32 %5, %4, %5, vcc\n\tv_cndmask_b32 %7, %6, %7, vcc\n\t"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(m) : "vcc");
    } else if constexpr (MODE == 4) {  // 8 v_add_f32_dpp
      asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %2, %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %4, %5, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %6, %7, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %1, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %3, %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %5, %4, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %7, %6, %7 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (MODE == 5) {  // 8 v_permlane16_swap
      asm volatile(REP8("v_permlane16_swap_b32 %0, %1\n\t") "" : "+v"(a), "+v"(b));
    } else if constexpr (MODE == 6) {  // 8 v_permlane32_swap
      asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n\t") "" : "+v"(a), "+v"(b));
    } else if constexpr (MODE == 7) {  // 8 ds_bpermute
      int idx = (threadIdx.x ^ 1) * 4;
      asm volatile("ds_bpermute_b32 %0, %8, %0\n\tds_bpermute_b32 %1, %8, %1\n\tds_bpermute_b32 %2, %8, %2\n\tds_bpermute_b32 %3, %8, %3\n\t"
                   "ds_bpermute_b32 %4, %8, %4\n\tds_bpermute_b32 %5, %8, %5\n\tds_bpermute_b32 %6, %8, %6\n\tds_bpermute_b32 %7, %8, %7\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(idx));
    } else if constexpr (MODE == 8) {  // 8 v_bfi (3-operand plain VOP3)
      asm volatile(REP8("v_bfi_b32 %0, %2, %1, %0\n\t") "" : "+v"(a), "+v"(b) : "v"(c));
    } else if constexpr (MODE == 9) {  // 8 v_fma with an SGPR constant
      asm volatile(REP8("v_fma_f32 %0, %1, %2, %0\n\t") "" : "+v"(a) : "v"(b), "s"(1.0001f));
    }
  }
  p[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

template <int MODE> void run(const char* name, float* d, int bpc) {
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
  printf("%-34s %d waves/SIMD: %6.2f SIMD cycles per instruction\n", name, bpc, ms * 1e-3 * 2.4e9 / iters / 8.0 / bpc);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  for (int w : {1, 4}) {
    run<0>("v_mov_b32 (dependent pair)", d, w);
    run<9>("v_fma_f32 sgpr const (dependent)", d, w);
    run<8>("v_bfi_b32 (dependent)", d, w);
    run<1>("v_mov_b32_dpp", d, w);
    run<4>("v_add_f32_dpp", d, w);
    run<3>("v_cndmask_b32 vcc", d, w);
    run<2>("v_cndmask_b32_dpp vcc", d, w);
    run<5>("v_permlane16_swap", d, w);
    run<6>("v_permlane32_swap", d, w);
    run<7>("ds_bpermute_b32", d, w);
  }
  return 0;
}
