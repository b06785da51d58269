// Micro-benchmark: does a VALU instruction pay for source VGPRs that share a register bank (index mod 4)?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R8(x) x x x x x x x x
template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  // fixed physical registers: v40..v71 hold data
  asm volatile("v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\tv_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 1.0\n\tv_mov_b32 v46, 1.0\n\tv_mov_b32 v47, 1.0\n\t"
               "v_mov_b32 v48, 1.0\n\tv_mov_b32 v49, 1.0\n\tv_mov_b32 v50, 1.0\n\tv_mov_b32 v51, 1.0\n\tv_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 1.0\n\tv_mov_b32 v54, 1.0\n\tv_mov_b32 v55, 1.0\n\t"
               "v_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\tv_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t"
               ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) {  // two sources in DIFFERENT banks: v40(0) + v45(1) ...
      asm volatile(R8("v_add_f32 v56, v40, v45\n\tv_add_f32 v57, v41, v46\n\tv_add_f32 v58, v42, v47\n\tv_add_f32 v59, v43, v44\n\t")
                   ::: "v56","v57","v58","v59");
    } else if constexpr (MODE == 1) {  // two sources in the SAME bank: v40(0) + v44(0) ...
      asm volatile(R8("v_add_f32 v56, v40, v44\n\tv_add_f32 v57, v41, v45\n\tv_add_f32 v58, v42, v46\n\tv_add_f32 v59, v43, v47\n\t")
                   ::: "v56","v57","v58","v59");
    } else if constexpr (MODE == 2) {  // fma, three sources, all different banks (dst = src2)
      asm volatile(R8("v_fma_f32 v56, v40, v45, v58\n\tv_fma_f32 v57, v41, v46, v59\n\tv_fma_f32 v60, v42, v47, v61\n\tv_fma_f32 v62, v43, v44, v63\n\t")
                   ::: "v56","v57","v60","v62");
    } else if constexpr (MODE == 3) {  // fma, src0 and src1 same bank
      asm volatile(R8("v_fma_f32 v56, v40, v44, v58\n\tv_fma_f32 v57, v41, v45, v59\n\tv_fma_f32 v60, v42, v46, v61\n\tv_fma_f32 v62, v43, v47, v63\n\t")
                   ::: "v56","v57","v60","v62");
    } else if constexpr (MODE == 4) {  // fma, all three sources same bank
      asm volatile(R8("v_fma_f32 v56, v40, v44, v48\n\tv_fma_f32 v57, v41, v45, v49\n\tv_fma_f32 v58, v42, v46, v50\n\tv_fma_f32 v59, v43, v47, v51\n\t")
                   ::: "v56","v57","v58","v59");
    } else if constexpr (MODE == 5) {  // v_fmac (dst is also a source): dst bank == src0 bank
      asm volatile(R8("v_fmac_f32 v56, v40, v45\n\tv_fmac_f32 v57, v41, v46\n\tv_fmac_f32 v58, v42, v47\n\tv_fmac_f32 v59, v43, v44\n\t")
                   ::: "v56","v57","v58","v59");
    } else if constexpr (MODE == 6) {  // v_fmac, src0/src1 same bank as each other and as dst
      asm volatile(R8("v_fmac_f32 v56, v40, v44\n\tv_fmac_f32 v57, v41, v45\n\tv_fmac_f32 v58, v42, v46\n\tv_fmac_f32 v59, v43, v47\n\t")
                   ::: "v56","v57","v58","v59");
    }
  }
  float r;
  asm volatile("v_add_f32 %0, v56, v57\n\tv_add_f32 %0, %0, v58\n\tv_add_f32 %0, %0, v59" : "=v"(r));
  p[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, float* d) {
  printf("%-52s", name);
  for (int bpc : {1, 4, 8}) {
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
  printf("\n");
}
int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  run<0>("v_add_f32  srcs in different banks", d);
  run<1>("v_add_f32  srcs in the SAME bank (idx mod 4)", d);
  run<2>("v_fma_f32  3 srcs, different banks", d);
  run<3>("v_fma_f32  src0/src1 same bank", d);
  run<4>("v_fma_f32  all 3 srcs same bank", d);
  run<5>("v_fmac_f32 srcs different banks", d);
  run<6>("v_fmac_f32 srcs same bank", d);
  return 0;
}
