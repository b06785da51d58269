// Micro-benchmark: cost of an IEEE fp32 division (the compiler's v_div_scale / v_div_fmas / v_div_fixup sequence)
// against rcp-based forms, per wave64 division, 8 independent streams per lane (gfx950).  Numerator and
// denominator swap roles every division (r = r / d; d = d / r), so nothing of a division is loop-invariant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "tdk_fastdiv.h"
template <int MODE> __device__ __forceinline__ float dv(float a, float b) {
  if constexpr (MODE == 0) return a / b;                                   // IEEE (default flags: correctly rounded)
  else if constexpr (MODE == 1) return a * __builtin_amdgcn_rcpf(b);       // fast, ~1 ulp
  else return tdk::div_core(a, b);                                          // the bare core of the IEEE expansion (exact in range)
}
template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  float r[8], d[8];
  for (int i = 0; i < 8; i++) { r[i] = p[threadIdx.x + i] + 1.0f; d[i] = 1.0f + 1e-3f * (float)(i + 1) + p[threadIdx.x + 9]; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      r[i] = dv<MODE>(r[i], d[i]);
      d[i] = dv<MODE>(d[i], r[i]);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += r[i];
  p[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d) {
  printf("%-40s", name);
  for (int bpc : {1, 4, 8}) {
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
    printf("  %dw: %6.1f", bpc, ms * 1e-3 * 2.4e9 / iters / 16.0 / bpc);
  }
  printf("   (SIMD cycles per wave64 division @2.4 GHz)\n");
}
int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  run<0>("a / b (IEEE, div_scale/fmas/fixup)", d);
  run<1>("a * rcp(b)", d);
  run<2>("tdk::div_core (rcp + 7 fma/mul)", d);
  return 0;
}
