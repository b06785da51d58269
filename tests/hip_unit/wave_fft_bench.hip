// Micro-benchmark of the tdk_wave_fft.h building blocks (run on the GPU box): SIMD pipe cycles per call at
// 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -I torch-darktable_amd/csrc -o tests/hip_unit/build/wave_fft_bench tests/hip_unit/wave_fft_bench.hip
#include <stdio.h>

#include "dpp_transpose.h"

void tdk_set_error(const char*, ...) {}
bool g_tdk_profile_on = false;
bool tdk_timer_begin(const char*, hipStream_t) { return false; }
void tdk_timer_end(hipStream_t) {}

template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  constexpr int K = 32;
  float re[K], im[K];
#pragma unroll
  for (int k = 0; k < K; k++) { re[k] = p[(threadIdx.x * K + k) & 1023]; im[k] = re[k] * 0.5f; }
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) { tdk_fft::transpose_inreg<K>(re); }
    else if constexpr (MODE == 1) { tdk_fft::fft_inreg<K, false>(re, im); }
    else if constexpr (MODE == 2) {
#pragma unroll
      for (int k = 0; k < K; k++) re[k] = __builtin_fmaf(re[k], 1.0001f, im[k]);   // 32 plain FMAs
    } else if constexpr (MODE == 3) {  // LDS transpose of the first kernel generation, for comparison
      __shared__ float buf[4][2 * 32 * 33];
      float* b = buf[threadIdx.x >> 6] + ((threadIdx.x & 63) >> 5) * (32 * 33);
      const int row = threadIdx.x & 31;
#pragma unroll
      for (int k = 0; k < K; k++) b[row * 33 + k] = re[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < K; k++) re[k] = b[k * 33 + row];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    } else if constexpr (MODE == 4) {  // 16 x swap_rows16 only
#pragma unroll
      for (int r = 0; r < 16; r++) tdk_fft::swap_rows16(re[r], re[r | 16]);
    } else if constexpr (MODE == 5) {  // one DPP stage (m = 1) only: 32 v_cndmask_b32_dpp
#pragma unroll
      for (int q = 0; q < 16; q += 4) tdk_fft::xpose4<0>(re[2 * q], re[2 * q + 2], re[2 * q + 4], re[2 * q + 6], re[2 * q + 1], re[2 * q + 3], re[2 * q + 5], re[2 * q + 7]);
    }
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < K; k++) s += re[k] + im[k];
  p[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* d, int blocks_per_cu) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  bench<MODE><<<blocks, 256>>>(d, 10);
  (void)hipEventRecord(a);
  bench<MODE><<<blocks, 256>>>(d, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  // each SIMD hosts blocks_per_cu waves; pipe cycles per call per wave = time * clock / (iters * waves per SIMD)
  const double cyc = ms * 1e-3 * 2.4e9 / iters;
  printf("%-28s %d waves/SIMD: %8.1f cycles per call per SIMD-round, %7.1f per wave-call\n", name, blocks_per_cu, cyc, cyc / blocks_per_cu);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  for (int w : {1, 2, 4}) {
    run<2>("32 plain FMAs", d, w);
    run<1>("fft_inreg<32>", d, w);
    run<0>("transpose_inreg<32> (DPP)", d, w);
    run<4>("16 x permlane16_swap", d, w);
    run<5>("32 x v_cndmask_dpp stage", d, w);
    run<3>("LDS transpose 32x32", d, w);
  }
  return 0;
}
