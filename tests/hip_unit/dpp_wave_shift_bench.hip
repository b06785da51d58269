// Micro-benchmark + semantics check: v_mov_b32_dpp with the wavefront shifts (wave_shr:1 / wave_shl:1) on gfx950 -- do they
// cross the 16-lane rows, and what do they cost next to a row shift and a plain move?
//   hipcc -O3 --offload-arch=gfx950 -o tests/hip_unit/build/dpp_wave_shift_bench tests/hip_unit/dpp_wave_shift_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define D(ctrl, a, b) "v_mov_b32_dpp %" #a ", %" #b " " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define EIGHT(ctrl) D(ctrl, 0, 1) D(ctrl, 2, 3) D(ctrl, 4, 5) D(ctrl, 6, 7) D(ctrl, 1, 0) D(ctrl, 3, 2) D(ctrl, 5, 4) D(ctrl, 7, 6)
template <int MODE> __global__ __launch_bounds__(256) void bench(float* p, int iters) {
  float a = p[threadIdx.x], b = a * 2, c = a * 3, d = a * 4, e = a * 5, f = a * 6, g = a * 7, h = a * 8;
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) asm volatile(EIGHT("row_shr:1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    else if constexpr (MODE == 1) asm volatile(EIGHT("wave_shr:1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    else if constexpr (MODE == 2) asm volatile(EIGHT("wave_shl:1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    else asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %4, %5\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %3, %2\n\tv_mov_b32 %5, %4\n\tv_mov_b32 %7, %6"
                      : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
  }
  p[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

__global__ void semantics(int* out) {
  const int lane = threadIdx.x;
  int r = -1, l = -1;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(lane));
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(l) : "v"(lane));
  out[lane] = r;
  out[64 + lane] = l;
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
  printf("%-26s %d waves/SIMD: %6.2f SIMD cycles per instruction\n", name, bpc, ms * 1e-3 * 2.4e9 / iters / 8.0 / bpc);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
  int* s;
  (void)hipMalloc(&s, 128 * sizeof(int));
  semantics<<<1, 64>>>(s);
  int h[128];
  (void)hipMemcpy(h, s, sizeof h, hipMemcpyDeviceToHost);
  printf("wave_shr:1 -> lane i reads lane:"); for (int i : {0, 1, 15, 16, 17, 31, 32, 33, 63}) printf(" %d:%d", i, h[i]);
  printf("\nwave_shl:1 -> lane i reads lane:"); for (int i : {0, 1, 15, 16, 17, 31, 32, 33, 62, 63}) printf(" %d:%d", i, h[64 + i]);
  printf("\n");
  for (int bpc : {1, 4}) {
    run<3>("v_mov_b32", d, bpc);
    run<0>("v_mov_b32_dpp row_shr:1", d, bpc);
    run<1>("v_mov_b32_dpp wave_shr:1", d, bpc);
    run<2>("v_mov_b32_dpp wave_shl:1", d, bpc);
  }
  return 0;
}
