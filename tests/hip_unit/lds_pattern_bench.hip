// Micro-benchmark: LDS cycles per wave-instruction for the access shapes of the RCD tile kernel (gfx950).
// One 1024-thread workgroup per CU on a 84 x 88-float plane (row stride 88), every wave issuing the same
// pattern back to back; result = CU cycles per wave-instruction (ideal: ds_read_b32 2, ds_read_b128 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int S = 88, ROWS = 84, NT = 1024;
template <int MODE> __device__ __forceinline__ int lane_offset(int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  if (MODE == 0) { const int i = tid % (78 * 10); return (3 + i / 78) * S + 3 + i % 78; }                 // b32: 78-wide rows, lanes along the row
  if (MODE == 1) { return 3 * S + 4 * tid; }                                                                        // b128: fully contiguous 16-B pieces
  if (MODE == 2) { const int b = tid % (19 * 39); return (3 + 2 * (b / 19)) * S + 4 + 4 * (b % 19); }       // b128: 19 blocks per row, lanes along the row
  if (MODE == 3) { const int bx = 4 * (wave % 5) + (lane & 3), by = 16 * (wave / 5) % 32 + (lane >> 2); return (3 + 2 * by) * S + 4 + 4 * (bx % 19); }  // b128: 4 x 16 per wave
  if (MODE == 4) { const int bx = 16 * (wave % 2) % 19 + (lane & 15), by = 4 * (wave / 2) + (lane >> 4); return (3 + 2 * by) * S + 4 + 4 * (bx % 19); } // b128: 16 x 4 per wave
  if (MODE == 5) { return 2 * tid; }                                                                        // b64 contiguous
  if (MODE == 6) { const int i = tid % (74 * 12); const int c = 5 + i % 74; return (5 + 2 * (i / 74) + (c & 1)) * S + c; }  // b32 checkerboard rows
  return tid;
}
template <int MODE> __global__ __launch_bounds__(NT) void bench(float* out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < ROWS * S * 5; i += NT) lds[i] = (float)i;
  __syncthreads();
  const int off = lane_offset<MODE>(threadIdx.x);
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    int o = off;
    asm volatile("" : "+v"(o));  // opaque per iteration: no hoisting of the reads
    const float* p = lds + o + (it & 3) * ROWS * S;
    if (MODE == 0 || MODE == 6) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = p[(k - 3) * (MODE == 0 ? S : 2 * S)];
#pragma unroll
      for (int k = 0; k < 8; k++) acc += v[k];
    } else if (MODE == 5) {
      float2 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { v[k] = *reinterpret_cast<const float2*>(p + k * S * 8); asm volatile("" : "+v"(v[k].x), "+v"(v[k].y)); }
#pragma unroll
      for (int k = 0; k < 8; k++) acc += v[k].x + v[k].y;
    } else {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { v[k] = *reinterpret_cast<const float4*>(p + (k - 3) * S); asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w)); }
#pragma unroll
      for (int k = 0; k < 8; k++) acc += v[k].x + v[k].w;
    }
  }
  out[blockIdx.x * NT + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name, float* d, int per_iter) {
  const int iters = 4000, blocks = 256;
  const size_t lds = (size_t)ROWS * S * 5 * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  bench<MODE><<<blocks, NT, lds>>>(d, iters / 4);
  (void)hipEventRecord(a);
  bench<MODE><<<blocks, NT, lds>>>(d, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  printf("%-62s %6.2f CU cycles per wave-instruction @2.4 GHz\n", name, ms * 1e-3 * 2.4e9 / iters / per_iter / 16.0);
}
int main() {
  float* d;
  (void)hipMalloc(&d, 256 * NT * 4);
  run<0>("ds_read_b32, lanes along 78-wide rows (vertical taps)", d, 8);
  run<6>("ds_read_b32, checkerboard lanes on a 2-row band", d, 8);
  run<5>("ds_read_b64, contiguous", d, 8);
  run<1>("ds_read_b128, contiguous", d, 8);
  run<2>("ds_read_b128, 19 four-column blocks per row, lanes along rows", d, 8);
  run<3>("ds_read_b128, wave = 4 block columns x 16 block rows", d, 8);
  run<4>("ds_read_b128, wave = 16 block columns x 4 block rows", d, 8);
  return 0;
}
