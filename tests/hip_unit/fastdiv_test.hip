// Unit check: tdk::div_core (the bare core of hipcc's IEEE fp32 division, tdk_fastdiv.h) returns the same BITS as
// `a / b` over the operand ranges its callers guarantee -- denominators in [2^-33, 2^44], numerators +0 or of
// magnitude in [2^-80, 2^41] -- and shows where it does not outside them (tiny numerators, -0).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "tdk_fastdiv.h"

__device__ __forceinline__ uint32_t mix32(uint64_t x) {  // splitmix64 finaliser
  x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return (uint32_t)((x ^ (x >> 31)) >> 16);
}
// float with biased exponent in [elo, ehi], mantissa by `kind`: 0 random, 1 all ones, 2 zero, 3 one low bit, 4 top bits
__device__ __forceinline__ float make(uint32_t r, int elo, int ehi, int kind, bool neg) {
  const uint32_t e = (uint32_t)elo + (r >> 23) % (uint32_t)(ehi - elo + 1);
  uint32_t m = r & 0x7fffffu;
  if (kind == 1) m = 0x7fffffu; else if (kind == 2) m = 0; else if (kind == 3) m = 1u << (r & 3); else if (kind == 4) m &= 0x7f0000u;
  return __builtin_bit_cast(float, (neg ? 0x80000000u : 0u) | (e << 23) | m);
}
// mode 0: guarded range; 1: numerators below 2^-102 (core is expected to differ sometimes); 2: a == -0
__global__ void sweep(unsigned long long* bad, unsigned long long* first, int mode, int iters, uint64_t seed) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; it++) {
    const uint64_t k = (t * (uint64_t)iters + it) * 4 + seed;
    const uint32_t r0 = mix32(k), r1 = mix32(k + 1), r2 = mix32(k + 2);
    const int ka = r2 % 5, kb = (r2 / 5) % 5;
    const float b = make(r1, 127 - 33, 127 + 43, kb, false);
    float a;
    if (mode == 0) a = ((r2 >> 8) % 64 == 0) ? 0.0f : make(r0, 127 - 80, 127 + 40, ka, (r2 >> 16) & 1);
    else if (mode == 1) a = make(r0, 1, 127 - 103, ka, (r2 >> 16) & 1);
    else a = -0.0f;
    const float q = tdk::div_core(a, b), ref = a / b;
    if (__builtin_bit_cast(uint32_t, q) != __builtin_bit_cast(uint32_t, ref)) {
      if (nbad == 0 && mode == 0) atomicCAS(first, 0ull, ((unsigned long long)__builtin_bit_cast(uint32_t, a) << 32) | __builtin_bit_cast(uint32_t, b));
      nbad++;
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}
int main() {
  unsigned long long *d, h[2];
  (void)hipMalloc(&d, 16);
  const char* names[3] = {"guarded range (b in [2^-33,2^44], a = +0 or |a| in [2^-80,2^41])", "numerators below 2^-102 (outside the contract)", "a = -0 (outside the contract)"};
  int rc = 0;
  for (int mode = 0; mode < 3; mode++) {
    (void)hipMemset(d, 0, 16);
    const int blocks = 4096, threads = 256, iters = mode == 0 ? 4096 : 64;
    sweep<<<blocks, threads>>>(d, d + 1, mode, iters, 0x1234567ull * (mode + 1));
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-75s %llu pairs, %llu differ from a / b", names[mode], (unsigned long long)blocks * threads * iters, h[0]);
    if (mode == 0 && h[0]) { printf("  first: a=0x%08llx b=0x%08llx", h[1] >> 32, h[1] & 0xffffffffull); rc = 1; }
    printf("\n");
  }
  printf(rc ? "FAILED\n" : "OK\n");
  return rc;
}
