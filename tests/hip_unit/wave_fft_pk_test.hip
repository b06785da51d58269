// Stand-alone GPU test + timing of tdk_wave_fft_pk.h (packed-fp32 in-register FFT) against a host DFT and against the
// plain-instruction fft_inreg:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I torch-darktable_amd/csrc -o tests/hip_unit/build/wave_fft_pk_test tests/hip_unit/wave_fft_pk_test.hip
#include <math.h>
#include <stdio.h>

#include <vector>

#include "tdk_wave_fft_pk.h"

void tdk_set_error(const char*, ...) {}
bool g_tdk_profile_on = false;
bool tdk_timer_begin(const char*, hipStream_t) { return false; }
void tdk_timer_end(hipStream_t) {}
int tdk_raise_lds_limit(const void*, int, const char*) { return 0; }

using tdk_fft::v2f;

template <bool INV, bool PK> __global__ void k_fft(const float* in, float* out, int reps) {
  constexpr int K = 32;
  if constexpr (PK) {
    v2f z[K];
#pragma unroll
    for (int k = 0; k < K; k++) z[k] = v2f{in[(threadIdx.x * K + k) * 2], in[(threadIdx.x * K + k) * 2 + 1]};
    for (int r = 0; r < reps; r++) tdk_fft::fft_inreg_pk<K, INV>(z);
#pragma unroll
    for (int k = 0; k < K; k++) { out[(threadIdx.x * K + k) * 2] = z[k].x; out[(threadIdx.x * K + k) * 2 + 1] = z[k].y; }
  } else {
    float re[K], im[K];
#pragma unroll
    for (int k = 0; k < K; k++) { re[k] = in[(threadIdx.x * K + k) * 2]; im[k] = in[(threadIdx.x * K + k) * 2 + 1]; }
    for (int r = 0; r < reps; r++) tdk_fft::fft_inreg<K, INV>(re, im);
#pragma unroll
    for (int k = 0; k < K; k++) { out[(threadIdx.x * K + k) * 2] = re[k]; out[(threadIdx.x * K + k) * 2 + 1] = im[k]; }
  }
}

template <bool INV> int check() {
  constexpr int K = 32, n = 64 * K * 2;
  std::vector<float> h(n), o(n), o2(n);
  srand(7);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *d_in, *d_out;
  hipMalloc(&d_in, n * 4); hipMalloc(&d_out, n * 4);
  hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
  k_fft<INV, true><<<1, 64>>>(d_in, d_out, 1);
  hipMemcpy(o.data(), d_out, n * 4, hipMemcpyDeviceToHost);
  k_fft<INV, false><<<1, 64>>>(d_in, d_out, 1);
  hipMemcpy(o2.data(), d_out, n * 4, hipMemcpyDeviceToHost);
  double worst = 0, worst2 = 0;
  for (int lane = 0; lane < 64; lane++)
    for (int f = 0; f < K; f++) {
      double sr = 0, si = 0;
      for (int k = 0; k < K; k++) {
        const double a = (INV ? 2.0 : -2.0) * M_PI * f * k / K, xr = h[(lane * K + k) * 2], xi = h[(lane * K + k) * 2 + 1];
        sr += xr * cos(a) - xi * sin(a);
        si += xr * sin(a) + xi * cos(a);
      }
      worst = fmax(worst, fmax(fabs(sr - o[(lane * K + f) * 2]), fabs(si - o[(lane * K + f) * 2 + 1])));
      worst2 = fmax(worst2, fmax(fabs((double)o2[(lane * K + f) * 2] - o[(lane * K + f) * 2]), fabs((double)o2[(lane * K + f) * 2 + 1] - o[(lane * K + f) * 2 + 1])));
    }
  printf("fft_inreg_pk<32, %s>: max |err| vs host DFT %.3e, vs fft_inreg %.3e  %s\n", INV ? "inverse" : "forward", worst, worst2, worst < 2e-5 ? "ok" : "FAIL");
  hipFree(d_in); hipFree(d_out);
  return worst < 2e-5 ? 0 : 1;
}

template <bool PK> void timeit(const char* name, float* d) {
  printf("%-22s", name);
  for (int bpc : {1, 2, 4}) {
    const int reps = 4000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_fft<false, PK><<<256 * bpc, 256>>>(d, d + (1 << 20), 10);
    (void)hipEventRecord(a);
    k_fft<false, PK><<<256 * bpc, 256>>>(d, d + (1 << 20), reps);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9 / reps;
    printf("  %dw/SIMD: %7.1f cycles per FFT per wave, %7.1f per SIMD-FFT", bpc, cyc, cyc / bpc);
  }
  printf("\n");
}

int main() {
  int bad = check<false>() + check<true>();
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  timeit<false>("fft_inreg<32> plain", d);
  timeit<true>("fft_inreg_pk<32>", d);
  return bad;
}
