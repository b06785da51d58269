// Stand-alone GPU unit test of tdk_wave_fft.h (run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 -I torch-darktable_amd/csrc -o tests/hip_unit/build/wave_fft_test tests/hip_unit/wave_fft_test.hip
//   tests/hip_unit/build/wave_fft_test
// Checks the in-register K x K transposition (K = 16, 32) element by element and the forward + inverse
// in-register FFT against a host DFT.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "dpp_transpose.h"

void tdk_set_error(const char*, ...) {}
bool g_tdk_profile_on = false;
bool tdk_timer_begin(const char*, hipStream_t) { return false; }
void tdk_timer_end(hipStream_t) {}

template <int K> __global__ void k_transpose(const float* in, float* out) {
  float v[K];
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = in[threadIdx.x * K + k];
  tdk_fft::transpose_inreg<K>(v);
#pragma unroll
  for (int k = 0; k < K; k++) out[threadIdx.x * K + k] = v[k];
}

template <int K, bool INV> __global__ void k_fft(const float* re_in, const float* im_in, float* re_out, float* im_out) {
  float re[K], im[K];
#pragma unroll
  for (int k = 0; k < K; k++) { re[k] = re_in[threadIdx.x * K + k]; im[k] = im_in[threadIdx.x * K + k]; }
  tdk_fft::fft_inreg<K, INV>(re, im);
#pragma unroll
  for (int k = 0; k < K; k++) { re_out[threadIdx.x * K + k] = re[k]; im_out[threadIdx.x * K + k] = im[k]; }
}

template <int K> int test_transpose() {
  const int n = 64 * K;
  std::vector<float> h(n), o(n);
  for (int i = 0; i < n; i++) h[i] = (float)i;
  float *d_in, *d_out;
  hipMalloc(&d_in, n * 4); hipMalloc(&d_out, n * 4);
  hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
  k_transpose<K><<<1, 64>>>(d_in, d_out);
  hipMemcpy(o.data(), d_out, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; lane++)
    for (int k = 0; k < K; k++) {
      const int slot = lane / K, r = lane % K;
      const float want = h[(slot * K + k) * K + r];  // v[k] of lane r  <-  v[r] of lane k (same slot)
      if (o[lane * K + k] != want && bad++ < 8) printf("K=%d lane %d reg %d: got %g want %g\n", K, lane, k, o[lane * K + k], want);
    }
  printf("transpose K=%d: %s (%d bad)\n", K, bad ? "FAIL" : "ok", bad);
  hipFree(d_in); hipFree(d_out);
  return bad;
}

template <int K> int test_fft() {
  const int n = 64 * K;
  std::vector<float> hr(n), hi(n), orr(n), oi(n);
  srand(7);
  for (int i = 0; i < n; i++) { hr[i] = rand() / (float)RAND_MAX - 0.5f; hi[i] = rand() / (float)RAND_MAX - 0.5f; }
  float *dr, *di, *er, *ei;
  hipMalloc(&dr, n * 4); hipMalloc(&di, n * 4); hipMalloc(&er, n * 4); hipMalloc(&ei, n * 4);
  hipMemcpy(dr, hr.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(di, hi.data(), n * 4, hipMemcpyHostToDevice);
  int bad = 0;
  for (int inv = 0; inv < 2; inv++) {
    if (inv) k_fft<K, true><<<1, 64>>>(dr, di, er, ei); else k_fft<K, false><<<1, 64>>>(dr, di, er, ei);
    hipMemcpy(orr.data(), er, n * 4, hipMemcpyDeviceToHost); hipMemcpy(oi.data(), ei, n * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int lane = 0; lane < 64; lane++)
      for (int f = 0; f < K; f++) {
        double sr = 0, si = 0;
        for (int t = 0; t < K; t++) {
          const double a = (inv ? 2.0 : -2.0) * M_PI * f * t / K;
          sr += hr[lane * K + t] * cos(a) - hi[lane * K + t] * sin(a);
          si += hr[lane * K + t] * sin(a) + hi[lane * K + t] * cos(a);
        }
        worst = fmax(worst, fmax(fabs(sr - orr[lane * K + f]), fabs(si - oi[lane * K + f])));
      }
    printf("fft K=%d inv=%d: max err %.3g %s\n", K, inv, worst, worst < 2e-5 ? "ok" : "FAIL");
    bad += worst >= 2e-5;
  }
  return bad;
}

int main() {
  int bad = 0;
  bad += test_transpose<16>();
  bad += test_transpose<32>();
  bad += test_fft<16>();
  bad += test_fft<32>();
  printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad ? 1 : 0;
}
