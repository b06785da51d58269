// color.hip -- per-pixel colour operators and luminance extract / replace.
//
// Replaces reference csrc/color_conversions.cu:78-314 (convert_color_kernel with its nine
// functors, extract_channel_kernel, modify_color_kernel).  Math: tdk_color.h namespace cA
// (the reference's device_conversions.h).
//
// MI355X design: streaming kernels over the flattened pixel list; each thread owns four
// consecutive pixels, i.e. 48 contiguous bytes in and out (three 16-B accesses, 3 KiB per
// wave), grid-strided over at most 8 workgroups per CU.  The reference's 16x16 2-D blocks
// with three scalar 4-B loads per pixel are gone.  Unaligned buffers or a pixel count that
// is not a multiple of four fall back to the one-pixel-per-thread tail kernel.
// color_transform_3x3 reads its matrix on the device (the reference dereferences the device
// pointer on the host, color_conversions.cu:158-159).
#include "tdk_color.h"

namespace {

struct OpArgs {
  float p0, p1, p2;
  const float* matrix;
};

template <int OP> __device__ __forceinline__ f3 apply_op(f3 c, const OpArgs& a, const float m[9]) {
  if constexpr (OP == TDK_RGB_TO_XYZ) return cA::rgb_to_xyz(c);
  else if constexpr (OP == TDK_XYZ_TO_LAB) return cA::xyz_to_lab(c);
  else if constexpr (OP == TDK_LAB_TO_XYZ) return cA::lab_to_xyz(c);
  else if constexpr (OP == TDK_XYZ_TO_RGB) return cA::xyz_to_rgb(c);
  else if constexpr (OP == TDK_RGB_TO_LAB) return cA::rgb_to_lab(c);
  else if constexpr (OP == TDK_LAB_TO_RGB) return cA::lab_to_rgb(c);
  else if constexpr (OP == TDK_MODIFY_HSL) return cA::modify_hsl(c, a.p0, a.p1, a.p2);
  else if constexpr (OP == TDK_MODIFY_VIBRANCE) return cA::vibrance(c, a.p0);
  else return clip3(mat3_mul(m, c));  // device_conversions.h:209-211
}

inline int stream_grid(int64_t nthreads) {
  int64_t b = tdk_div_up64(nthreads, 256);
  return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

template <int OP, typename T> __global__ __launch_bounds__(256) void color_vec4(const T* __restrict__ in, T* __restrict__ out, int64_t ngroups, OpArgs a) {
  float m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (OP == TDK_TRANSFORM_3X3) {
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = a.matrix[k];
  }
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256) {
    float v[12];
    rgb4_io<T>::load(in, g, v);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f3 r = apply_op<OP>(mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2]), a, m);
      v[3 * k] = r.x; v[3 * k + 1] = r.y; v[3 * k + 2] = r.z;
    }
    rgb4_io<T>::store(out, g, v);
  }
}

template <int OP, typename T> __global__ __launch_bounds__(256) void color_tail(const T* __restrict__ in, T* __restrict__ out, int64_t first, int64_t npix, OpArgs a) {
  float m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (OP == TDK_TRANSFORM_3X3) {
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = a.matrix[k];
  }
  for (int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const f3 r = apply_op<OP>(mk3(ld<T>(in, 3 * i), ld<T>(in, 3 * i + 1), ld<T>(in, 3 * i + 2)), a, m);
    st<T>(out, 3 * i, r.x); st<T>(out, 3 * i + 1, r.y); st<T>(out, 3 * i + 2, r.z);
  }
}

template <int OP, typename T> int run_color_t(const T* in, T* out, int64_t npix, OpArgs a, hipStream_t s) {
  int64_t done = 0;
  if (tdk_aligned(in, 4 * sizeof(T)) && tdk_aligned(out, 4 * sizeof(T)) && npix >= 4) {
    const int64_t ng = npix / 4;
    TDK_LAUNCH("tdk_color_op", (color_vec4<OP, T>), dim3(stream_grid(ng)), dim3(256), 0, s, in, out, ng, a);
    done = ng * 4;
  }
  if (done < npix) {
    TDK_LAUNCH("tdk_color_op", (color_tail<OP, T>), dim3(stream_grid(npix - done)), dim3(256), 0, s, in, out, done, npix, a);
  }
  return TDK_OK;
}
template <int OP> int run_color(const void* in, void* out, int64_t npix, OpArgs a, int dtype, hipStream_t s) {
  if (dtype == TDK_F16) return run_color_t<OP, __half>(reinterpret_cast<const __half*>(in), reinterpret_cast<__half*>(out), npix, a, s);
  return run_color_t<OP, float>(reinterpret_cast<const float*>(in), reinterpret_cast<float*>(out), npix, a, s);
}

// ---- luminance extract: rgb (T_RGB) -> plane (T_L)
template <typename TR, typename TL, bool LOG>
__global__ __launch_bounds__(256) void lum_extract_vec4(const TR* __restrict__ rgb, TL* __restrict__ lum, int64_t ngroups, float eps) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256) {
    float v[12], l[4];
    rgb4_io<TR>::load(rgb, g, v);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float y = cA::rgb_to_lab_l(clip3(mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2])));
      l[k] = LOG ? tdk_log(fmaxf(eps, y)) : y;
    }
    s4_io<TL>::store(lum, g, l);
  }
}
template <typename TR, typename TL, bool LOG>
__global__ __launch_bounds__(256) void lum_extract_tail(const TR* __restrict__ rgb, TL* __restrict__ lum, int64_t first, int64_t npix, float eps) {
  for (int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const float y = cA::rgb_to_lab_l(clip3(mk3(ld(rgb, 3 * i), ld(rgb, 3 * i + 1), ld(rgb, 3 * i + 2))));
    st(lum, i, LOG ? tdk_log(fmaxf(eps, y)) : y);
  }
}

// ---- Lab hand-over of the fused chain Wiener.process_log_luminance -> Bilateral.process_rgb (torch_darktable/denoise.py:54-58,
// local_contrast.py:109-114 in the reference).  The two stages each convert the RGB pixel to Lab, replace L and convert back
// (device_conversions.h:213-225); between them only L changes, so the chain can carry the pixel as (L, a, b): this kernel
// emits the log-lightness plane the denoiser works on AND the pixel's Lab chroma (a, b), the denoiser's finish kernel
// (wiener.hip: wiener_finish_lab) turns the denoised log-lightness into the lightness of its result (re-deriving L, a, b only
// for the pixels the reference's clip to [0, 1] changes), and the bilateral epilogue converts (L'', a, b) to RGB once.
// Per pixel 13 + 1 + 6 transcendentals instead of 9 + 27 + 18.  Same values as the two-stage chain up to the rounding of the
// skipped sRGB encode -> (store) -> decode round trip: parity by the colour operators' tolerance (2e-5), not bit for bit.
// bounds (nullable): the pipeline's normalize_image, (x - bounds[0]) / (bounds[1] - bounds[0]) (reference pipeline/util.py:8-10), applied
// to the samples as they are loaded -- the same IEEE expression as normalize_kernel below, so for float32 images the result is
// that of normalize_image followed by this kernel, without the normalised image ever being stored
template <typename TR, int VEC>
__global__ __launch_bounds__(256) void lum_lab_extract(const TR* __restrict__ rgb, float* __restrict__ loglum, float* __restrict__ ab, int64_t first,
                                                       int64_t ngroups, float eps, const float* __restrict__ bounds) {
  TDK_STREAMING_KERNEL_PROLOGUE();
  const float b0 = bounds ? bounds[0] : 0.0f, range = bounds ? bounds[1] - bounds[0] : 1.0f;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256) {
    float v[3 * VEC], l[VEC], c2[2 * VEC];
    if constexpr (VEC == 4) rgb4_io<TR>::load(rgb, g, v);
    else { v[0] = ld(rgb, 3 * (first + g)); v[1] = ld(rgb, 3 * (first + g) + 1); v[2] = ld(rgb, 3 * (first + g) + 2); }
    if (bounds) {  // kernel argument: uniform
#pragma unroll
      for (int k = 0; k < 3 * VEC; k++) v[k] = (v[k] - b0) / range;
    }
    bool outside = false;
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      const f3 c = mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2]);
      const f3 lab = cA::rgb_to_lab(c);  // of the pixel as it is (modify_log_luminance does not clip its input)
      l[k] = fmaxf(0.0f, lab.x);         // == rgb_to_lab_l(clip3(c)) when no channel leaves [0, 1]: the same Y, the same lab_f
      c2[2 * k] = lab.y; c2[2 * k + 1] = lab.z;
      outside = outside || !(c.x >= 0.0f && c.x <= 1.0f && c.y >= 0.0f && c.y <= 1.0f && c.z >= 0.0f && c.z <= 1.0f);
    }
    if (__builtin_amdgcn_ballot_w64(outside) != 0) {  // compute_log_luminance clips the pixel first (device_conversions.h:197-207)
#pragma unroll
      for (int k = 0; k < VEC; k++) l[k] = cA::rgb_to_lab_l(clip3(mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2])));
    }
#pragma unroll
    for (int k = 0; k < VEC; k++) l[k] = tdk_log(fmaxf(eps, l[k]));
    if constexpr (VEC == 4) {
      s4_io<float>::store(loglum, g, l);
      s4_io<float>::store(ab, 2 * g, c2);
      s4_io<float>::store(ab, 2 * g + 1, c2 + 4);
    } else {
      loglum[first + g] = l[0];
      ab[2 * (first + g)] = c2[0]; ab[2 * (first + g) + 1] = c2[1];
    }
  }
}

// ---- luminance replace
template <typename TR, typename TL, bool LOG>
__global__ __launch_bounds__(256) void lum_modify_vec4(const TR* __restrict__ rgb, const TL* __restrict__ lum, TR* __restrict__ out, int64_t ngroups) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256) {
    float v[12], l[4];
    rgb4_io<TR>::load(rgb, g, v);
    s4_io<TL>::load(lum, g, l);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f3 c = mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2]);
      const f3 r = LOG ? cA::modify_log_luminance(c, l[k]) : cA::modify_luminance(c, l[k]);
      v[3 * k] = r.x; v[3 * k + 1] = r.y; v[3 * k + 2] = r.z;
    }
    rgb4_io<TR>::store(out, g, v);
  }
}
template <typename TR, typename TL, bool LOG>
__global__ __launch_bounds__(256) void lum_modify_tail(const TR* __restrict__ rgb, const TL* __restrict__ lum, TR* __restrict__ out, int64_t first, int64_t npix) {
  for (int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const f3 c = mk3(ld(rgb, 3 * i), ld(rgb, 3 * i + 1), ld(rgb, 3 * i + 2));
    const f3 r = LOG ? cA::modify_log_luminance(c, ld(lum, i)) : cA::modify_luminance(c, ld(lum, i));
    st(out, 3 * i, r.x); st(out, 3 * i + 1, r.y); st(out, 3 * i + 2, r.z);
  }
}

template <typename TR, typename TL, bool LOG>
int run_extract(const void* rgb_, void* lum_, int64_t npix, float eps, hipStream_t s) {
  const TR* rgb = reinterpret_cast<const TR*>(rgb_);
  TL* lum = reinterpret_cast<TL*>(lum_);
  int64_t done = 0;
  if (tdk_aligned(rgb, 16) && tdk_aligned(lum, 16) && npix >= 4) {
    const int64_t ng = npix / 4;
    TDK_LAUNCH("tdk_compute_luminance", (lum_extract_vec4<TR, TL, LOG>), dim3(stream_grid(ng)), dim3(256), 0, s, rgb, lum, ng, eps);
    done = ng * 4;
  }
  if (done < npix) {
    TDK_LAUNCH("tdk_compute_luminance", (lum_extract_tail<TR, TL, LOG>), dim3(stream_grid(npix - done)), dim3(256), 0, s, rgb, lum, done, npix, eps);
  }
  return TDK_OK;
}

template <typename TR, typename TL, bool LOG>
int run_modify(const void* rgb_, const void* lum_, void* out_, int64_t npix, hipStream_t s) {
  const TR* rgb = reinterpret_cast<const TR*>(rgb_);
  const TL* lum = reinterpret_cast<const TL*>(lum_);
  TR* out = reinterpret_cast<TR*>(out_);
  int64_t done = 0;
  if (tdk_aligned(rgb, 16) && tdk_aligned(lum, 16) && tdk_aligned(out, 16) && npix >= 4) {
    const int64_t ng = npix / 4;
    TDK_LAUNCH("tdk_modify_luminance", (lum_modify_vec4<TR, TL, LOG>), dim3(stream_grid(ng)), dim3(256), 0, s, rgb, lum, out, ng);
    done = ng * 4;
  }
  if (done < npix) {
    TDK_LAUNCH("tdk_modify_luminance", (lum_modify_tail<TR, TL, LOG>), dim3(stream_grid(npix - done)), dim3(256), 0, s, rgb, lum, out, done, npix);
  }
  return TDK_OK;
}

// normalize_image of the pipeline (reference torch_darktable/pipeline/util.py:8-10, a torch.compile'd
// elementwise expression): (x - bounds[0]) / (bounds[1] - bounds[0]) with the bounds on the device.
template <typename T>
__global__ __launch_bounds__(256) void normalize_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t n, const float* __restrict__ bounds) {
  const float b0 = bounds[0], range = bounds[1] - bounds[0];
  const int64_t n4 = n / 4;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n4; g += (int64_t)gridDim.x * 256) {
    float v[4];
    s4_io<T>::load(in, (size_t)g, v);
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (v[k] - b0) / range;
    s4_io<T>::store(out, (size_t)g, v);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) st(out, (size_t)i, (ld(in, (size_t)i) - b0) / range);
}

}  // namespace

TDK_EXPORT int tdk_color_op_ex(const void* in, void* out, int64_t npix, int op, const float host_params[3], const float* device_matrix, int dtype,
                               tdk_stream_t stream) {
  TDK_REQUIRE(npix >= 0, "tdk_color_op: negative pixel count");
  TDK_REQUIRE(dtype == TDK_F32 || dtype == TDK_F16, "unsupported dtype tag %d", dtype);
  if (npix == 0) return TDK_OK;
  TDK_REQUIRE(in && out, "tdk_color_op: null pointer");
  OpArgs a{0.0f, 0.0f, 0.0f, device_matrix};
  if (host_params) { a.p0 = host_params[0]; a.p1 = host_params[1]; a.p2 = host_params[2]; }
  hipStream_t s = tdk_stream(stream);
  switch (op) {
    case TDK_RGB_TO_XYZ: return run_color<TDK_RGB_TO_XYZ>(in, out, npix, a, dtype, s);
    case TDK_XYZ_TO_LAB: return run_color<TDK_XYZ_TO_LAB>(in, out, npix, a, dtype, s);
    case TDK_LAB_TO_XYZ: return run_color<TDK_LAB_TO_XYZ>(in, out, npix, a, dtype, s);
    case TDK_XYZ_TO_RGB: return run_color<TDK_XYZ_TO_RGB>(in, out, npix, a, dtype, s);
    case TDK_RGB_TO_LAB: return run_color<TDK_RGB_TO_LAB>(in, out, npix, a, dtype, s);
    case TDK_LAB_TO_RGB: return run_color<TDK_LAB_TO_RGB>(in, out, npix, a, dtype, s);
    case TDK_MODIFY_HSL: return run_color<TDK_MODIFY_HSL>(in, out, npix, a, dtype, s);
    case TDK_MODIFY_VIBRANCE: return run_color<TDK_MODIFY_VIBRANCE>(in, out, npix, a, dtype, s);
    case TDK_TRANSFORM_3X3:
      TDK_REQUIRE(device_matrix != nullptr, "tdk_color_op: TDK_TRANSFORM_3X3 needs a device matrix");
      return run_color<TDK_TRANSFORM_3X3>(in, out, npix, a, dtype, s);
    default: tdk_set_error("tdk_color_op: unknown op %d", op); return TDK_ERR_INVALID_ARGUMENT;
  }
}

TDK_EXPORT int tdk_color_op(const float* in, float* out, int64_t npix, int op, const float host_params[3], const float* device_matrix,
                            tdk_stream_t stream) {
  return tdk_color_op_ex(in, out, npix, op, host_params, device_matrix, TDK_F32, stream);
}

#define TDK_LUM_DISPATCH(FN, ...)                                                                    \
  do {                                                                                               \
    if (rgb_dtype == TDK_F32 && lum_dtype == TDK_F32) return log_mode ? FN<float, float, true>(__VA_ARGS__) : FN<float, float, false>(__VA_ARGS__); \
    if (rgb_dtype == TDK_F16 && lum_dtype == TDK_F32) return log_mode ? FN<__half, float, true>(__VA_ARGS__) : FN<__half, float, false>(__VA_ARGS__); \
    if (rgb_dtype == TDK_F16 && lum_dtype == TDK_F16) return log_mode ? FN<__half, __half, true>(__VA_ARGS__) : FN<__half, __half, false>(__VA_ARGS__); \
    if (rgb_dtype == TDK_F32 && lum_dtype == TDK_F16) return log_mode ? FN<float, __half, true>(__VA_ARGS__) : FN<float, __half, false>(__VA_ARGS__); \
    tdk_set_error("unsupported dtype combination (%d, %d)", rgb_dtype, lum_dtype);                   \
    return TDK_ERR_INVALID_ARGUMENT;                                                                 \
  } while (0)

TDK_EXPORT int tdk_compute_luminance(const void* rgb, void* lum, int64_t npix, int log_mode, float eps, int rgb_dtype, int lum_dtype,
                                     tdk_stream_t stream) {
  TDK_REQUIRE(npix >= 0, "tdk_compute_luminance: negative pixel count");
  if (npix == 0) return TDK_OK;
  TDK_REQUIRE(rgb && lum, "tdk_compute_luminance: null pointer");
  TDK_REQUIRE(!log_mode || eps > 0.0f, "Epsilon must be positive");
  TDK_LUM_DISPATCH(run_extract, rgb, lum, npix, eps, tdk_stream(stream));
}

TDK_EXPORT int tdk_compute_log_luminance_lab(const void* rgb, float* loglum, float* ab, int64_t npix, float eps, const float* bounds, int rgb_dtype,
                                             tdk_stream_t stream) {
  TDK_REQUIRE(npix >= 0, "tdk_compute_log_luminance_lab: negative pixel count");
  if (npix == 0) return TDK_OK;
  TDK_REQUIRE(rgb && loglum && ab, "tdk_compute_log_luminance_lab: null pointer");
  TDK_REQUIRE(eps > 0.0f, "Epsilon must be positive");
  hipStream_t s = tdk_stream(stream);
  TDK_DISPATCH_DTYPE(rgb_dtype, T, {
    const T* in = reinterpret_cast<const T*>(rgb);
    int64_t done = 0;
    if (tdk_aligned(rgb, 16) && tdk_aligned(loglum, 16) && tdk_aligned(ab, 16) && npix >= 4) {
      const int64_t ng = npix / 4;
      TDK_LAUNCH("tdk_compute_luminance(lab)", (lum_lab_extract<T, 4>), dim3(stream_grid(ng)), dim3(256), 0, s, in, loglum, ab, (int64_t)0, ng, eps, bounds);
      done = ng * 4;
    }
    if (done < npix) TDK_LAUNCH("tdk_compute_luminance(lab)", (lum_lab_extract<T, 1>), dim3(stream_grid(npix - done)), dim3(256), 0, s, in, loglum, ab, done, npix - done, eps, bounds);
  });
  return TDK_OK;
}

TDK_EXPORT int tdk_modify_luminance(const void* rgb, const void* lum, void* rgb_out, int64_t npix, int log_mode, int rgb_dtype,
                                    int lum_dtype, tdk_stream_t stream) {
  TDK_REQUIRE(npix >= 0, "tdk_modify_luminance: negative pixel count");
  if (npix == 0) return TDK_OK;
  TDK_REQUIRE(rgb && lum && rgb_out, "tdk_modify_luminance: null pointer");
  TDK_LUM_DISPATCH(run_modify, rgb, lum, rgb_out, npix, tdk_stream(stream));
}

TDK_EXPORT int tdk_normalize(const void* in, void* out, int64_t count, const float* bounds, int dtype, tdk_stream_t stream) {
  TDK_REQUIRE(count >= 0, "tdk_normalize: negative element count");
  if (count == 0) return TDK_OK;
  TDK_REQUIRE(in && out && bounds, "tdk_normalize: null pointer");
  TDK_REQUIRE(tdk_aligned(in, 16) && tdk_aligned(out, 16), "tdk_normalize: buffers must be 16-byte aligned");
  TDK_DISPATCH_DTYPE(dtype, T, TDK_LAUNCH("tdk_normalize", normalize_kernel<T>, dim3(stream_grid(count / 4 + 1)), dim3(256), 0, tdk_stream(stream),
                                                  reinterpret_cast<const T*>(in), reinterpret_cast<T*>(out), count, bounds));
  return TDK_OK;
}
