/*
 * tdk_hip.h -- C ABI of libtdk_hip.so, the MI355X (gfx950) RAW-ISP kernel library.
 *
 * This is the drop-in boundary for the hot path of uc-vision/torch-darktable.  Each entry
 * point replaces one op (or workspace method) that the reference registers on its pybind11
 * module `torch_darktable.torch_darktable_extension` (reference csrc/extension.cpp:50-248,
 * typed by torch_darktable_extension.pyi); the reference file:line each one stands in for
 * is cited on the declaration.  A binding (ctypes / cgo / JNI / pybind) allocates outputs,
 * selects the device, and passes raw device pointers + sizes + the HIP stream.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer on the current HIP device unless named host_*;
 *  - images are row-major HWC (channels innermost), `width`/`height` in pixels;
 *  - `dtype` selects the STORAGE type of image operands: TDK_F32 (the reference's only
 *    mode) or TDK_F16 (extension: fp16 storage, fp32 arithmetic);
 *  - `pattern` is the reference's BayerPattern word (csrc/debayer/demosaic.h:7-12);
 *  - `stream` is a hipStream_t (NULL = the legacy default stream); all work is enqueued
 *    asynchronously, nothing synchronises the host, nothing allocates device memory;
 *  - scratch comes from the caller: tdk_*_workspace_bytes() reports the size, the caller
 *    passes a device buffer of at least that many bytes, 256-byte aligned;
 *  - return value: TDK_OK (0) or a tdk_status code; tdk_last_error() gives the message
 *    of the last failure on the calling thread.
 */
#ifndef TDK_HIP_H
#define TDK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDK_ABI_VERSION 4

typedef void* tdk_stream_t; /* hipStream_t */

enum tdk_status { TDK_OK = 0, TDK_ERR_INVALID_ARGUMENT = 1, TDK_ERR_LAUNCH = 2, TDK_ERR_UNSUPPORTED = 3 };
enum tdk_dtype { TDK_F32 = 0, TDK_F16 = 1 };

#define TDK_PATTERN_RGGB 0x94949494u
#define TDK_PATTERN_BGGR 0x16161616u
#define TDK_PATTERN_GRBG 0x61616161u
#define TDK_PATTERN_GBRG 0x49494949u

int tdk_abi_version(void);
const char* tdk_last_error(void);

/* Optional per-kernel device timing (the reference's counterpart: CudaTimer, csrc/cuda_utils.h:40-85).
 * While enabled every kernel launch is bracketed by two events on its stream.
 * tdk_profile_report blocks until they complete, writes "name launches total_ms\n" lines
 * (NUL-terminated, truncated to cap) and returns the bytes needed.  Enabling clears old records.
 * tdk_profile_filter restricts the timer to launches whose name contains `substr` (NULL or "" =
 * all), so one kernel can be timed inside a throughput run without events between the others. */
int tdk_profile_enable(int on);
int tdk_profile_filter(const char* substr);
int64_t tdk_profile_report(char* buf, int64_t cap);

/* ---- 12-bit packed raw codec: reference csrc/packed.cu:158-280 (extension.cpp:159-169).
 * Flat buffers, `num_pairs` pixel pairs <-> 3 * num_pairs bytes. */
int tdk_encode12_u16(const uint16_t* in, uint8_t* out, int64_t num_pairs, int ids_format, tdk_stream_t stream);
int tdk_encode12_f32(const float* in, uint8_t* out, int64_t num_pairs, int ids_format, int scaled, tdk_stream_t stream);
int tdk_decode12_f32(const uint8_t* in, float* out, int64_t num_pairs, int ids_format, int scaled, tdk_stream_t stream);
int tdk_decode12_f16(const uint8_t* in, void* out_half, int64_t num_pairs, int ids_format, int scaled, tdk_stream_t stream);
int tdk_decode12_u16(const uint8_t* in, uint16_t* out, int64_t num_pairs, int ids_format, tdk_stream_t stream);

/* ---- demosaic.  bayer: (H, W) ; rgb: (H, W, 3) */
/* bilinear5x5_demosaic: reference csrc/debayer/bilinear.cu:104-148 (extension.cpp:205-206) */
int tdk_bilinear5x5(const void* bayer, void* rgb, int width, int height, uint32_t pattern, int dtype, tdk_stream_t stream);

/* PPG.process: reference csrc/debayer/ppg.cu:413-464 (extension.cpp:57-65).
 * median_threshold > 0 enables the pre-median and needs width*height*4 bytes of scratch. */
size_t tdk_ppg_workspace_bytes(int width, int height, float median_threshold);
int tdk_ppg(const void* bayer, void* rgb, void* workspace, int width, int height, uint32_t pattern, float median_threshold,
            int dtype, tdk_stream_t stream);

/* RCD.process: reference csrc/debayer/rcd.cu:601-671 (extension.cpp:67-74).  Pure function
 * with the reference's first-call (zero-initialised scratch) semantics.  width must be even. */
size_t tdk_rcd_workspace_bytes(int width, int height);
int tdk_rcd(const void* bayer, void* rgb, void* workspace, int width, int height, uint32_t pattern, int dtype, tdk_stream_t stream);
/* The same with per-call flags.  Frames of at least 128 x 64 whose rows load and store as sample pairs run as column strips
 * walked down the frame (csrc/tdk_rcd_stream.h), the others as 64 x 64 LDS tiles; TDK_RCD_TILE_KERNEL takes the tile kernel
 * for any frame.  Both give the same bits (the GPU tests compare them through this flag); nothing is process-global. */
#define TDK_RCD_TILE_KERNEL 1u
/* TDK_RCD_CONCURRENT: the caller keeps other kernels in flight on other streams (several frames at once).  The column strips
 * then run as their register-blocked variant (csrc/tdk_rcd_quad.h): the same bits on half the waves per CU, which is slower
 * with the GPU to itself and faster for the whole when other frames' kernels can use the wave slots and registers it leaves. */
#define TDK_RCD_CONCURRENT 2u
/* Arithmetic of a binary16 (TDK_F16) result.  float32 results are always the oracle's bits (its operation order, no
 * contraction, correctly rounded quotients).  A binary16 result of the column strips is by default computed with the
 * approximate flavour (csrc/tdk_rcd_stream.h: a * v_rcp_f32(b) quotients, fused sums of products -- the class of arithmetic of
 * the reference's own nvcc --use_fast_math build, setup.py:36): a few fp32 ulps before the store, i.e. one binary16 ulp on
 * ~5e-5 of the values against the exact result rounded once, far inside the 2e-3 relative tolerance of fp16 storage, for 23 %
 * fewer instructions.  TDK_RCD_EXACT asks for the exact flavour rounded once instead (the tile kernel always is). */
#define TDK_RCD_EXACT 4u
int tdk_rcd_ex(const void* bayer, void* rgb, void* workspace, int width, int height, uint32_t pattern, int dtype, unsigned flags,
               tdk_stream_t stream);

/* decode12_float -> apply_white_balance -> RCD.process as ONE call -- the head of the reference pipeline
 * (torch_darktable/pipeline/image_processor.py:190-255: load_bytes, debayer) -- bit for bit the result of the three
 * calls.  Inside: one streaming kernel decodes the packed bytes (csrc/packed.cu:8-31, scaled by 1/4095) and applies the
 * per-colour gain with a clamp to [0, 1] (csrc/white_balance.cu:10-42) into the workspace plane, then the RCD tile
 * kernel runs on it -- one mosaic plane of traffic instead of two, two launches instead of three.
 * packed: width * height * 3 / 2 bytes; gains: 3 device floats (R, G, B) or NULL (no white balance); workspace:
 * tdk_decode12_wb_rcd_workspace_bytes (the fp32 mosaic); out_dtype: storage type of rgb (H, W, 3).  width must be even. */
size_t tdk_decode12_wb_rcd_workspace_bytes(int width, int height);
int tdk_decode12_wb_rcd(const uint8_t* packed, void* rgb, void* workspace, const float* gains, int width, int height, uint32_t pattern,
                        int ids_format, int out_dtype, tdk_stream_t stream);
/* The same with the per-call flags of tdk_rcd_ex (TDK_RCD_CONCURRENT, TDK_RCD_EXACT, TDK_RCD_TILE_KERNEL). */
int tdk_decode12_wb_rcd_ex(const uint8_t* packed, void* rgb, void* workspace, const float* gains, int width, int height, uint32_t pattern,
                           int ids_format, int out_dtype, unsigned flags, tdk_stream_t stream);

/* PostProcess.process: reference csrc/debayer/postprocess.cu:311-390 (extension.cpp:77-90).
 * in and out must not alias.  The global green ratio is computed and consumed on the
 * device (no host sync). */
size_t tdk_postprocess_workspace_bytes(int width, int height, int color_smoothing_passes, int green_eq_local, int green_eq_global);
int tdk_postprocess(const float* rgb_in, float* rgb_out, void* workspace, int width, int height, uint32_t pattern,
                    int color_smoothing_passes, int green_eq_local, int green_eq_global, float green_eq_threshold,
                    tdk_stream_t stream);
/* The same with the storage type of rgb_in / rgb_out as `dtype` (TDK_F16: an extension; the reference is float32-only).  The images
 * between the stages of a call stay fp32 in the workspace, so a binary16 result is the fp32 result rounded once. */
size_t tdk_postprocess_workspace_bytes_ex(int width, int height, int color_smoothing_passes, int green_eq_local, int green_eq_global, int dtype);
int tdk_postprocess_ex(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, uint32_t pattern, int color_smoothing_passes,
                       int green_eq_local, int green_eq_global, float green_eq_threshold, int dtype, tdk_stream_t stream);

/* apply_white_balance: reference csrc/white_balance.cu:164-183 (extension.cpp:209-210).
 * gains: 3 floats on the DEVICE (no host read-back). out may alias in. */
int tdk_apply_white_balance(const float* bayer_in, float* bayer_out, const float* gains, int width, int height, uint32_t pattern,
                            tdk_stream_t stream);
int tdk_apply_white_balance_ex(const void* bayer_in, void* bayer_out, const float* gains, int width, int height, uint32_t pattern, int dtype,
                               tdk_stream_t stream); /* dtype: storage type of the mosaic (TDK_F16: extension) */

/* estimate_white_balance, sample collection: reference csrc/white_balance.cu:57-126 (collect_samples;
 * extension.cpp:211-212).  One sample per cell of the (height/stride) x (width/stride) grid, n = sh * sw:
 * chroma[n][2] = (r, g) / (r + g + b), intensity[n] = r + g + b, mask[n] = max(2x2 quad) < 1; the
 * skipped last row / column of cells (:69) is written as invalid (the reference leaves it
 * uninitialised).  literal_positions != 0 reads the quad at pos * 2 as the reference does (:71) -- the binding's
 * default --, 0 reads it at pos * stride (the documented intent, an opt-in correction).  The quantile / mean (:149-161) are the
 * binding's device ops, as in the reference.  bayer: (H, W) float32. */
int tdk_wb_collect_samples(const float* bayer, int width, int height, uint32_t pattern, int stride, int literal_positions, float* chroma,
                           float* intensity, uint8_t* mask, tdk_stream_t stream);
int tdk_wb_collect_samples_ex(const void* bayer, int width, int height, uint32_t pattern, int stride, int literal_positions, float* chroma,
                              float* intensity, uint8_t* mask, int dtype, tdk_stream_t stream); /* dtype: storage type of the mosaic; the samples stay fp32 */

/* ---- colour operators: reference csrc/color_conversions.cu (extension.cpp:127-156).
 * (H, W, 3) -> (H, W, 3), npix = H * W. */
enum tdk_color_op {
  TDK_RGB_TO_XYZ = 0, TDK_XYZ_TO_LAB = 1, TDK_LAB_TO_XYZ = 2, TDK_XYZ_TO_RGB = 3, TDK_RGB_TO_LAB = 4, TDK_LAB_TO_RGB = 5,
  TDK_MODIFY_HSL = 6,      /* params = {hue_adjust, sat_adjust, lum_adjust}   color_conversions.cu:143-145 */
  TDK_MODIFY_VIBRANCE = 7, /* params = {amount}                               color_conversions.cu:147-149 */
  TDK_TRANSFORM_3X3 = 8    /* device_matrix = 9 row-major floats ON DEVICE    color_conversions.cu:153-161 */
};
int tdk_color_op(const float* in, float* out, int64_t npix, int op, const float host_params[3], const float* device_matrix,
                 tdk_stream_t stream);
int tdk_color_op_ex(const void* in, void* out, int64_t npix, int op, const float host_params[3], const float* device_matrix, int dtype,
                    tdk_stream_t stream); /* dtype: storage type of in and out (TDK_F16: extension; fp32 arithmetic, rounded once) */

/* compute_luminance / compute_log_luminance: color_conversions.cu:226-233.  rgb (H,W,3) -> lum (H,W) */
int tdk_compute_luminance(const void* rgb, void* lum, int64_t npix, int log_mode, float eps, int rgb_dtype, int lum_dtype,
                          tdk_stream_t stream);
/* modify_luminance / modify_log_luminance: color_conversions.cu:307-314 */
int tdk_modify_luminance(const void* rgb, const void* lum, void* rgb_out, int64_t npix, int log_mode, int rgb_dtype, int lum_dtype,
                         tdk_stream_t stream);

/* normalize_image of the pipeline: reference torch_darktable/pipeline/util.py:8-10.
 * out[i] = (in[i] - bounds[0]) / (bounds[1] - bounds[0]) over `count` samples; bounds[2] on device. */
int tdk_normalize(const void* in, void* out, int64_t count, const float* bounds, int dtype, tdk_stream_t stream);

/* ---- image statistics + tonemaps: reference csrc/tonemap/ (extension.cpp:172-195) */
/* compute_image_bounds: color_adaption.cu:90-120.  bounds[2] on device; call tdk_image_bounds_init
 * once, then tdk_image_bounds_accumulate per image. */
int tdk_image_bounds_init(float* bounds, tdk_stream_t stream);
int tdk_image_bounds_accumulate(const void* rgb, int width, int height, int stride, float* bounds, int dtype, tdk_stream_t stream);
/* The same without the init launch.  state: 4 device words that are {0xffffffff, 0, 0, 0} ("idle") before the first use and are left
 * idle again by the call that finishes a list.  One call per image, stream-ordered; the calls of a list but the last pass
 * bounds = NULL, the last one passes bounds (2 device floats) and total_tickets = the sum of tdk_image_bounds_tickets(width, height,
 * stride) over ALL images of the list: the workgroup that draws that last ticket writes bounds and resets the state.  Results
 * identical to _init + _accumulate (minimum and maximum are exact whatever the order). */
int tdk_image_bounds_tickets(int width, int height, int stride);
int tdk_image_bounds(const void* rgb, int width, int height, int stride, uint32_t* state, float* bounds, unsigned total_tickets, int dtype,
                     tdk_stream_t stream);
/* compute_image_metrics: color_adaption.cu:122-166.  acc: 8 floats of device scratch, zeroed by
 * _init; bounds: 2 device floats ({0,1} unless rescaling); _finish writes metrics[5] on device
 * (normalised by max(valid, 1)) without a host sync. */
int tdk_image_metrics_init(float* acc, tdk_stream_t stream);
int tdk_image_metrics_accumulate(const void* rgb, int width, int height, int stride, float min_gray, const float* bounds, float* acc,
                                 int dtype, tdk_stream_t stream);
int tdk_image_metrics_finish(const float* acc, float* metrics, tdk_stream_t stream);
/* The same statistics with a wide accumulator: TDK_METRICS_ACC_FLOATS device floats (TDK_METRICS_SLOTS rows of 8, 16-byte
 * aligned, ZERO before the first use).  _accumulate_rows adds one image's sums -- one partial per workgroup, spread over
 * the rows, so the grid is not capped by same-address atomics; _finish_reset sums the rows in a fixed order, writes
 * metrics[5] normalised by max(valid, 1) and zeroes the accumulator again (stream-ordered launches).
 * tdk_image_metrics = both for ONE image (or the last image of a list) in ONE launch: the workgroup that draws the last
 * ticket (word 6 of row 0, agent-scope counter) sums the rows and resets them; results identical to the two calls. */
#define TDK_METRICS_SLOTS 1024
#define TDK_METRICS_ACC_FLOATS (TDK_METRICS_SLOTS * 8)
int tdk_image_metrics_accumulate_rows(const void* rgb, int width, int height, int stride, float min_gray, const float* bounds, float* acc_rows,
                                      int dtype, tdk_stream_t stream);
int tdk_image_metrics_finish_reset(float* acc_rows, float* metrics, tdk_stream_t stream);
int tdk_image_metrics(const void* rgb, int width, int height, int stride, float min_gray, const float* bounds, float* acc_rows,
                      float* metrics, int dtype, tdk_stream_t stream);

enum tdk_tonemap { TDK_TONEMAP_REINHARD = 0, TDK_TONEMAP_ACES = 1, TDK_TONEMAP_ACES_ADAPTIVE = 2, TDK_TONEMAP_LINEAR = 3 };
/* reinhard_tonemap / aces_tonemap / adaptive_aces_tonemap / linear_tonemap:
 * reinhard.cu:48-81, aces.cu:92-154, linear.cu:43-76.  rgb (H,W,3) -> u8 (H,W,3).
 * metrics: 5 device floats (ignored by TDK_TONEMAP_ACES, may be NULL there). */
int tdk_tonemap(const void* rgb, uint8_t* out, int64_t npix, int mode, const float* metrics, float gamma, float intensity,
                float light_adapt, float vibrance, int dtype, tdk_stream_t stream);

/* ---- Wiener.process: reference csrc/denoise/denoise.cu:266-345 (extension.cpp:215-223).
 * image (H, W, C), C in {1,3}; sigmas: C device floats; tile_size in {16,32}; overlap in {2,4,8}. */
size_t tdk_wiener_workspace_bytes(int width, int height, int channels, int tile_size, int overlap_factor);
int tdk_wiener(const void* in, void* out, void* workspace, int width, int height, int channels, int tile_size, int overlap_factor,
               const float* sigmas, int dtype, tdk_stream_t stream);

/* Wiener.process_log_luminance as ONE call (reference torch_darktable/denoise.py:54-58 composes
 * compute_log_luminance -> Wiener.process -> modify_log_luminance): rgb (H,W,3) -> rgb (H,W,3).
 * The log-luminance plane lives in the workspace in fp32 and its denoised version never reaches HBM.
 * sigma: one device float. */
size_t tdk_wiener_log_luminance_workspace_bytes(int width, int height, int tile_size, int overlap_factor);
int tdk_wiener_log_luminance(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, int tile_size, int overlap_factor,
                             const float* sigma, float eps, int dtype, tdk_stream_t stream);

/* The same call that ALSO writes the fp32 (H, W) plane compute_luminance(rgb_out) (lum_log_mode 0) or
 * compute_log_luminance(rgb_out, lum_eps) (1) -- bit for bit what those ops return on the stored result -- for a
 * consumer that would extract it next: reference torch_darktable/pipeline/image_processor.py:257-271 runs
 * Wiener.process_log_luminance and then Bilateral.process_rgb, whose first step is that extraction
 * (local_contrast.py:109-114). */
int tdk_wiener_log_luminance_lum(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, int tile_size, int overlap_factor,
                                 const float* sigma, float eps, int dtype, float* lum_out, int lum_log_mode, float lum_eps, tdk_stream_t stream);

/* ---- Bilateral.process: reference csrc/local_contrast/bilateral.cu:358-385 (extension.cpp:111-121).
 * sigma_s <= 4 (the pipeline default 2): one LDS tile kernel; otherwise splat / blur / blur / slice as in bilateral.cu. */
int tdk_bilateral_grid_size(int width, int height, float sigma_s, float sigma_r, int size_xyz[3]);
size_t tdk_bilateral_workspace_bytes(int width, int height, float sigma_s, float sigma_r);
int tdk_bilateral(const void* lum_in, void* lum_out, void* workspace, int width, int height, float sigma_s, float sigma_r,
                  float detail, int dtype, tdk_stream_t stream);

/* Bilateral.process_rgb / process_log_rgb as ONE call (reference torch_darktable/local_contrast.py:109-125:
 * compute_[log_]luminance -> Bilateral.process -> modify_[log_]luminance). */
size_t tdk_bilateral_rgb_workspace_bytes(int width, int height, float sigma_s, float sigma_r);
int tdk_bilateral_rgb(const void* rgb_in, void* rgb_out, void* workspace, int width, int height, float sigma_s, float sigma_r, float detail,
                      int log_mode, float eps, int dtype, tdk_stream_t stream);

/* The workspace of a Bilateral object outlives its calls (the reference keeps its grids in BilateralImpl and drops them when a
 * sigma changes, bilateral.cu:389-390).  What the tile kernel keeps there are its axis tables: cell ranges, sample coordinates
 * and splat weights per tile column / row, a function of (width, height, sigma_s, sigma_r) only.  tdk_bilateral_prepare builds
 * them once, at the start of a workspace of either layout (plane or rgb: same place); a later call on the same workspace,
 * geometry and sigmas passes TDK_BILATERAL_PREPARED and skips the table launch.  Without the flag every call builds them
 * itself (the plain entry points above).  TDK_BILATERAL_GENERAL_PATH takes the four-kernel path where the tile kernel would
 * run (same bits; the GPU tests compare the two through it).  lum_in of the rgb form: the fp32 plane
 * compute_[log_]luminance(rgb_in) if the caller has it (16-byte aligned), or NULL. */
#define TDK_BILATERAL_PREPARED 1u
#define TDK_BILATERAL_GENERAL_PATH 2u
int tdk_bilateral_prepare(void* workspace, int width, int height, float sigma_s, float sigma_r, tdk_stream_t stream);
int tdk_bilateral_ex(const void* lum_in, void* lum_out, void* workspace, int width, int height, float sigma_s, float sigma_r, float detail,
                     int dtype, unsigned flags, tdk_stream_t stream);
int tdk_bilateral_rgb_ex(const void* rgb_in, const float* lum_in, void* rgb_out, void* workspace, int width, int height, float sigma_s,
                         float sigma_r, float detail, int log_mode, float eps, int dtype, unsigned flags, tdk_stream_t stream);

/* Bilateral.process_rgb / process_log_rgb when the producer of rgb_in already has the fp32 (H, W) plane
 * compute_[log_]luminance(rgb_in) (tdk_wiener_log_luminance_lum): the extraction pass is skipped, the result is the same. */
int tdk_bilateral_rgb_lum(const void* rgb_in, const float* lum_in, void* rgb_out, void* workspace, int width, int height, float sigma_s,
                          float sigma_r, float detail, int log_mode, float eps, int dtype, tdk_stream_t stream);

/* ---- Lab hand-over: Wiener.process_log_luminance -> Bilateral.process_rgb as ONE colour round trip.
 * Both stages of the reference convert the RGB pixel to Lab, replace L and convert back (torch_darktable/denoise.py:54-58,
 * local_contrast.py:109-114, csrc/device_conversions.h:213-225); between them only L changes.  These entry points carry the
 * pixel as fp32 lightness plane (H, W) + fp32 chroma plane (H, W, 2) = (a, b) instead of materialising the intermediate RGB
 * image: tdk_wiener_log_luminance_lab leaves compute_luminance(denoised) in lum_out and the denoised pixels' (a, b) in ab_out
 * (the pixels the reference's clip to [0, 1] changes are re-derived from the clipped pixel), tdk_bilateral_lab filters lum_in and
 * writes modify_luminance's result as RGB.  13 + 1 + 6 transcendentals per pixel instead of 9 + 27 + 18; the same result as the
 * two-stage chain up to the rounding of the skipped sRGB encode / decode round trip (tolerance of the colour operators, 2e-5;
 * one binary16 rounding fewer for float16 images).  Helper: tdk_compute_log_luminance_lab = compute_log_luminance(rgb, eps) +
 * the (a, b) of rgb_to_lab(rgb) in one pass.  workspace of the Wiener call: tdk_wiener_log_luminance_workspace_bytes; of the
 * bilateral call: tdk_bilateral_workspace_bytes (tdk_bilateral_prepare + TDK_BILATERAL_PREPARED as above). */
/* bounds (2 device floats or NULL): normalize_image of the reference pipeline (pipeline/util.py:8-10, image_processor.py:257-271),
 * (x - bounds[0]) / (bounds[1] - bounds[0]), applied to rgb as it is read -- the chain then never stores the normalised image. */
int tdk_compute_log_luminance_lab(const void* rgb, float* loglum, float* ab, int64_t npix, float eps, const float* bounds, int rgb_dtype,
                                  tdk_stream_t stream);
int tdk_wiener_log_luminance_lab(const void* rgb_in, void* workspace, int width, int height, int tile_size, int overlap_factor, const float* sigma,
                                 float eps, const float* bounds, int dtype, float* lum_out, float* ab_out, tdk_stream_t stream);
int tdk_bilateral_lab(const float* lum_in, const float* ab_in, void* rgb_out, void* workspace, int width, int height, float sigma_s, float sigma_r,
                      float detail, int out_dtype, unsigned flags, tdk_stream_t stream);

/* ---- Laplacian.process: reference csrc/local_contrast/laplacian.cu:433-480 (extension.cpp:94-108).
 * num_gamma must be 6 (laplacian.cu:625-634). */
size_t tdk_laplacian_workspace_bytes(int width, int height, int num_gamma);
int tdk_laplacian(const float* lum_in, float* lum_out, void* workspace, int width, int height, int num_gamma, float sigma,
                  float shadows, float highlights, float clarity, tdk_stream_t stream);
/* dtype: storage type of lum_in / lum_out.  The reference pads the image to binary16 and writes binary16 values back as float32
 * (laplacian.cu:70-108): a binary16 image in and out carries the SAME values, nothing is rounded twice. */
int tdk_laplacian_ex(const void* lum_in, void* lum_out, void* workspace, int width, int height, int num_gamma, float sigma, float shadows,
                     float highlights, float clarity, int dtype, tdk_stream_t stream);

/* ---- Jpeg.encode: reference csrc/jpeg_encoder.cu:104-180 (extension.cpp:226-246), an nvjpeg wrapper: nvjpegEncodeImage with
 * quality, optimised Huffman tables, sampling factors 444 / 422 / GRAY, baseline or progressive, then
 * nvjpegEncodeRetrieveBitstream (length query, then the bytes).  Here a device encoder (csrc/jpeg.hip) in the same two steps:
 * tdk_jpeg_encode runs colour conversion, FDCT, quantisation, Huffman coding and byte stuffing on the GPU, leaves the JFIF stream
 * in `workspace` (tdk_jpeg_workspace_bytes, 256-byte aligned) and returns its length; tdk_jpeg_retrieve copies it to host memory.
 * image: uint8 on the device, contiguous; input_format 0 = BGR and 1 = RGB planar (3, H, W), 2 = BGRI and 3 = RGBI interleaved
 * (H, W, 3) (JpegInputFormat, csrc/jpeg_encoder.h); subsampling 0 = 4:4:4, 1 = 4:2:2, 2 = gray (JpegSubsampling).
 * tdk_jpeg_encode synchronises `stream` (once per scan for the Huffman statistics, once at the end), like the reference's call.
 * tdk_jpeg_coefficients (tests): the quantised coefficients of the last encode, 64 zig-zag int16 per block, component planes
 * (padded to whole MCUs) back to back. */
size_t tdk_jpeg_workspace_bytes(int width, int height, int subsampling);
int tdk_jpeg_encode(const void* image, int width, int height, int input_format, int quality, int subsampling, int progressive,
                    void* workspace, size_t* length, tdk_stream_t stream);
int tdk_jpeg_retrieve(const void* workspace, int width, int height, int subsampling, uint8_t* out_host, size_t length, tdk_stream_t stream);
int tdk_jpeg_coefficients(const void* workspace, int width, int height, int subsampling, int16_t* out_host, tdk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
