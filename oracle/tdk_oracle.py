"""numpy front-end of the CPU oracle (oracle/libtdk_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under torch-darktable_amd/ may import this module.

Every function takes/returns contiguous numpy arrays in the reference's layouts
(HWC float32 images, (H, W) planes, flat 1-D codec buffers).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / 'libtdk_oracle.so'

RGGB, BGGR, GRBG, GBRG = 0x94949494, 0x16161616, 0x61616161, 0x49494949
PATTERNS = {'RGGB': RGGB, 'BGGR': BGGR, 'GRBG': GRBG, 'GBRG': GBRG}

COLOR_OPS = {
  'rgb_to_xyz': 0, 'xyz_to_lab': 1, 'lab_to_xyz': 2, 'xyz_to_rgb': 3, 'rgb_to_lab': 4, 'lab_to_rgb': 5,
  'modify_hsl': 6, 'modify_vibrance': 7, 'color_transform_3x3': 8,
}
TONEMAPS = {'reinhard': 0, 'aces': 1, 'adaptive_aces': 2, 'linear': 3}


def build(force: bool = False) -> Path:
  """Compile the oracle with gcc (strict fp32).  Safe to call repeatedly."""
  srcs = list((_HERE / 'src').glob('*'))
  stale = (not _LIB_PATH.exists()) or any(s.stat().st_mtime > _LIB_PATH.stat().st_mtime for s in srcs)
  if force or stale:
    subprocess.run(['make', '-C', str(_HERE), '-B' if force else '-s'], check=True, capture_output=True)
  return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
  global _lib
  if _lib is None:
    build()
    _lib = C.CDLL(str(_LIB_PATH))
  return _lib


def _p(a: np.ndarray):
  return a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
  return np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n: int) -> None:
  os.environ['OMP_NUM_THREADS'] = str(n)


# ------------------------------------------------------------------ codec
def encode12_u16(x: np.ndarray, ids: bool = False) -> np.ndarray:
  x = np.ascontiguousarray(x, dtype=np.uint16)
  out = np.empty(x.size // 2 * 3, np.uint8)
  lib().oracle_encode12_u16(_p(x), _p(out), C.c_int64(x.size // 2), C.c_int(ids))
  return out


def encode12_f32(x: np.ndarray, ids: bool = False, scaled: bool = True) -> np.ndarray:
  x = _f32(x)
  out = np.empty(x.size // 2 * 3, np.uint8)
  lib().oracle_encode12_f32(_p(x), _p(out), C.c_int64(x.size // 2), C.c_int(ids), C.c_int(scaled))
  return out


def decode12_f32(b: np.ndarray, ids: bool = False, scaled: bool = True) -> np.ndarray:
  b = np.ascontiguousarray(b, dtype=np.uint8)
  out = np.empty(b.size // 3 * 2, np.float32)
  lib().oracle_decode12_f32(_p(b), _p(out), C.c_int64(b.size // 3), C.c_int(ids), C.c_int(scaled))
  return out


def decode12_f16(b: np.ndarray, ids: bool = False, scaled: bool = True) -> np.ndarray:
  b = np.ascontiguousarray(b, dtype=np.uint8)
  out = np.empty(b.size // 3 * 2, np.uint16)
  lib().oracle_decode12_f16(_p(b), _p(out), C.c_int64(b.size // 3), C.c_int(ids), C.c_int(scaled))
  return out.view(np.float16)


def decode12_u16(b: np.ndarray, ids: bool = False) -> np.ndarray:
  b = np.ascontiguousarray(b, dtype=np.uint8)
  out = np.empty(b.size // 3 * 2, np.uint16)
  lib().oracle_decode12_u16(_p(b), _p(out), C.c_int64(b.size // 3), C.c_int(ids))
  return out


# ------------------------------------------------------------------ demosaic
def _bayer2d(bayer: np.ndarray) -> np.ndarray:
  bayer = _f32(bayer)
  if bayer.ndim == 3:
    assert bayer.shape[2] == 1
    bayer = bayer[:, :, 0]
  return np.ascontiguousarray(bayer)


def bilinear5x5(bayer: np.ndarray, pattern: int) -> np.ndarray:
  b = _bayer2d(bayer)
  h, w = b.shape
  out = np.empty((h, w, 3), np.float32)
  lib().oracle_bilinear5x5(_p(b), _p(out), C.c_int(w), C.c_int(h), C.c_uint32(pattern))
  return out


def ppg(bayer: np.ndarray, pattern: int, median_threshold: float = 0.0) -> np.ndarray:
  b = _bayer2d(bayer)
  h, w = b.shape
  out = np.empty((h, w, 3), np.float32)
  lib().oracle_ppg(_p(b), _p(out), C.c_int(w), C.c_int(h), C.c_uint32(pattern), C.c_float(median_threshold))
  return out


def rcd(bayer: np.ndarray, pattern: int) -> np.ndarray:
  b = _bayer2d(bayer)
  h, w = b.shape
  out = np.empty((h, w, 3), np.float32)
  rc = lib().oracle_rcd(_p(b), _p(out), C.c_int(w), C.c_int(h), C.c_uint32(pattern))
  if rc != 0:
    raise ValueError('oracle_rcd: width must be even')
  return out


def rcd_planes(bayer: np.ndarray, pattern: int):
  """(rgb, PQ_dir plane, p_diff plane, q_diff plane): the three half-density planes as they stand after
  step 4.2, in the reference's flat `idx / 2` slot layout (H*W floats each, only the first H*W/2 slots used)."""
  b = _bayer2d(bayer)
  h, w = b.shape
  out = np.empty((h, w, 3), np.float32)
  pq, pd, qd = (np.empty(h * w, np.float32) for _ in range(3))
  rc = lib().oracle_rcd_planes(_p(b), _p(out), C.c_int(w), C.c_int(h), C.c_uint32(pattern), _p(pq), _p(pd), _p(qd))
  if rc != 0:
    raise ValueError('oracle_rcd: width must be even')
  return out, pq, pd, qd


def border_interpolate(bayer: np.ndarray, pattern: int, border: int = 3) -> np.ndarray:
  b = _bayer2d(bayer)
  h, w = b.shape
  out = np.zeros((h, w, 3), np.float32)
  lib().oracle_border_interpolate(_p(b), _p(out), C.c_int(w), C.c_int(h), C.c_uint32(pattern), C.c_int(border))
  return out


def green_eq_sums(rgb: np.ndarray, pattern: int):
  rgb = _f32(rgb)
  h, w, _ = rgb.shape
  s32 = np.zeros(2, np.float32)
  s64 = np.zeros(2, np.float64)
  lib().oracle_green_eq_sums(_p(rgb), C.c_int(w), C.c_int(h), C.c_uint32(pattern), _p(s32), _p(s64))
  return s32, s64


def postprocess(rgb: np.ndarray, pattern: int, color_smoothing_passes: int = 0, green_eq_local: bool = False,
                green_eq_global: bool = False, green_eq_threshold: float = 0.04, ratio_override: float = -1.0) -> np.ndarray:
  rgb = _f32(rgb)
  h, w, _ = rgb.shape
  out = np.empty_like(rgb)
  lib().oracle_postprocess(_p(rgb), _p(out), C.c_int(w), C.c_int(h), C.c_uint32(pattern), C.c_int(color_smoothing_passes),
                           C.c_int(green_eq_local), C.c_int(green_eq_global), C.c_float(green_eq_threshold),
                           C.c_float(ratio_override))
  return out


def apply_white_balance(bayer: np.ndarray, gains, pattern: int) -> np.ndarray:
  b = _f32(bayer)
  h, w = b.shape
  g = _f32(gains)
  out = np.empty_like(b)
  lib().oracle_apply_white_balance(_p(b), _p(out), C.c_int(w), C.c_int(h), _p(g), C.c_uint32(pattern))
  return out


def wb_collect_samples(bayer: np.ndarray, pattern: int, stride: int = 8, literal_positions: bool = True):
  """(chroma (n, 2), intensity (n,), mask (n,) bool) over the full (H/stride) x (W/stride) cell grid."""
  b = _f32(bayer)
  h, w = b.shape
  n = (h // stride) * (w // stride)
  chroma, inten, mask = np.empty((n, 2), np.float32), np.empty(n, np.float32), np.empty(n, np.uint8)
  lib().oracle_wb_collect_samples(_p(b), C.c_int(w), C.c_int(h), C.c_uint32(pattern), C.c_int(stride), C.c_int(literal_positions), _p(chroma),
                                  _p(inten), _p(mask))
  return chroma, inten, mask.astype(bool)


def estimate_white_balance(bayer_images, pattern: int, quantile: float = 0.95, stride: int = 8, literal_positions: bool = True) -> np.ndarray:
  """reference csrc/white_balance.cu:129-161 on top of wb_collect_samples: keep valid samples, select those with
  intensity >= quantile(intensity, q) (torch.quantile: linear interpolation), mean chroma (mr, mg) ->
  gains (mr / mg, 1, (1 - mr - mg) / mg); (1, 1, 1) when nothing is valid."""
  parts = [wb_collect_samples(b, pattern, stride, literal_positions) for b in bayer_images]
  mask = np.concatenate([p[2] for p in parts])
  chroma = np.concatenate([p[0] for p in parts])[mask]
  inten = np.concatenate([p[1] for p in parts])[mask]
  if chroma.shape[0] == 0:
    return np.ones(3, np.float32)
  thr = np.quantile(inten.astype(np.float64), quantile).astype(np.float32)
  bright = chroma[inten >= thr]
  if bright.shape[0] == 0:
    return np.ones(3, np.float32)
  m = bright.astype(np.float64).mean(0)
  return np.array([m[0] / m[1], 1.0, (1.0 - m[0] - m[1]) / m[1]], np.float32)


# ------------------------------------------------------------------ colour
def color_op(name: str, img: np.ndarray, params=None) -> np.ndarray:
  img = _f32(img)
  out = np.empty_like(img)
  prm = _f32(params if params is not None else [0.0] * 9).ravel()
  if prm.size < 9:
    prm = np.concatenate([prm, np.zeros(9 - prm.size, np.float32)])
  lib().oracle_color_op(_p(img), _p(out), C.c_int64(img.size // 3), C.c_int(COLOR_OPS[name]), _p(prm))
  return out


def compute_luminance(img: np.ndarray, log: bool = False, eps: float = 1e-6) -> np.ndarray:
  img = _f32(img)
  out = np.empty(img.shape[:2], np.float32)
  lib().oracle_compute_luminance(_p(img), _p(out), C.c_int64(out.size), C.c_int(log), C.c_float(eps))
  return out


def modify_luminance(img: np.ndarray, lum: np.ndarray, log: bool = False) -> np.ndarray:
  img, lum = _f32(img), _f32(lum)
  out = np.empty_like(img)
  lib().oracle_modify_luminance(_p(img), _p(lum), _p(out), C.c_int64(lum.size), C.c_int(log))
  return out


def tonemap(name: str, img: np.ndarray, metrics=None, gamma=1.0, intensity=0.0, light_adapt=0.8, vibrance=0.0,
            return_float: bool = False):
  img = _f32(img)
  m = _f32(metrics if metrics is not None else np.zeros(5))
  u8 = np.empty(img.shape, np.uint8)
  f32 = np.empty(img.shape, np.float32) if return_float else None
  lib().oracle_tonemap(_p(img), _p(u8), _p(f32) if return_float else None, C.c_int64(img.size // 3), C.c_int(TONEMAPS[name]),
                       _p(m), C.c_float(gamma), C.c_float(intensity), C.c_float(light_adapt), C.c_float(vibrance))
  return (u8, f32) if return_float else u8


def image_bounds(images, stride: int = 8) -> np.ndarray:
  b = np.array([np.finfo(np.float32).max, -np.finfo(np.float32).max], np.float32)
  for img in images:
    img = _f32(img)
    h, w, _ = img.shape
    lib().oracle_image_bounds(_p(img), C.c_int(w), C.c_int(h), C.c_int(stride), _p(b))
  return b


def image_metrics(images, stride: int = 8, min_gray: float = 1e-4, rescale: bool = False) -> np.ndarray:
  bounds = image_bounds(images, stride) if rescale else np.array([0.0, 1.0], np.float32)
  acc = np.zeros(6, np.float64)
  for img in images:
    img = _f32(img)
    h, w, _ = img.shape
    lib().oracle_image_metrics_accumulate(_p(img), C.c_int(w), C.c_int(h), C.c_int(stride), C.c_float(min_gray), _p(bounds), _p(acc))
  out = np.zeros(5, np.float32)
  lib().oracle_image_metrics_finish(_p(acc), _p(out))
  return out


# ------------------------------------------------------------------ denoise / local contrast
def wiener_window(K: int, weight: float = 0.3) -> np.ndarray:
  w = np.zeros(K, np.float32)
  lib().oracle_wiener_window(C.c_int(K), C.c_float(weight), _p(w))
  return w


def wiener(img: np.ndarray, sigmas, tile_size: int = 32, overlap_factor: int = 4) -> np.ndarray:
  img = _f32(img)
  h, w, c = img.shape
  s = _f32(np.broadcast_to(np.asarray(sigmas, np.float32), (c,)))
  out = np.empty_like(img)
  rc = lib().oracle_wiener(_p(img), _p(out), C.c_int(w), C.c_int(h), C.c_int(c), C.c_int(tile_size), C.c_int(overlap_factor), _p(s))
  if rc != 0:
    raise ValueError(f'oracle_wiener: invalid parameters (rc={rc})')
  return out


def bilateral_grid_size(width: int, height: int, sigma_s: float, sigma_r: float):
  sz = (C.c_int * 3)()
  lib().oracle_bilateral_grid_size(C.c_int(width), C.c_int(height), C.c_float(sigma_s), C.c_float(sigma_r), sz)
  return tuple(sz)


def bilateral(lum: np.ndarray, sigma_s: float, sigma_r: float, detail: float) -> np.ndarray:
  lum = _f32(lum)
  h, w = lum.shape
  out = np.empty_like(lum)
  lib().oracle_bilateral(_p(lum), _p(out), C.c_int(w), C.c_int(h), C.c_float(sigma_s), C.c_float(sigma_r), C.c_float(detail))
  return out


def laplacian(lum: np.ndarray, sigma=0.2, shadows=1.0, highlights=1.0, clarity=0.0) -> np.ndarray:
  lum = _f32(lum)
  h, w = lum.shape
  out = np.empty_like(lum)
  rc = lib().oracle_laplacian(_p(lum), _p(out), C.c_int(w), C.c_int(h), C.c_float(sigma), C.c_float(shadows),
                              C.c_float(highlights), C.c_float(clarity))
  if rc != 0:
    raise ValueError('oracle_laplacian: image too small')
  return out


# ------------------------------------------------------------------ pure-numpy helpers
def cfa_color(row, col, pattern: int):
  """0=R 1=G 2=B; reference csrc/debayer/bayer_device.h:9-11."""
  return (pattern >> ((((row << 1) & 14) + (col & 1)) << 1)) & 3


def mosaic(rgb: np.ndarray, pattern: int = RGGB) -> np.ndarray:
  """Sample an (H, W, 3) image on the CFA that fc()/cfa_color defines for `pattern`.

  Unlike rgb_to_bayer below this is consistent with the demosaic kernels for all four
  patterns (the reference's rgb_to_bayer channel table is wrong for GRBG / GBRG)."""
  h, w, _ = rgb.shape
  rows, cols = np.mgrid[0:h, 0:w]
  ch = cfa_color(rows, cols, pattern)
  return np.take_along_axis(rgb, ch[:, :, None], axis=2).astype(rgb.dtype)


# reference torch_darktable/bayer.py
def rgb_to_bayer(rgb: np.ndarray, pattern: int = RGGB) -> np.ndarray:
  """Mosaic an (H, W, 3) image: reference torch_darktable/bayer.py:24-47,84-120."""
  ch = {RGGB: (0, 1, 1, 2), BGGR: (2, 1, 1, 0), GRBG: (1, 0, 1, 2), GBRG: (1, 2, 1, 0)}[pattern]
  h, w, _ = rgb.shape
  out = np.zeros((h // 2 * 2, w // 2 * 2), rgb.dtype)
  out[0::2, 0::2] = rgb[0:h // 2 * 2:2, 0:w // 2 * 2:2, ch[0]]
  out[0::2, 1::2] = rgb[0:h // 2 * 2:2, 1:w // 2 * 2:2, ch[1]]
  out[1::2, 0::2] = rgb[1:h // 2 * 2:2, 0:w // 2 * 2:2, ch[2]]
  out[1::2, 1::2] = rgb[1:h // 2 * 2:2, 1:w // 2 * 2:2, ch[3]]
  return out[:, :, None]


# ------------------------------------------------------------------ JPEG (the format after the path; oracle/src/jpeg.c)
def jpeg_encode(img: np.ndarray, quality: int = 94, input_format: int = 3, subsampling: int = 1, progressive: bool = False,
                return_coefs: bool = False):
  """uint8 image, (H, W, 3) for the interleaved formats (2 = BGRI, 3 = RGBI) or (3, H, W) for the planar ones (0 = BGR,
  1 = RGB) -> the JPEG byte stream as a uint8 array (and the quantised zig-zag coefficients, component planes back to back)."""
  img = np.ascontiguousarray(img, dtype=np.uint8)
  h, w = (img.shape[1], img.shape[2]) if input_format < 2 else (img.shape[0], img.shape[1])
  hs = 2 if subsampling == 1 else 1
  nmx, nmy = -(-w // (8 * hs)), -(-h // 8)
  nblocks = nmx * nmy * (1 if subsampling == 2 else hs + 2)
  cap = 1024 + nblocks * 420
  out = np.empty(cap, np.uint8)
  coefs = np.empty(nblocks * 64, np.int16) if return_coefs else None
  f = lib().oracle_jpeg_encode
  f.restype = C.c_int64
  n = f(_p(img), C.c_int(w), C.c_int(h), C.c_int(input_format), C.c_int(quality), C.c_int(subsampling), C.c_int(bool(progressive)),
        _p(out), C.c_int64(cap), _p(coefs) if return_coefs else None)
  assert 0 < n <= cap
  return (out[:n].copy(), coefs) if return_coefs else out[:n].copy()
