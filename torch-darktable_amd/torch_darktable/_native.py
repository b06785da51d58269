"""ctypes loader for libtdk_hip.so -- the C-ABI kernel library (include/tdk_hip.h).

The library is built in-tree by torch-darktable_amd/build.py.  There is no CPU fallback: if
the library is missing, import of `torch_darktable.torch_darktable_extension` fails, and every
op rejects non-GPU tensors exactly like the reference's TORCH_CHECKs do.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

_LIB_PATH = Path(__file__).resolve().parent / 'libtdk_hip.so'
# Measurement knob (profiles/ab_*.sh, profiles/rcd_ab.py): load another build of the same library (an experiment variant under
# variants/) instead of copying it over the in-tree one.  Never set by the product, the tests or bench.py; announced on stderr.
if os.environ.get('TDK_LIB_PATH'):
  _LIB_PATH = Path(os.environ['TDK_LIB_PATH']).resolve()
  import sys as _sys

  print(f'torch_darktable: loading the kernel library from TDK_LIB_PATH={_LIB_PATH}', file=_sys.stderr)

c_void_p, c_int, c_int64, c_uint32, c_float, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/tdk_hip.h declaration by declaration
SIGNATURES = {
  'tdk_abi_version': (c_int, []),
  'tdk_last_error': (C.c_char_p, []),
  'tdk_profile_enable': (c_int, [c_int]),
  'tdk_profile_filter': (c_int, [C.c_char_p]),
  'tdk_profile_report': (c_int64, [C.c_char_p, c_int64]),
  'tdk_encode12_u16': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
  'tdk_encode12_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
  'tdk_decode12_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
  'tdk_decode12_f16': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
  'tdk_decode12_u16': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
  'tdk_bilinear5x5': (c_int, [c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, c_void_p]),
  'tdk_ppg_workspace_bytes': (c_size_t, [c_int, c_int, c_float]),
  'tdk_ppg': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_float, c_int, c_void_p]),
  'tdk_rcd_workspace_bytes': (c_size_t, [c_int, c_int]),
  'tdk_rcd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, c_void_p]),
  'tdk_rcd_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, C.c_uint, c_void_p]),
  'tdk_decode12_wb_rcd_workspace_bytes': (c_size_t, [c_int, c_int]),
  'tdk_decode12_wb_rcd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, c_int, c_void_p]),
  'tdk_decode12_wb_rcd_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, c_int, C.c_uint, c_void_p]),
  'tdk_postprocess_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
  'tdk_postprocess': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, c_int, c_int, c_float, c_void_p]),
  'tdk_apply_white_balance': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_void_p]),
  'tdk_wb_collect_samples': (c_int, [c_void_p, c_int, c_int, c_uint32, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
  'tdk_color_op': (c_int, [c_void_p, c_void_p, c_int64, c_int, C.POINTER(c_float), c_void_p, c_void_p]),
  'tdk_compute_luminance': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_int, c_void_p]),
  'tdk_modify_luminance': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
  'tdk_normalize': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
  'tdk_image_bounds_init': (c_int, [c_void_p, c_void_p]),
  'tdk_image_bounds_accumulate': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
  'tdk_image_bounds_tickets': (c_int, [c_int, c_int, c_int]),
  'tdk_image_bounds': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, C.c_uint, c_int, c_void_p]),
  'tdk_image_metrics_init': (c_int, [c_void_p, c_void_p]),
  'tdk_image_metrics_accumulate': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
  'tdk_image_metrics_finish': (c_int, [c_void_p, c_void_p, c_void_p]),
  'tdk_image_metrics_accumulate_rows': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
  'tdk_image_metrics_finish_reset': (c_int, [c_void_p, c_void_p, c_void_p]),
  'tdk_image_metrics': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
  'tdk_tonemap': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_float, c_float, c_float, c_float, c_int, c_void_p]),
  'tdk_wiener_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
  'tdk_wiener': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
  'tdk_wiener_log_luminance_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
  'tdk_wiener_log_luminance': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, c_int, c_void_p]),
  'tdk_wiener_log_luminance_lum': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, c_int, c_void_p, c_int, c_float, c_void_p]),
  'tdk_bilateral_rgb_lum': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_float, c_int, c_void_p]),
  'tdk_compute_log_luminance_lab': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_int, c_void_p]),
  'tdk_wiener_log_luminance_lab': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
  'tdk_bilateral_lab': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, C.c_uint, c_void_p]),
  'tdk_bilateral_grid_size': (c_int, [c_int, c_int, c_float, c_float, C.POINTER(c_int)]),
  'tdk_bilateral_workspace_bytes': (c_size_t, [c_int, c_int, c_float, c_float]),
  'tdk_bilateral_prepare': (c_int, [c_void_p, c_int, c_int, c_float, c_float, c_void_p]),
  'tdk_bilateral_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, C.c_uint, c_void_p]),
  'tdk_bilateral_rgb_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_float, c_int, C.c_uint, c_void_p]),
  'tdk_bilateral': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_void_p]),
  'tdk_bilateral_rgb_workspace_bytes': (c_size_t, [c_int, c_int, c_float, c_float]),
  'tdk_bilateral_rgb': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_float, c_int, c_void_p]),
  'tdk_laplacian_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
  'tdk_laplacian': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_void_p]),
  'tdk_postprocess_workspace_bytes_ex': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
  'tdk_postprocess_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, c_int, c_int, c_float, c_int, c_void_p]),
  'tdk_apply_white_balance_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, c_void_p]),
  'tdk_wb_collect_samples_ex': (c_int, [c_void_p, c_int, c_int, c_uint32, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
  'tdk_color_op_ex': (c_int, [c_void_p, c_void_p, c_int64, c_int, C.POINTER(c_float), c_void_p, c_int, c_void_p]),
  'tdk_laplacian_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, c_void_p]),
  'tdk_jpeg_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
  'tdk_jpeg_encode': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, C.POINTER(c_size_t), c_void_p]),
  'tdk_jpeg_retrieve': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
  'tdk_jpeg_coefficients': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
}

TDK_F32, TDK_F16 = 0, 1


def load() -> C.CDLL:
  if not _LIB_PATH.exists():
    raise ImportError(
      f'{_LIB_PATH} is missing: build the HIP kernel library first (python torch-darktable_amd/build.py). '
      'torch_darktable has no CPU or pure-PyTorch fallback.'
    )
  lib = C.CDLL(str(_LIB_PATH))
  for name, (restype, argtypes) in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError here == ABI mismatch between header and library
    fn.restype = restype
    fn.argtypes = argtypes
  if lib.tdk_abi_version() != 4:
    raise ImportError(f'libtdk_hip.so ABI version {lib.tdk_abi_version()} != 4')
  return lib


lib = load()


def check(status: int) -> None:
  """Map a tdk_status to the reference's error type (TORCH_CHECK -> RuntimeError)."""
  if status != 0:
    raise RuntimeError(lib.tdk_last_error().decode('utf-8', 'replace'))


def profile_enable(on: bool, only: str | None = None) -> None:
  """Switch the library's per-kernel event timer on (clearing old records) or off.  `only`
  restricts it to kernels whose name contains that string (None: every launch)."""
  check(lib.tdk_profile_filter(only.encode() if only else None))
  check(lib.tdk_profile_enable(int(on)))


def profile_report() -> dict:
  """{kernel name: (launches, total device ms)} for everything launched since profile_enable(True)."""
  need = lib.tdk_profile_report(None, 0)
  buf = C.create_string_buffer(int(need) + 16)
  lib.tdk_profile_report(buf, len(buf))
  out = {}
  for line in buf.value.decode().splitlines():
    name, count, ms = line.rsplit(' ', 2)
    out[name] = (int(count), float(ms))
  return out
