"""`torch_darktable.torch_darktable_extension` for MI355X -- the op surface of the reference's
pybind11 module (reference csrc/extension.cpp:50-248, torch_darktable_extension.pyi) bound to
the hand-written HIP kernels in libtdk_hip.so (include/tdk_hip.h).

Same names, argument order, defaults, tensor conventions (HWC float32, `(width, height)` in
constructors) and error types (RuntimeError where the reference TORCH_CHECKs).  Differences,
all deliberate:
  * every op runs on the tensor's own device under a device guard and on PyTorch's CURRENT
    stream (the reference has no guard and uses the legacy default stream for colour, codec,
    white-balance and statistics kernels);
  * nothing synchronises with the host: green-equilibration ratio, metrics normalisation and
    white-balance gains stay on the device (reference: `.item()` at postprocess.cu:364,
    color_adaption.cu:162, white_balance.cu:174);
  * float16 image storage is accepted where noted (extension; the reference is float32-only);
  * `RCD.process` returns a fresh tensor (the reference returns its persistent workspace
    buffer, rcd.cu:670) and is a pure function of its input;
  * there is no CPU path: non-GPU tensors are rejected, as in the reference.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import enum
import threading
from typing import Sequence

import torch

from . import _native
from ._native import TDK_F16, TDK_F32, check, lib

__all__ = [
  'BayerPattern', 'PPG', 'RCD', 'PostProcess', 'Laplacian', 'Bilateral', 'Wiener', 'TonemapParams',
  'encode12_u16', 'encode12_float', 'decode12_float', 'decode12_half', 'decode12_u16',
  'compute_luminance', 'modify_luminance', 'compute_log_luminance', 'modify_log_luminance', 'modify_hsl', 'modify_vibrance',
  'rgb_to_xyz', 'xyz_to_lab', 'lab_to_xyz', 'xyz_to_rgb', 'rgb_to_lab', 'lab_to_rgb', 'color_transform_3x3',
  'compute_image_bounds', 'compute_image_metrics', 'MetricsAccumulator', 'reinhard_tonemap', 'aces_tonemap', 'adaptive_aces_tonemap', 'linear_tonemap',
  'bilinear5x5_demosaic', 'apply_white_balance', 'estimate_white_balance', 'create_wiener',
  'Jpeg', 'JpegException', 'JpegInputFormat', 'JpegSubsampling',
]


class BayerPattern(enum.IntEnum):
  """reference csrc/debayer/demosaic.h:7-12"""

  RGGB = 0x94949494
  BGGR = 0x16161616
  GRBG = 0x61616161
  GBRG = 0x49494949


# ------------------------------------------------------------------ verification paths
# Two ops have a second kernel path that gives the same bits (RCD: 64 x 64 LDS tiles instead of the column strips; Bilateral:
# the four-kernel path instead of the LDS tile kernel).  The library selects per call (the `flags` of tdk_rcd_ex /
# tdk_bilateral_ex); this thread-local context is how the GPU tests ask for the other path.  Nothing is process-global.
TDK_RCD_TILE_KERNEL, TDK_RCD_CONCURRENT, TDK_RCD_EXACT, TDK_BILATERAL_PREPARED, TDK_BILATERAL_GENERAL_PATH = 1, 2, 4, 1, 2
_verify = threading.local()


@contextlib.contextmanager
def verification_paths(rcd_tiles: bool = False, bilateral_general: bool = False, rcd_exact: bool = False):
  """Inside the context (this thread only) RCD.process takes the tile kernel and / or Bilateral the four-kernel path.
  rcd_exact: a float16 result of RCD.process is the exact flavour's result rounded once (TDK_RCD_EXACT) instead of the default
  approximate arithmetic of the column strips (include/tdk_hip.h); float32 results are always exact."""
  old = (getattr(_verify, 'rcd', 0), getattr(_verify, 'bil', 0))
  _verify.rcd = (TDK_RCD_TILE_KERNEL if rcd_tiles else 0) | (TDK_RCD_EXACT if rcd_exact else 0)
  _verify.bil = TDK_BILATERAL_GENERAL_PATH if bilateral_general else 0
  try:
    yield
  finally:
    _verify.rcd, _verify.bil = old


@contextlib.contextmanager
def concurrent_frames(on: bool = True):
  """Inside the context (this thread only) the ops are told that other frames' kernels are in flight on other streams
  (TDK_RCD_CONCURRENT: RCD.process takes the strips variant that runs on half the waves per CU -- the same bits, slower alone,
  faster for the whole when the other frames' kernels can use what it leaves).  sharding.FrameStreams enters it for the frames
  it spreads over more than one stream."""
  old = getattr(_verify, 'concurrent', 0)
  _verify.concurrent = TDK_RCD_CONCURRENT if on else 0
  try:
    yield
  finally:
    _verify.concurrent = old


# ------------------------------------------------------------------ helpers
def _stream() -> C.c_void_p:
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
  return C.c_void_p(t.data_ptr() if t is not None else 0)


def _require(cond: bool, msg: str) -> None:
  if not cond:
    raise RuntimeError(msg)


def _dtype_tag(t: torch.Tensor, what: str = 'Input') -> int:
  if t.dtype == torch.float32:
    return TDK_F32
  if t.dtype == torch.float16:
    return TDK_F16
  raise RuntimeError(f'{what} tensor must be float32 (or float16 storage)')


def _pattern(p) -> int:
  if isinstance(p, enum.Enum) and not isinstance(p, BayerPattern):
    p = p.value  # torch_darktable.bayer.BayerPattern wraps the extension enum
  return int(BayerPattern(int(p)))


def _workspace(nbytes: int, device: torch.device) -> torch.Tensor | None:
  return torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes > 0 else None


def _check_rgb(image: torch.Tensor, name: str = 'image', allow_half: bool = False) -> None:
  _require(image.is_cuda, f'{name} must be CUDA')
  ok = image.dtype == torch.float32 or (allow_half and image.dtype == torch.float16)
  _require(ok, f'{name} must be float32')
  _require(image.dim() == 3 and image.size(2) == 3, f'{name} must be (H,W,3)')


# ------------------------------------------------------------------ 12-bit codec (csrc/packed.cu:158-280)
def _check_flat(t: torch.Tensor, dtype: torch.dtype, multiple: int, what: str) -> torch.Tensor:
  _require(t.is_cuda, 'Input must be on CUDA device')
  _require(t.dtype == dtype, f'Input must be {what}')
  _require(t.dim() == 1, 'Input must be 1D tensor')
  _require(t.size(0) % multiple == 0, 'Input length must be even' if multiple == 2 else 'Input length must be multiple of 3')
  return t.contiguous()


def encode12_u16(input: torch.Tensor, ids_format: bool = False) -> torch.Tensor:
  x = _check_flat(input, torch.uint16, 2, 'uint16')
  n = x.size(0) // 2
  out = torch.empty(n * 3, dtype=torch.uint8, device=x.device)
  with torch.cuda.device(x.device):
    check(lib.tdk_encode12_u16(_ptr(x), _ptr(out), n, int(ids_format), _stream()))
  return out


def encode12_float(input: torch.Tensor, ids_format: bool = False, scaled: bool = True) -> torch.Tensor:
  x = _check_flat(input, torch.float32, 2, 'float32')
  n = x.size(0) // 2
  out = torch.empty(n * 3, dtype=torch.uint8, device=x.device)
  with torch.cuda.device(x.device):
    check(lib.tdk_encode12_f32(_ptr(x), _ptr(out), n, int(ids_format), int(scaled), _stream()))
  return out


def decode12_float(input: torch.Tensor, ids_format: bool = False, scaled: bool = True) -> torch.Tensor:
  x = _check_flat(input, torch.uint8, 3, 'uint8')
  n = x.size(0) // 3
  out = torch.empty(n * 2, dtype=torch.float32, device=x.device)
  with torch.cuda.device(x.device):
    check(lib.tdk_decode12_f32(_ptr(x), _ptr(out), n, int(ids_format), int(scaled), _stream()))
  return out


def decode12_half(input: torch.Tensor, ids_format: bool = False, scaled: bool = True) -> torch.Tensor:
  x = _check_flat(input, torch.uint8, 3, 'uint8')
  n = x.size(0) // 3
  out = torch.empty(n * 2, dtype=torch.float16, device=x.device)
  with torch.cuda.device(x.device):
    check(lib.tdk_decode12_f16(_ptr(x), _ptr(out), n, int(ids_format), int(scaled), _stream()))
  return out


def decode12_u16(input: torch.Tensor, ids_format: bool = False) -> torch.Tensor:
  x = _check_flat(input, torch.uint8, 3, 'uint8')
  n = x.size(0) // 3
  out = torch.empty(n * 2, dtype=torch.uint16, device=x.device)
  with torch.cuda.device(x.device):
    check(lib.tdk_decode12_u16(_ptr(x), _ptr(out), n, int(ids_format), _stream()))
  return out


# ------------------------------------------------------------------ demosaic
def _check_bayer(input: torch.Tensor) -> torch.Tensor:
  _require(input.is_cuda, 'Input tensor must be on CUDA device')
  _require(input.dtype in (torch.float32, torch.float16), 'Input tensor must be float32')
  _require(input.dim() == 3, 'Input tensor must be 3D (H, W, 1)')
  _require(input.size(2) == 1, 'Input must have single channel (raw Bayer)')
  return input.contiguous()


def bilinear5x5_demosaic(input: torch.Tensor, pattern) -> torch.Tensor:
  """reference csrc/debayer/bilinear.cu:104-148"""
  x = _check_bayer(input)
  h, w = x.size(0), x.size(1)
  out = torch.empty((h, w, 3), dtype=x.dtype, device=x.device)
  with torch.cuda.device(x.device):
    check(lib.tdk_bilinear5x5(_ptr(x), _ptr(out), w, h, _pattern(pattern), _dtype_tag(x), _stream()))
  return out


class _Workspace:
  """Common (device, width, height) bookkeeping of the reference's *Impl structs."""

  def __init__(self, device: torch.device, width: int, height: int):
    device = torch.device(device)
    _require(device.type == 'cuda', f'torch_darktable ops need a GPU device, got {device}')
    if device.index is None:
      device = torch.device('cuda', torch.cuda.current_device())
    self._device = device
    self._width = int(width)
    self._height = int(height)

  @property
  def width(self) -> int:
    return self._width

  @property
  def height(self) -> int:
    return self._height

  def _check_size(self, input: torch.Tensor) -> None:
    _require(input.size(0) == self._height and input.size(1) == self._width, 'Input dimensions must match workspace size')

  def _workspace(self, nbytes: int, device: torch.device) -> torch.Tensor:
    """Cached scratch of at least `nbytes`, one buffer per CUDA stream: a workspace object may be used from
    several streams (or threads with different current streams) without its kernels sharing slabs / grids."""
    cache = self.__dict__.setdefault('_scratch', {})
    key = torch.cuda.current_stream(device).cuda_stream
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
      buf = cache[key] = _workspace(max(int(nbytes), 256), device)
    return buf

  def _drop_workspaces(self) -> None:
    self.__dict__.pop('_scratch', None)


class PPG(_Workspace):
  """reference csrc/debayer/ppg.cu:391-476 (extension.cpp:57-65)"""

  def __init__(self, device, width: int, height: int, pattern, median_threshold: float = 0.0):
    super().__init__(device, width, height)
    self._pattern = _pattern(pattern)
    self.median_threshold = float(median_threshold)

  def process(self, input: torch.Tensor) -> torch.Tensor:
    x = _check_bayer(input)
    self._check_size(x)
    out = torch.empty((self._height, self._width, 3), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
      ws = _workspace(lib.tdk_ppg_workspace_bytes(self._width, self._height, self.median_threshold), x.device)
      check(lib.tdk_ppg(_ptr(x), _ptr(out), _ptr(ws), self._width, self._height, self._pattern, self.median_threshold, _dtype_tag(x), _stream()))
    return out


class RCD(_Workspace):
  """reference csrc/debayer/rcd.cu:565-681 (extension.cpp:67-74)"""

  def __init__(self, device, width: int, height: int, pattern):
    super().__init__(device, width, height)
    self._pattern = _pattern(pattern)

  def process(self, input: torch.Tensor) -> torch.Tensor:
    x = _check_bayer(input)
    self._check_size(x)
    out = torch.empty((self._height, self._width, 3), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
      check(lib.tdk_rcd_ex(_ptr(x), _ptr(out), None, self._width, self._height, self._pattern, _dtype_tag(x), getattr(_verify, 'rcd', 0) | getattr(_verify, 'concurrent', 0), _stream()))
    return out


  def process_packed12(self, packed: torch.Tensor, gains: torch.Tensor | None = None, ids_format: bool = False,
                       out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """decode12_float -> apply_white_balance -> process as one library call (the head of the reference pipeline,
    torch_darktable/pipeline/image_processor.py:190-255), bit for bit the result of the three calls.
    packed: flat uint8 tensor of width * height * 3 / 2 bytes; gains: 3 floats (R, G, B) or None."""
    _require(packed.is_cuda and packed.dtype == torch.uint8 and packed.dim() == 1, 'packed must be a 1-D uint8 CUDA tensor')
    _require(packed.numel() == self._width * self._height * 3 // 2, 'packed size does not match the workspace image size')
    _require(out_dtype in (torch.float32, torch.float16), 'out_dtype must be float32 or float16')
    x = packed.contiguous()
    g = gains.to(device=x.device, dtype=torch.float32).contiguous() if gains is not None else None
    if g is not None:
      _require(g.numel() == 3, 'gains must have 3 elements')
    out = torch.empty((self._height, self._width, 3), dtype=out_dtype, device=x.device)
    with torch.cuda.device(x.device):
      ws = self._workspace(lib.tdk_decode12_wb_rcd_workspace_bytes(self._width, self._height), x.device)
      check(lib.tdk_decode12_wb_rcd_ex(_ptr(x), _ptr(out), _ptr(ws), _ptr(g), self._width, self._height, self._pattern, int(ids_format), _dtype_tag(out),
                                       getattr(_verify, 'rcd', 0) | getattr(_verify, 'concurrent', 0), _stream()))
    return out


class PostProcess(_Workspace):
  """reference csrc/debayer/postprocess.cu:264-416 (extension.cpp:77-90)"""

  def __init__(self, device, width: int, height: int, pattern, color_smoothing_passes: int = 0, green_eq_local: bool = False,
               green_eq_global: bool = False, green_eq_threshold: float = 0.04):
    super().__init__(device, width, height)
    self._pattern = _pattern(pattern)
    self.color_smoothing_passes = int(color_smoothing_passes)
    self.green_eq_local = bool(green_eq_local)
    self.green_eq_global = bool(green_eq_global)
    self.green_eq_threshold = float(green_eq_threshold)

  def process(self, input: torch.Tensor) -> torch.Tensor:
    _require(input.is_cuda, 'Input tensor must be on CUDA device')
    tag = _dtype_tag(input)  # float32, or float16 storage (extension): the fp32 result rounded once
    _require(input.dim() == 3, 'Input tensor must be 3D (H, W, 3)')
    _require(input.size(2) == 3, 'Input must have 3 channels (RGB)')
    _require(input.size(0) == self._height and input.size(1) == self._width,
             f'Input size {input.size(0)}x{input.size(1)} does not match expected {self._height}x{self._width}')
    x = input.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
      nbytes = lib.tdk_postprocess_workspace_bytes_ex(self._width, self._height, self.color_smoothing_passes, int(self.green_eq_local),
                                                      int(self.green_eq_global), tag)
      ws = _workspace(nbytes, x.device)
      check(lib.tdk_postprocess_ex(_ptr(x), _ptr(out), _ptr(ws), self._width, self._height, self._pattern, self.color_smoothing_passes,
                                   int(self.green_eq_local), int(self.green_eq_global), self.green_eq_threshold, tag, _stream()))
    return out


# ------------------------------------------------------------------ white balance (csrc/white_balance.cu)
def apply_white_balance(bayer_image: torch.Tensor, gains: torch.Tensor, pattern) -> torch.Tensor:
  _require(bayer_image.is_cuda and bayer_image.dtype in (torch.float32, torch.float16) and bayer_image.dim() == 2,
           'bayer_image must be a float32 CUDA (H, W) tensor')  # float16 storage: extension
  x = bayer_image.contiguous()
  g = gains.to(device=x.device, dtype=torch.float32).contiguous()
  _require(g.numel() == 3, 'gains must have 3 elements')
  out = torch.empty_like(x)
  with torch.cuda.device(x.device):
    check(lib.tdk_apply_white_balance_ex(_ptr(x), _ptr(out), _ptr(g), x.size(1), x.size(0), _pattern(pattern), _dtype_tag(x), _stream()))
  return out


def estimate_white_balance(bayer_images: Sequence[torch.Tensor], pattern, quantile: float = 0.95, stride: int = 8,
                           literal_positions: bool = True) -> torch.Tensor:
  """reference csrc/white_balance.cu:57-161.  Sample collection is one HIP kernel per image
  (tdk_wb_collect_samples); masking, the intensity quantile and the chroma mean are torch ops on
  the device, as in the reference (white_balance.cu:119-161) -- no host synchronisation except the
  shape-dependent boolean indexing the reference has too.

  Default = the reference's read pattern: the quad of grid cell `pos` is read at `pos * 2` although the grid is sized by
  `stride` (white_balance.cu:71), so the estimate only looks at the top-left (2 / stride)^2 of the frame -- deterministic
  reference behaviour, reproduced so that a drop-in caller gets the reference's gains.  `literal_positions=False` is the
  opt-in correction (quads at `pos * stride`, the whole frame).  The one deliberate deviation in both modes: the reference
  leaves the skipped last row / column of cells uninitialised (:69, 107-109: its result depends on stale memory); here every
  entry is defined (skipped cells are invalid)."""
  if len(bayer_images) == 0:
    raise RuntimeError('No images provided')
  pat = _pattern(pattern)
  stride = int(stride)
  first = bayer_images[0]
  _require(first.is_cuda and first.dim() == 2, 'bayer images must be CUDA (H, W) tensors')
  dev = first.device
  h, w = first.shape
  _require(stride >= 2 and h >= stride and w >= stride, 'stride must be >= 2 and not larger than the image')
  per = (h // stride) * (w // stride)
  n = per * len(bayer_images)
  chroma = torch.empty((n, 2), dtype=torch.float32, device=dev)
  inten = torch.empty(n, dtype=torch.float32, device=dev)
  mask = torch.empty(n, dtype=torch.bool, device=dev)
  with torch.cuda.device(dev):
    for i, img in enumerate(bayer_images):
      _require(img.is_cuda and img.dim() == 2 and img.dtype in (torch.float32, torch.float16), 'bayer images must be float32 CUDA (H, W) tensors')
      _require(tuple(img.shape) == (h, w) and img.device == dev, 'all bayer images must have the same size and device')
      x = img.contiguous()
      check(lib.tdk_wb_collect_samples_ex(_ptr(x), w, h, pat, stride, int(literal_positions), _ptr(chroma[i * per:]), _ptr(inten[i * per:]),
                                          _ptr(mask[i * per:]), _dtype_tag(x), _stream()))
  chroma_all, inten_all = chroma[mask], inten[mask]
  if chroma_all.size(0) == 0:
    return torch.tensor([1.0, 1.0, 1.0], device=dev)
  bright = chroma_all[inten_all >= torch.quantile(inten_all, quantile)]
  if bright.size(0) == 0:
    return torch.tensor([1.0, 1.0, 1.0], device=dev)
  m = bright.mean(0)
  return torch.stack((m[0] / m[1], torch.tensor(1.0, device=dev), (1.0 - m[0] - m[1]) / m[1]))


# ------------------------------------------------------------------ colour (csrc/color_conversions.cu)
_COLOR_OPS = {'rgb_to_xyz': 0, 'xyz_to_lab': 1, 'lab_to_xyz': 2, 'xyz_to_rgb': 3, 'rgb_to_lab': 4, 'lab_to_rgb': 5,
              'modify_hsl': 6, 'modify_vibrance': 7, 'color_transform_3x3': 8}


def _color_op(name: str, input: torch.Tensor, params=(0.0, 0.0, 0.0), matrix: torch.Tensor | None = None) -> torch.Tensor:
  _require(input.dtype in (torch.float32, torch.float16), 'Input must be float32')  # float16 storage: extension (fp32 arithmetic, rounded once)
  _require(input.dim() == 3 and input.size(2) == 3, 'Input must be (H, W, 3)')
  _require(input.is_cuda, 'Input must be on CUDA device')
  _require(input.is_contiguous(), 'Input tensor must be contiguous')
  out = torch.empty_like(input)
  prm = (C.c_float * 3)(*[float(p) for p in params])
  with torch.cuda.device(input.device):
    check(lib.tdk_color_op_ex(_ptr(input), _ptr(out), input.size(0) * input.size(1), _COLOR_OPS[name], prm, _ptr(matrix), _dtype_tag(input), _stream()))
  return out


def rgb_to_xyz(rgb: torch.Tensor) -> torch.Tensor:
  return _color_op('rgb_to_xyz', rgb)


def xyz_to_lab(xyz: torch.Tensor) -> torch.Tensor:
  return _color_op('xyz_to_lab', xyz)


def lab_to_xyz(lab: torch.Tensor) -> torch.Tensor:
  return _color_op('lab_to_xyz', lab)


def xyz_to_rgb(xyz: torch.Tensor) -> torch.Tensor:
  return _color_op('xyz_to_rgb', xyz)


def rgb_to_lab(rgb: torch.Tensor) -> torch.Tensor:
  return _color_op('rgb_to_lab', rgb)


def lab_to_rgb(lab: torch.Tensor) -> torch.Tensor:
  return _color_op('lab_to_rgb', lab)


def modify_hsl(rgb: torch.Tensor, hue_adjust: float = 0.0, sat_adjust: float = 0.0, lum_adjust: float = 0.0) -> torch.Tensor:
  return _color_op('modify_hsl', rgb, (hue_adjust, sat_adjust, lum_adjust))


def modify_vibrance(rgb: torch.Tensor, amount: float = 0.0) -> torch.Tensor:
  return _color_op('modify_vibrance', rgb, (amount, 0.0, 0.0))


def color_transform_3x3(input: torch.Tensor, matrix_3x3: torch.Tensor) -> torch.Tensor:
  _require(matrix_3x3.dtype == torch.float32, 'Matrix must be float32')
  _require(matrix_3x3.dim() == 2 and matrix_3x3.size(0) == 3 and matrix_3x3.size(1) == 3, 'Matrix must be (3, 3)')
  _require(matrix_3x3.is_cuda and matrix_3x3.is_contiguous(), 'Matrix tensor must be contiguous CUDA tensor')
  return _color_op('color_transform_3x3', input, matrix=matrix_3x3)


def _extract_luminance(rgb: torch.Tensor, log_mode: bool, eps: float, out_dtype: torch.dtype | None = None) -> torch.Tensor:
  _require(rgb.dtype in (torch.float32, torch.float16), 'Input must be float32')
  _require(rgb.dim() == 3 and rgb.size(2) == 3, 'Input must be (H, W, 3)')
  _require(rgb.is_cuda, 'Input must be on CUDA device')
  _require(rgb.is_contiguous(), 'Input tensor must be contiguous')
  out = torch.empty((rgb.size(0), rgb.size(1)), dtype=out_dtype or rgb.dtype, device=rgb.device)
  with torch.cuda.device(rgb.device):
    check(lib.tdk_compute_luminance(_ptr(rgb), _ptr(out), rgb.size(0) * rgb.size(1), int(log_mode), float(eps), _dtype_tag(rgb),
                                    _dtype_tag(out), _stream()))
  return out


def compute_luminance(rgb: torch.Tensor) -> torch.Tensor:
  return _extract_luminance(rgb, False, 1e-6)


def compute_log_luminance(rgb: torch.Tensor, eps: float) -> torch.Tensor:
  _require(eps > 0.0, 'Epsilon must be positive')
  return _extract_luminance(rgb, True, eps)


def _replace_luminance(rgb: torch.Tensor, lum: torch.Tensor, log_mode: bool) -> torch.Tensor:
  _require(rgb.dtype in (torch.float32, torch.float16), 'Input1 must be float32')
  _require(lum.dtype in (torch.float32, torch.float16), 'Input2 must be float32')
  _require(rgb.dim() == 3 and rgb.size(2) == 3, 'Input1 must be (H, W, 3)')
  _require(lum.dim() == 2, 'Input2 must be (H, W)')
  _require(rgb.is_cuda and lum.is_cuda, 'Inputs must be on CUDA device')
  _require(rgb.is_contiguous() and lum.is_contiguous(), 'Input tensors must be contiguous')
  _require(lum.size(0) == rgb.size(0) and lum.size(1) == rgb.size(1), 'Input dimensions must match')
  out = torch.empty_like(rgb)
  with torch.cuda.device(rgb.device):
    check(lib.tdk_modify_luminance(_ptr(rgb), _ptr(lum), _ptr(out), lum.numel(), int(log_mode), _dtype_tag(rgb), _dtype_tag(lum), _stream()))
  return out


def modify_luminance(rgb: torch.Tensor, new_luminance: torch.Tensor) -> torch.Tensor:
  return _replace_luminance(rgb, new_luminance, False)


def modify_log_luminance(rgb: torch.Tensor, log_luminance: torch.Tensor, eps: float) -> torch.Tensor:
  _require(eps > 0.0, 'Epsilon must be positive')
  return _replace_luminance(rgb, log_luminance, True)


# ------------------------------------------------------------------ statistics + tonemaps (csrc/tonemap/)
class TonemapParams:
  """reference csrc/tonemap/tonemap.h:6-15"""

  def __init__(self, gamma: float = 1.0, intensity: float = 0.0, light_adapt: float = 0.8, vibrance: float = 0.0):
    self.gamma = float(gamma)
    self.intensity = float(intensity)
    self.light_adapt = float(light_adapt)
    self.vibrance = float(vibrance)

  def __repr__(self) -> str:
    return f'TonemapParams(gamma={self.gamma}, intensity={self.intensity}, light_adapt={self.light_adapt}, vibrance={self.vibrance})'


def normalize_image(image: torch.Tensor, bounds: torch.Tensor) -> torch.Tensor:
  """(image - bounds[0]) / (bounds[1] - bounds[0]) in one pass, bounds stay on the device
  (reference torch_darktable/pipeline/util.py:8-10; not part of the reference's extension module)."""
  _require(image.is_cuda and image.dtype in (torch.float32, torch.float16), 'image must be a CUDA float32/float16 tensor')
  _require(bounds.numel() == 2 and bounds.dtype == torch.float32, 'bounds must be 2 float32 values')
  x = image.contiguous()
  b = bounds.to(x.device).contiguous()
  out = torch.empty_like(x)
  with torch.cuda.device(x.device):
    check(lib.tdk_normalize(_ptr(x), _ptr(out), x.numel(), _ptr(b), _dtype_tag(x), _stream()))
  return out


_BOUNDS_STATE: dict = {}


def compute_image_bounds(images: Sequence[torch.Tensor], stride: int = 8) -> torch.Tensor:
  """One launch per image (tdk_image_bounds): minimum and maximum go through a persistent per-(device, stream) state of four words that is
  idle between calls, the workgroup that draws the list's last ticket writes the result and resets it -- no init launch."""
  _require(len(images) > 0, 'images must be non-empty')
  dev = images[0].device
  for img in images:  # every image is checked before the first launch: a rejected list leaves the cached state idle
    _check_rgb(img, allow_half=True)
    _require(img.device == dev, f'image is on {img.device}, the first one on {dev}')
  bounds = torch.empty(2, dtype=torch.float32, device=dev)
  key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
  with torch.cuda.device(dev):
    state = _BOUNDS_STATE.get(key)
    if state is None:
      state = _BOUNDS_STATE[key] = torch.tensor([-1, 0, 0, 0], dtype=torch.int32, device=dev)  # 0xffffffff, 0, 0, 0 (host copy: complete on return)
    xs = [img.contiguous() for img in images]
    total = sum(lib.tdk_image_bounds_tickets(x.size(1), x.size(0), int(stride)) for x in xs)
    try:
      for i, x in enumerate(xs):
        last = i == len(xs) - 1
        check(lib.tdk_image_bounds(_ptr(x), x.size(1), x.size(0), int(stride), _ptr(state), _ptr(bounds) if last else None, total, _dtype_tag(x), _stream()))
    except Exception:
      _BOUNDS_STATE.pop(key, None)  # a failed launch may have left tickets behind: start from a fresh state next time
      raise
  return bounds


_UNIT_BOUNDS: dict = {}
_METRICS_STATE: dict = {}


def _unit_bounds(dev: torch.device) -> torch.Tensor:
  """Device-resident [0, 1] (no per-call host-to-device copy)."""
  key = (dev.type, dev.index)
  if key not in _UNIT_BOUNDS:
    _UNIT_BOUNDS[key] = torch.tensor([0.0, 1.0], dtype=torch.float32, device=dev)
  return _UNIT_BOUNDS[key]


def compute_image_metrics(images: Sequence[torch.Tensor], stride: int = 8, min_gray: float = 1e-4, rescale: bool = False) -> torch.Tensor:
  _require(len(images) > 0, 'images must be non-empty')
  dev = images[0].device
  _require(dev.type == 'cuda', 'image must be CUDA')
  bounds = compute_image_bounds(images, stride) if rescale else None
  key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
  acc = _METRICS_STATE.get(key)
  if acc is None:
    acc = _METRICS_STATE[key] = MetricsAccumulator(dev)
  for img in images:  # every image is checked before the first one is added: a bad list leaves nothing behind in the cached accumulator
    _check_rgb(img, allow_half=True)
    _require(img.device == dev, f'image is on {img.device}, the first one on {dev}')
  acc.stride, acc.min_gray = int(stride), float(min_gray)
  acc.bounds = bounds if bounds is not None else _unit_bounds(dev)
  try:
    for img in images:
      acc.add(img)
    return acc.finish()
  except Exception:
    acc.reset()
    raise


class MetricsAccumulator:
  """compute_image_metrics in two halves: `add(image)` takes one image's sample-grid sums (a wide accumulator, so the launch
  is not capped by same-address atomics), `finish()` normalises by the valid-sample count (color_adaption.cu:161-165), returns
  the 5 metrics on the device and leaves the accumulator zero for the next frame.  Stream order is the only synchronisation.
  compute_image_metrics(images) == [acc.add(i) for i in images]; acc.finish().
  The kernel of the LAST image added also does the finish (tdk_image_metrics: one launch, the workgroup that draws the last
  ticket sums the rows), so the usual one image per frame costs one launch; for that the launch of an added image is issued
  when the next image arrives or at finish() -- on the stream that is current THEN (if that is another stream than the one
  current at add(), it first waits for the work queued there: the image's producer).  Contract: an added image must not be written
  to (nor its storage reused) until the next add() / finish(): the accumulator keeps a reference, not a copy."""

  def __init__(self, device, stride: int = 8, min_gray: float = 1e-4, bounds: torch.Tensor | None = None):
    device = torch.device(device)
    _require(device.type == 'cuda', 'MetricsAccumulator needs a GPU device')
    self.stride, self.min_gray = int(stride), float(min_gray)
    self.bounds = bounds.to(device=device, dtype=torch.float32).contiguous() if bounds is not None else _unit_bounds(device)
    self.acc = torch.zeros(8192, dtype=torch.float32, device=device)  # TDK_METRICS_ACC_FLOATS: 1024 rows of 8
    self._pending: torch.Tensor | None = None
    self._pending_stream = None  # the stream that was current when the pending image was added

  def _launch(self, x: torch.Tensor, metrics: torch.Tensor | None) -> None:
    with torch.cuda.device(x.device):
      here = torch.cuda.current_stream(x.device)
      if self._pending_stream is not None and self._pending_stream != here:
        here.wait_stream(self._pending_stream)  # the image was produced on the stream of its add()
      self._pending_stream = None
      try:
        if metrics is None:
          check(lib.tdk_image_metrics_accumulate_rows(_ptr(x), x.size(1), x.size(0), self.stride, self.min_gray, _ptr(self.bounds), _ptr(self.acc),
                                                      _dtype_tag(x), _stream()))
        else:
          check(lib.tdk_image_metrics(_ptr(x), x.size(1), x.size(0), self.stride, self.min_gray, _ptr(self.bounds), _ptr(self.acc), _ptr(metrics),
                                      _dtype_tag(x), _stream()))
      except Exception:
        self.reset()  # never leave half-accumulated sums behind
        raise

  def add(self, image: torch.Tensor) -> None:
    try:
      _check_rgb(image, allow_half=True)
      _require(image.device == self.acc.device, f'image is on {image.device}, the accumulator on {self.acc.device}')
    except Exception:
      self.reset()  # a rejected image ends the list: nothing of it may leak into the next one
      raise
    if self._pending is not None:
      pending, self._pending = self._pending, None
      self._launch(pending, None)
    self._pending = image.contiguous()
    self._pending_stream = torch.cuda.current_stream(image.device)

  def finish(self) -> torch.Tensor:
    metrics = torch.empty(5, dtype=torch.float32, device=self.acc.device)
    if self._pending is not None:
      pending, self._pending = self._pending, None
      self._launch(pending, metrics)
    else:  # nothing added since the last finish: the metrics of an empty list (all zero: the valid count is clamped to 1)
      with torch.cuda.device(self.acc.device):
        check(lib.tdk_image_metrics_finish_reset(_ptr(self.acc), _ptr(metrics), _stream()))
    return metrics

  def reset(self) -> None:
    self._pending = None
    self._pending_stream = None
    self.acc.zero_()


def _tonemap(mode: int, image: torch.Tensor, metrics: torch.Tensor | None, params: TonemapParams) -> torch.Tensor:
  _check_rgb(image, allow_half=True)
  if metrics is not None:
    _require(metrics.dtype == torch.float32 and metrics.numel() == 5, 'metrics must be 5 float32 values')
    metrics = metrics.to(image.device).contiguous()
  x = image.contiguous()
  out = torch.empty((x.size(0), x.size(1), 3), dtype=torch.uint8, device=x.device)
  with torch.cuda.device(x.device):
    check(lib.tdk_tonemap(_ptr(x), _ptr(out), x.size(0) * x.size(1), mode, _ptr(metrics), params.gamma, params.intensity, params.light_adapt,
                          params.vibrance, _dtype_tag(x), _stream()))
  return out


def reinhard_tonemap(image: torch.Tensor, metrics: torch.Tensor, params: TonemapParams) -> torch.Tensor:
  return _tonemap(0, image, metrics, params)


def aces_tonemap(image: torch.Tensor, params: TonemapParams) -> torch.Tensor:
  return _tonemap(1, image, None, params)


def adaptive_aces_tonemap(image: torch.Tensor, metrics: torch.Tensor, params: TonemapParams) -> torch.Tensor:
  return _tonemap(2, image, metrics, params)


def linear_tonemap(image: torch.Tensor, metrics: torch.Tensor, params: TonemapParams) -> torch.Tensor:
  return _tonemap(3, image, metrics, params)


# ------------------------------------------------------------------ Wiener (csrc/denoise/denoise.cu:245-364)
class Wiener(_Workspace):
  def __init__(self, device, width: int, height: int, overlap_factor: int = 4, tile_size: int = 32):
    super().__init__(device, width, height)
    self._overlap_factor = int(overlap_factor)
    self._tile_size = int(tile_size)

  @property
  def overlap_factor(self) -> int:
    return self._overlap_factor

  def process(self, input: torch.Tensor, noise_sigmas: torch.Tensor) -> torch.Tensor:
    _require(input.dim() == 3, 'expected HWC tensor')
    _require(input.device == self._device, 'input device mismatch')
    c = input.size(2)
    _require(c in (1, 3), f'input channels must be 1 or 3, got {c}')
    _require(noise_sigmas.numel() == c, 'noise_sigmas must have C elements')
    x = input.contiguous()
    sig = noise_sigmas.to(device=x.device, dtype=torch.float32).contiguous()
    h, w = x.size(0), x.size(1)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
      nbytes = lib.tdk_wiener_workspace_bytes(w, h, c, self._tile_size, self._overlap_factor)
      ws = self._workspace(nbytes, x.device)
      check(lib.tdk_wiener(_ptr(x), _ptr(out), _ptr(ws), w, h, c, self._tile_size, self._overlap_factor, _ptr(sig), _dtype_tag(x), _stream()))
    return out

  def process_log_luminance(self, image: torch.Tensor, noise_sigmas: torch.Tensor, eps: float = 1e-4, luminance_out: torch.Tensor | None = None,
                            luminance_log: bool = False, luminance_eps: float = 1e-6) -> torch.Tensor:
    """Fused form of the reference wrapper's compute_log_luminance -> process -> modify_log_luminance
    (torch_darktable/denoise.py:54-58): one library call, the log-luminance planes stay in fp32
    scratch and the denoised one is consumed in place.

    luminance_out: optional float32 (H, W) tensor that receives compute_luminance(result) (or
    compute_log_luminance(result, luminance_eps) with luminance_log) -- bit for bit what those ops return -- for a
    consumer that would extract it next (Bilateral.process_rgb(..., luminance=...))."""
    _check_rgb(image, 'image', allow_half=True)
    _require(image.device == self._device, 'input device mismatch')
    _require(image.size(0) == self._height and image.size(1) == self._width, 'Input dimensions must match workspace size')
    _require(eps > 0.0, 'Epsilon must be positive')
    x = image.contiguous()
    sig = noise_sigmas.to(device=x.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
      nbytes = lib.tdk_wiener_log_luminance_workspace_bytes(self._width, self._height, self._tile_size, self._overlap_factor)
      ws = self._workspace(nbytes, x.device)
      if luminance_out is None:
        check(lib.tdk_wiener_log_luminance(_ptr(x), _ptr(out), _ptr(ws), self._width, self._height, self._tile_size, self._overlap_factor,
                                           _ptr(sig), float(eps), _dtype_tag(x), _stream()))
      else:
        _require(luminance_out.dtype == torch.float32 and luminance_out.is_contiguous() and luminance_out.device == x.device
                 and tuple(luminance_out.shape) == (self._height, self._width), 'luminance_out must be a contiguous float32 (H, W) tensor on the image device')
        check(lib.tdk_wiener_log_luminance_lum(_ptr(x), _ptr(out), _ptr(ws), self._width, self._height, self._tile_size, self._overlap_factor,
                                               _ptr(sig), float(eps), _dtype_tag(x), _ptr(luminance_out), int(luminance_log), float(luminance_eps), _stream()))
    return out


  def process_log_luminance_lab(self, image: torch.Tensor, noise_sigmas: torch.Tensor, eps: float = 1e-4, luminance_out: torch.Tensor | None = None,
                                chroma_out: torch.Tensor | None = None, bounds: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """Lab hand-over form of process_log_luminance (include/tdk_hip.h: tdk_wiener_log_luminance_lab): instead of the denoised RGB
    image it returns (luminance, chroma) = the float32 (H, W) plane compute_luminance(denoised) and the float32 (H, W, 2) plane
    of the denoised pixels' Lab (a, b) -- what Bilateral.process_lab takes.  The RGB image between the two stages is never
    formed: one colour round trip for the two stages instead of two (tolerance of the colour operators, not bit for bit).
    bounds: optional 2 device floats; the image is normalised as pipeline.util.normalize_image(image, bounds) would while it is read."""
    _check_rgb(image, 'image', allow_half=True)
    _require(image.device == self._device, 'input device mismatch')
    _require(image.size(0) == self._height and image.size(1) == self._width, 'Input dimensions must match workspace size')
    _require(eps > 0.0, 'Epsilon must be positive')
    x = image.contiguous()
    sig = noise_sigmas.to(device=x.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
    lum = luminance_out if luminance_out is not None else torch.empty((self._height, self._width), dtype=torch.float32, device=x.device)
    ab = chroma_out if chroma_out is not None else torch.empty((self._height, self._width, 2), dtype=torch.float32, device=x.device)
    _require(lum.dtype == torch.float32 and lum.is_contiguous() and lum.device == x.device and tuple(lum.shape) == (self._height, self._width),
             'luminance_out must be a contiguous float32 (H, W) tensor on the image device')
    _require(ab.dtype == torch.float32 and ab.is_contiguous() and ab.device == x.device and tuple(ab.shape) == (self._height, self._width, 2),
             'chroma_out must be a contiguous float32 (H, W, 2) tensor on the image device')
    with torch.cuda.device(x.device):
      nbytes = lib.tdk_wiener_log_luminance_workspace_bytes(self._width, self._height, self._tile_size, self._overlap_factor)
      ws = self._workspace(nbytes, x.device)
      if bounds is not None:
        _require(bounds.dtype == torch.float32 and bounds.numel() == 2 and bounds.device == x.device and bounds.is_contiguous(), 'bounds must be 2 float32 values on the image device')
      check(lib.tdk_wiener_log_luminance_lab(_ptr(x), _ptr(ws), self._width, self._height, self._tile_size, self._overlap_factor, _ptr(sig), float(eps),
                                             _ptr(bounds), _dtype_tag(x), _ptr(lum), _ptr(ab), _stream()))
    return lum, ab


def create_wiener(device, width: int, height: int, overlap_factor: int = 4, tile_size: int = 32) -> Wiener:
  """reference torch_darktable_extension.pyi:171-177 (declared there; the reference's extension.cpp never
  registers it -- its Python helper denoise.create_wiener builds the wrapper class instead)."""
  return Wiener(device, width, height, overlap_factor, tile_size)


# ------------------------------------------------------------------ local contrast
class Bilateral(_Workspace):
  """reference csrc/local_contrast/bilateral.cu:252-404 (extension.cpp:111-121)"""

  def __init__(self, device, width: int, height: int, sigma_s: float = 8.0, sigma_r: float = 0.1):
    super().__init__(device, width, height)
    _require(self._width > 0 and self._height > 0, 'Invalid dimensions')
    self._sigma_s = float(sigma_s)
    self._sigma_r = float(sigma_r)

  @property
  def sigma_s(self) -> float:
    return self._sigma_s

  @sigma_s.setter
  def sigma_s(self, v: float) -> None:
    self._sigma_s = float(v)
    self._drop_workspaces()

  @property
  def sigma_r(self) -> float:
    return self._sigma_r

  @sigma_r.setter
  def sigma_r(self, v: float) -> None:
    self._sigma_r = float(v)
    self._drop_workspaces()

  def grid_size(self) -> tuple[int, int, int]:
    sz = (C.c_int * 3)()
    check(lib.tdk_bilateral_grid_size(self._width, self._height, self._sigma_s, self._sigma_r, sz))
    return tuple(sz)

  def _prepared_workspace(self, nbytes: int, device: torch.device) -> tuple[torch.Tensor, int]:
    """The per-stream workspace and the flags of a call on it.  The tile kernel's axis tables live at the start of the
    workspace and depend on the geometry and the sigmas only: they are built when a workspace buffer is (re)allocated
    (tdk_bilateral_prepare) and every later call vouches for them -- the reference likewise keeps its grids in the object and
    drops them when a sigma changes (bilateral.cu:389-390; the sigma setters above drop the workspaces)."""
    ws = self._workspace(nbytes, device)
    ready = self.__dict__.setdefault('_tables_ready', {})
    key = torch.cuda.current_stream(device).cuda_stream
    if ready.get(key) is not ws:  # the buffer object itself, not its address: a freed block can come back at the same address
      check(lib.tdk_bilateral_prepare(_ptr(ws), self._width, self._height, self._sigma_s, self._sigma_r, _stream()))
      ready[key] = ws
    return ws, TDK_BILATERAL_PREPARED | getattr(_verify, 'bil', 0)

  def _drop_workspaces(self) -> None:
    super()._drop_workspaces()
    self.__dict__.pop('_tables_ready', None)

  def process(self, luminance: torch.Tensor, detail: float) -> torch.Tensor:
    _require(luminance.dtype in (torch.float32, torch.float16), 'Input must be float32')
    _require(luminance.dim() == 2, 'Input must be 2D (H,W)')
    _require(luminance.size(0) == self._height and luminance.size(1) == self._width, 'Input shape must match (H,W)')
    _require(luminance.is_cuda, 'Input must be CUDA tensor')
    x = luminance.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
      # one buffer serves both entry points (the rgb layout is the larger one and starts with the same tables)
      ws, flags = self._prepared_workspace(lib.tdk_bilateral_workspace_bytes(self._width, self._height, self._sigma_s, self._sigma_r), x.device)
      check(lib.tdk_bilateral_ex(_ptr(x), _ptr(out), _ptr(ws), self._width, self._height, self._sigma_s, self._sigma_r, float(detail),
                                 _dtype_tag(x), flags, _stream()))
    return out

  def _process_rgb(self, image: torch.Tensor, detail: float, log_mode: bool, eps: float, luminance: torch.Tensor | None = None,
                   metrics: 'MetricsAccumulator | None' = None) -> torch.Tensor:
    _check_rgb(image, 'image', allow_half=True)
    _require(image.size(0) == self._height and image.size(1) == self._width, 'Input shape must match (H,W)')
    x = image.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
      ws, flags = self._prepared_workspace(lib.tdk_bilateral_rgb_workspace_bytes(self._width, self._height, self._sigma_s, self._sigma_r), x.device)
      if luminance is not None:
        _require(luminance.dtype == torch.float32 and luminance.is_contiguous() and luminance.device == x.device
                 and tuple(luminance.shape) == (self._height, self._width), 'luminance must be a contiguous float32 (H, W) tensor on the image device')
      check(lib.tdk_bilateral_rgb_ex(_ptr(x), _ptr(luminance), _ptr(out), _ptr(ws), self._width, self._height, self._sigma_s, self._sigma_r,
                                     float(detail), int(log_mode), float(eps), _dtype_tag(x), flags, _stream()))
    if metrics is not None:
      metrics.add(out)
    return out

  def process_lab(self, luminance: torch.Tensor, chroma: torch.Tensor, detail: float, out_dtype: torch.dtype = torch.float32,
                  metrics: 'MetricsAccumulator | None' = None) -> torch.Tensor:
    """process_rgb for a pixel that arrives as Lab (Wiener.process_log_luminance_lab): filter the float32 (H, W) lightness plane
    and write modify_luminance's result -- clip(lab_to_rgb(filtered L, a, b)) -- as an (H, W, 3) image of out_dtype."""
    _require(luminance.is_cuda and luminance.dtype == torch.float32 and luminance.is_contiguous() and tuple(luminance.shape) == (self._height, self._width),
             'luminance must be a contiguous float32 (H, W) CUDA tensor')
    _require(chroma.dtype == torch.float32 and chroma.is_contiguous() and chroma.device == luminance.device
             and tuple(chroma.shape) == (self._height, self._width, 2), 'chroma must be a contiguous float32 (H, W, 2) tensor on the same device')
    _require(out_dtype in (torch.float32, torch.float16), 'out_dtype must be float32 or float16')
    out = torch.empty((self._height, self._width, 3), dtype=out_dtype, device=luminance.device)
    with torch.cuda.device(luminance.device):
      ws, flags = self._prepared_workspace(lib.tdk_bilateral_rgb_workspace_bytes(self._width, self._height, self._sigma_s, self._sigma_r), luminance.device)
      check(lib.tdk_bilateral_lab(_ptr(luminance), _ptr(chroma), _ptr(out), _ptr(ws), self._width, self._height, self._sigma_s, self._sigma_r, float(detail),
                                  _dtype_tag(out), flags, _stream()))
    if metrics is not None:
      metrics.add(out)
    return out

  def process_rgb(self, image: torch.Tensor, detail: float, luminance: torch.Tensor | None = None,
                  metrics: 'MetricsAccumulator | None' = None) -> torch.Tensor:
    """Fused compute_luminance -> process -> modify_luminance (reference local_contrast.py:109-114).
    luminance: the float32 plane compute_luminance(image) if the producer of `image` already has it
    (Wiener.process_log_luminance(..., luminance_out=...)); metrics: a MetricsAccumulator that is fed the result
    (the pipeline computes compute_image_metrics of it next)."""
    return self._process_rgb(image, detail, False, 1e-6, luminance, metrics)

  def process_log_rgb(self, image: torch.Tensor, detail: float, eps: float = 1e-6, luminance: torch.Tensor | None = None,
                      metrics: 'MetricsAccumulator | None' = None) -> torch.Tensor:
    """Fused compute_log_luminance -> process -> modify_log_luminance (reference local_contrast.py:116-125)."""
    _require(eps > 0.0, 'Epsilon must be positive')
    return self._process_rgb(image, detail, True, eps, luminance, metrics)


class Laplacian(_Workspace):
  """reference csrc/local_contrast/laplacian.cu:392-635 (extension.cpp:94-108)"""

  def __init__(self, device, width: int, height: int, num_gamma: int = 6, sigma: float = 0.2, shadows: float = 1.0,
               highlights: float = 1.0, clarity: float = 0.0):
    super().__init__(device, width, height)
    if int(num_gamma) != 6:
      raise RuntimeError(f'Unsupported gamma count: {num_gamma}')
    self._num_gamma = 6
    self.sigma = float(sigma)
    self.shadows = float(shadows)
    self.highlights = float(highlights)
    self.clarity = float(clarity)

  def process(self, input: torch.Tensor) -> torch.Tensor:
    tag = _dtype_tag(input)  # float16 storage (extension) carries the same values: the reference's result is binary16 stored as float32
    _require(input.dim() == 2, 'Input tensor must be 2D')
    _require(input.size(0) == self._height and input.size(1) == self._width, 'Input tensor dimensions must match workspace dimensions')
    _require(input.is_cuda, 'Input tensor must be on CUDA device')
    x = input.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
      ws = self._workspace(lib.tdk_laplacian_workspace_bytes(self._width, self._height, self._num_gamma), x.device)
      check(lib.tdk_laplacian_ex(_ptr(x), _ptr(out), _ptr(ws), self._width, self._height, self._num_gamma, self.sigma, self.shadows,
                                 self.highlights, self.clarity, tag, _stream()))
    return out


# ------------------------------------------------------------------ JPEG (after the hot path)
class JpegException(Exception):
  pass


class JpegInputFormat(enum.IntEnum):
  """reference csrc/jpeg_encoder.h (enum order)"""

  BGR = 0
  RGB = 1
  BGRI = 2
  RGBI = 3


class JpegSubsampling(enum.IntEnum):
  CSS_444 = 0
  CSS_422 = 1
  CSS_GRAY = 2


# pybind's .export_values() (extension.cpp:231-245): the members are also module attributes
globals().update(JpegInputFormat.__members__)
globals().update(JpegSubsampling.__members__)


class Jpeg:
  """Jpeg.encode of the reference (csrc/jpeg_encoder.cu:104-180, an nvjpeg wrapper): uint8 image on the device in, the JPEG byte
  stream as a CPU uint8 tensor out; optimised Huffman tables, 4:4:4 / 4:2:2 / gray, baseline or progressive.  The encoder is
  the device encoder of csrc/jpeg.hip (colour conversion, FDCT, quantisation, Huffman coding and byte stuffing on the GPU; only the
  finished stream crosses PCIe).  nvjpeg's exact bytes are not reproducible (closed library, nothing pinned in the reference):
  the stream is checked against the CPU restatement of T.81 in oracle/ byte for byte and against libjpeg's decode.
  Checks and messages as in the reference (createImage, interleavedImage / planarImage)."""

  def __init__(self):
    self._workspace = None  # (key, tensor): the coder's scratch is kept between calls like nvjpeg's encoder state

  def encode(self, image: torch.Tensor, quality: int, input_format: int, subsampling: int, progressive: bool) -> torch.Tensor:
    try:
      fmt, sub = JpegInputFormat(int(input_format)), JpegSubsampling(int(subsampling))
    except ValueError as e:
      raise RuntimeError(f'Invalid input format or subsampling: {e}') from e
    _require(image.is_cuda, 'Input image should be on CUDA device')
    _require(image.dtype == torch.uint8, 'Input image should be uint8')
    _require(image.is_contiguous(), 'Input data should be contiguous')
    if fmt in (JpegInputFormat.BGRI, JpegInputFormat.RGBI):
      _require(image.dim() == 3 and image.size(2) == 3, 'for interleaved (BGRI, RGBI) expected 3D tensor (H, W, C)')
      h, w = int(image.size(0)), int(image.size(1))
    else:
      _require(image.dim() == 3 and image.size(0) == 3, 'for planar (BGR, RGB) expected 3D tensor (C, H, W)')
      h, w = int(image.size(1)), int(image.size(2))
    with torch.cuda.device(image.device):
      nbytes = lib.tdk_jpeg_workspace_bytes(w, h, int(sub))
      if nbytes == 0:
        raise JpegException(f'nvjpegEncodeImage, image {w}x{h} not supported')
      key = (w, h, int(sub), image.device)
      if self._workspace is None or self._workspace[0] != key:
        self._workspace = (key, torch.empty(nbytes, dtype=torch.uint8, device=image.device))
      ws = self._workspace[1]
      length = C.c_size_t(0)
      rc = lib.tdk_jpeg_encode(_ptr(image), w, h, int(fmt), int(quality), int(sub), int(bool(progressive)), _ptr(ws), C.byref(length), _stream())
      if rc != 0:
        raise JpegException(f'nvjpegEncodeImage, {lib.tdk_last_error().decode()}')
      buffer = torch.empty(int(length.value), dtype=torch.uint8)
      rc = lib.tdk_jpeg_retrieve(_ptr(ws), w, h, int(sub), C.c_void_p(buffer.data_ptr()), length, _stream())
      if rc != 0:
        raise JpegException(f'nvjpegEncodeRetrieveBitstream, {lib.tdk_last_error().decode()}')
    return buffer

  def __repr__(self) -> str:
    return 'Jpeg'
