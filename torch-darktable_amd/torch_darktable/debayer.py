"""Demosaic workspaces, post-processing and the 12-bit raw codec (reference torch_darktable/debayer.py)."""

from __future__ import annotations

import torch

from .bayer import BayerPattern, PackedFormat
from .extension import extension


def _expect(tensor: torch.Tensor, shape: tuple, who: str) -> None:
    if tuple(tensor.shape) != tuple(shape):
        raise RuntimeError(f'{who} input shape {tuple(tensor.shape)} != expected {tuple(shape)}')


class Bilinear5x5:
    """Stateless 13-tap linear demosaic with the workspace-style `process` method."""

    def __init__(self, bayer_pattern: BayerPattern):
        self.bayer_pattern = bayer_pattern

    def process(self, image: torch.Tensor) -> torch.Tensor:
        return bilinear5x5_demosaic(image, self.bayer_pattern)


class PPG:
    """Pattern Pixel Grouping demosaic for a fixed image size."""

    def __init__(self, device: torch.device, image_size: tuple[int, int], bayer_pattern: BayerPattern, *, median_threshold: float = 0.0):
        w, h = image_size
        self._ppg = extension.PPG(device, w, h, bayer_pattern.value, float(median_threshold))

    def process(self, input_tensor: torch.Tensor) -> torch.Tensor:
        _expect(input_tensor, (self._ppg.height, self._ppg.width, 1), 'PPG')
        return self._ppg.process(input_tensor)

    @property
    def image_size(self) -> tuple[int, int]:
        return (self._ppg.width, self._ppg.height)

    @property
    def median_threshold(self) -> float:
        return self._ppg.median_threshold


class RCD:
    """Ratio Corrected Demosaic for a fixed image size."""

    def __init__(self, device: torch.device, image_size: tuple[int, int], bayer_pattern: BayerPattern):
        w, h = image_size
        self._rcd = extension.RCD(device, w, h, bayer_pattern.value)

    def process(self, input_tensor: torch.Tensor) -> torch.Tensor:
        _expect(input_tensor, (self._rcd.height, self._rcd.width, 1), 'RCD')
        return self._rcd.process(input_tensor)

    def process_packed(self, packed_data: torch.Tensor, white_balance: torch.Tensor | None = None,
                       format_type: PackedFormat = PackedFormat.Packed12, output_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """12-bit packed bytes -> demosaiced RGB in one library call: decode12 + apply_white_balance + process,
        with the same result as the three calls (the mosaic planes in between are never written)."""
        return self._rcd.process_packed12(packed_data, white_balance, format_type is PackedFormat.Packed12_IDS, output_dtype)

    @property
    def image_size(self) -> tuple[int, int]:
        return (self._rcd.width, self._rcd.height)


class PostProcess:
    """Colour smoothing and green equilibration after demosaic."""

    def __init__(self, device: torch.device, image_size: tuple[int, int], bayer_pattern: BayerPattern, *,
                 color_smoothing_passes: int = 0, green_eq_local: bool = False, green_eq_global: bool = False,
                 green_eq_threshold: float = 0.04):
        w, h = image_size
        self._postprocess = extension.PostProcess(device, w, h, bayer_pattern.value, int(color_smoothing_passes), bool(green_eq_local),
                                                  bool(green_eq_global), float(green_eq_threshold))

    def process(self, input_tensor: torch.Tensor) -> torch.Tensor:
        _expect(input_tensor, (self._postprocess.height, self._postprocess.width, 3), 'PostProcess')
        return self._postprocess.process(input_tensor)

    @property
    def image_size(self) -> tuple[int, int]:
        return (self._postprocess.width, self._postprocess.height)

    @property
    def color_smoothing_passes(self) -> int:
        return self._postprocess.color_smoothing_passes

    @property
    def green_eq_threshold(self) -> float:
        return self._postprocess.green_eq_threshold


# ---- 12-bit packed raw data
def encode(image: torch.Tensor, format_type: PackedFormat = PackedFormat.Packed12, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Pack a flat uint16 / float32 pixel buffer into 12-bit triples (floats are scaled by 4095)."""
    assert dtype in {torch.float32, torch.uint16}
    ids = format_type is PackedFormat.Packed12_IDS
    if image.dtype == torch.uint16:
        return extension.encode12_u16(image, ids_format=ids)
    if image.dtype == torch.float32:
        return extension.encode12_float(image, ids_format=ids)
    raise ValueError(f'Unsupported input dtype: {image.dtype}')


_DECODERS = {torch.float32: 'decode12_float', torch.float16: 'decode12_half', torch.uint16: 'decode12_u16'}


def decode12(packed_data: torch.Tensor, output_dtype: torch.dtype = torch.float32, format_type: PackedFormat = PackedFormat.Packed12) -> torch.Tensor:
    """Unpack 12-bit triples to float32 / float16 (scaled to [0, 1]) or uint16."""
    if output_dtype not in _DECODERS:
        raise ValueError(f'Unsupported output dtype: {output_dtype}')
    return getattr(extension, _DECODERS[output_dtype])(packed_data, ids_format=format_type is PackedFormat.Packed12_IDS)


encode12_u16 = extension.encode12_u16
encode12_float = extension.encode12_float
decode12_float = extension.decode12_float
decode12_half = extension.decode12_half
decode12_u16 = extension.decode12_u16


def bilinear5x5_demosaic(image: torch.Tensor, bayer_pattern: BayerPattern) -> torch.Tensor:
    """(H, W, 1) mosaic -> (H, W, 3) RGB with the 13-tap diamond filters."""
    return extension.bilinear5x5_demosaic(image, bayer_pattern.value)


__all__ = ['PPG', 'RCD', 'BayerPattern', 'Bilinear5x5', 'PackedFormat', 'PostProcess', 'bilinear5x5_demosaic', 'decode12',
           'decode12_float', 'decode12_half', 'decode12_u16', 'encode', 'encode12_float', 'encode12_u16']
