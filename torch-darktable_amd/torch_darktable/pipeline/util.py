"""Small tensor helpers of the pipeline (reference torch_darktable/pipeline/util.py)."""

from __future__ import annotations

import torch


def lerp(a: torch.Tensor, b: torch.Tensor, t: float) -> torch.Tensor:
    """a + (b - a) * t -- the exponential moving average step for bounds / metrics (stays on the device)."""
    return a + (b - a) * t


def normalize_image(rgb_raw: torch.Tensor, bounds: torch.Tensor) -> torch.Tensor:
    """Map [bounds[0], bounds[1]] to [0, 1]; bounds is a 2-element device tensor (no host read-back).
    GPU images take the library's one-pass kernel (same IEEE expression, bit-identical to the torch
    ops below, which remain for host tensors -- the reference compiles this line with torch.compile)."""
    if rgb_raw.is_cuda and rgb_raw.dtype in (torch.float32, torch.float16):
        from ..torch_darktable_extension import normalize_image as _normalize
        return _normalize(rgb_raw, bounds)
    return (rgb_raw - bounds[0]) / (bounds[1] - bounds[0])


def resize(image: torch.Tensor, size: tuple[int, int]) -> torch.Tensor:
    """Bilinear resize of an (H, W, C) image to `size` = (height, width) as torch interpolate takes it."""
    chw = image.permute(2, 0, 1).unsqueeze(0)
    out = torch.nn.functional.interpolate(chw, size=size, mode='bilinear', align_corners=False)
    return out.squeeze(0).permute(1, 2, 0).contiguous()


def resize_longest_edge(size: tuple[int, int], longest: int) -> tuple[int, int]:
    """Scale (w, h) so its longer edge becomes `longest` (0 = keep)."""
    if longest == 0:
        return size
    w, h = size
    return (longest, h * longest // w) if w > h else (w * longest // h, longest)


def resize_image(image: torch.Tensor, longest: int) -> torch.Tensor:
    h, w = image.shape[:2]
    return resize(image, resize_longest_edge((w, h), longest))
