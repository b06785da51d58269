"""Bayer-pattern helpers: enums, RGB -> mosaic, plane stacking (reference torch_darktable/bayer.py)."""

from __future__ import annotations

from enum import Enum
from pathlib import Path

import torch

from .extension import extension


class BayerPattern(Enum):
    """Wraps the extension enum, as the reference does (bayer.py:12-16)."""

    RGGB = extension.BayerPattern.RGGB
    BGGR = extension.BayerPattern.BGGR
    GRBG = extension.BayerPattern.GRBG
    GBRG = extension.BayerPattern.GBRG


class PackedFormat(Enum):
    Packed12 = 0
    Packed12_IDS = 1


# per pattern: site class (R=0, G1=1, G2=2, B=3) and RGB channel of the four 2x2 positions
# (row-major).  Values as in reference bayer.py:70-95 -- including its GRBG / GBRG channel rows,
# which list green for position (1, 0); rgb_to_bayer inherits that.
_SITE_CLASS = {
    BayerPattern.RGGB: (0, 1, 2, 3),
    BayerPattern.BGGR: (3, 1, 2, 0),
    BayerPattern.GRBG: (1, 0, 3, 2),
    BayerPattern.GBRG: (1, 3, 0, 2),
}
_CHANNEL = {
    BayerPattern.RGGB: (0, 1, 1, 2),
    BayerPattern.BGGR: (2, 1, 1, 0),
    BayerPattern.GRBG: (1, 0, 1, 2),
    BayerPattern.GBRG: (1, 2, 1, 0),
}


def pixel_order(pattern: BayerPattern) -> tuple[int, int, int, int]:
    if pattern not in _SITE_CLASS:
        raise ValueError(f'Invalid bayer pattern: {pattern}')
    return _SITE_CLASS[pattern]


def channels(pattern: BayerPattern) -> tuple[int, int, int, int]:
    if pattern not in _CHANNEL:
        raise ValueError(f'Invalid bayer pattern: {pattern}')
    return _CHANNEL[pattern]


def stack_bayer(bayer_image: torch.Tensor) -> torch.Tensor:
    """(H, W) mosaic -> (H/2, W/2, 4) planes in 2x2 row-major order."""
    return torch.stack([bayer_image[r::2, c::2] for r in (0, 1) for c in (0, 1)], dim=-1)


def expand_bayer(x: torch.Tensor) -> torch.Tensor:
    """(H/2, W/2, 4) planes -> (H, W, 1) mosaic (inverse of stack_bayer)."""
    h2, w2 = x.shape[0], x.shape[1]
    mosaic = torch.zeros(h2 * 2, w2 * 2, device=x.device, dtype=x.dtype)
    for k, (r, c) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        mosaic[r::2, c::2] = x[..., k]
    return mosaic.unsqueeze(-1)


def rgb_to_bayer(rgb_tensor: torch.Tensor, pattern: BayerPattern = BayerPattern.RGGB) -> torch.Tensor:
    """Sample an (H, W, 3) image on the CFA: returns the (H, W, 1) mosaic."""
    ch = channels(pattern)
    planes = [rgb_tensor[r::2, c::2, ch[2 * r + c]] for r in (0, 1) for c in (0, 1)]
    return expand_bayer(torch.stack(planes, dim=-1))


def load_as_bayer(image_path: Path, pattern: BayerPattern = BayerPattern.RGGB, device: torch.device = torch.device('cuda')) -> torch.Tensor:
    """Read an image file, scale to [0, 1] and mosaic it on `device`."""
    if not Path(image_path).exists():
        raise FileNotFoundError(f'Image not found: {image_path}')
    import numpy as np
    from PIL import Image  # Pillow instead of the reference's OpenCV (not installed here)

    rgb = np.asarray(Image.open(image_path).convert('RGB'), dtype=np.float32) / 255.0
    return rgb_to_bayer(torch.from_numpy(rgb).to(device), pattern)
