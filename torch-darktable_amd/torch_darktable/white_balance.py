"""White balance on the raw mosaic (reference torch_darktable/white_balance.py)."""

from __future__ import annotations

import torch

from .debayer import BayerPattern
from .extension import extension


def apply_white_balance(bayer_image: torch.Tensor, gains: torch.Tensor, pattern: BayerPattern) -> torch.Tensor:
    """Multiply each CFA site of an (H, W) mosaic by its channel gain [R, G, B], clamp to [0, 1]."""
    return extension.apply_white_balance(bayer_image, gains, pattern.value)


def estimate_white_balance(bayer_images: list, pattern: BayerPattern, quantile: float = 0.98, stride: int = 8) -> torch.Tensor:
    """Grey-world gains from the brightest (>= quantile) unsaturated 2x2 cells; green = 1."""
    return extension.estimate_white_balance(bayer_images, pattern.value, quantile, stride)


__all__ = ['apply_white_balance', 'estimate_white_balance']
