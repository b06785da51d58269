"""Colour-space operators on (H, W, 3) float32 images (reference torch_darktable/color_conversion.py).

sRGB / D65; Lab is normalised (L/100, a/128, b/128).  Thin pass-throughs to the kernel module."""

from __future__ import annotations

import torch

from .extension import extension


def compute_luminance(rgb_image: torch.Tensor) -> torch.Tensor:
    """Lab lightness (0..1) of the clipped image -> (H, W)."""
    return extension.compute_luminance(rgb_image)


def modify_luminance(rgb_image: torch.Tensor, luminance_multiplier: torch.Tensor) -> torch.Tensor:
    """Replace the Lab lightness by the given plane, keep a/b."""
    return extension.modify_luminance(rgb_image, luminance_multiplier)


def compute_log_luminance(rgb_image: torch.Tensor, eps: float) -> torch.Tensor:
    """log(max(eps, lightness)) -> (H, W)."""
    return extension.compute_log_luminance(rgb_image, eps)


def modify_log_luminance(rgb_image: torch.Tensor, log_luminance: torch.Tensor, eps: float) -> torch.Tensor:
    """Replace the lightness by exp(log_luminance), keep a/b."""
    return extension.modify_log_luminance(rgb_image, log_luminance, eps)


def modify_hsl(rgb_image: torch.Tensor, hue_adjust: float = 0.0, sat_adjust: float = 0.0, lum_adjust: float = 0.0) -> torch.Tensor:
    """Hue shift (turns), power-law saturation / lightness adjustment."""
    return extension.modify_hsl(rgb_image, hue_adjust, sat_adjust, lum_adjust)


def modify_vibrance(rgb_image: torch.Tensor, amount: float = 0.0) -> torch.Tensor:
    """darktable-style Lab vibrance: chroma-weighted saturation boost with slight darkening."""
    return extension.modify_vibrance(rgb_image, amount)


def rgb_to_lab(rgb_image: torch.Tensor) -> torch.Tensor:
    return extension.rgb_to_lab(rgb_image)


def lab_to_rgb(lab_image: torch.Tensor) -> torch.Tensor:
    return extension.lab_to_rgb(lab_image)


def rgb_to_xyz(rgb_image: torch.Tensor) -> torch.Tensor:
    return extension.rgb_to_xyz(rgb_image)


def xyz_to_lab(xyz_image: torch.Tensor) -> torch.Tensor:
    return extension.xyz_to_lab(xyz_image)


def lab_to_xyz(lab_image: torch.Tensor) -> torch.Tensor:
    return extension.lab_to_xyz(lab_image)


def xyz_to_rgb(xyz_image: torch.Tensor) -> torch.Tensor:
    return extension.xyz_to_rgb(xyz_image)


def color_transform_3x3(image: torch.Tensor, matrix_3x3: torch.Tensor) -> torch.Tensor:
    """out = clip(M @ rgb) per pixel, M a (3, 3) device tensor."""
    return extension.color_transform_3x3(image, matrix_3x3)


__all__ = ['color_transform_3x3', 'compute_log_luminance', 'compute_luminance', 'lab_to_rgb', 'lab_to_xyz', 'modify_hsl',
           'modify_log_luminance', 'modify_luminance', 'modify_vibrance', 'rgb_to_lab', 'rgb_to_xyz', 'xyz_to_lab', 'xyz_to_rgb']
