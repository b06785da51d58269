"""Output orientation of a processed frame (reference torch_darktable/pipeline/transform.py)."""

from __future__ import annotations

from enum import Enum

import torch


class ImageTransform(Enum):
    none = 0
    rotate_90 = 1
    rotate_180 = 2
    rotate_270 = 3
    transpose = 4
    flip_horiz = 5
    flip_vert = 6
    transverse = 7

    def next_rotation(self) -> 'ImageTransform':
        """Cycle within the rotation group (none -> 90 -> 180 -> 270) or the reflection group."""
        rotations = [ImageTransform.none, ImageTransform.rotate_90, ImageTransform.rotate_180, ImageTransform.rotate_270]
        reflections = [ImageTransform.transpose, ImageTransform.flip_horiz, ImageTransform.flip_vert, ImageTransform.transverse]
        for cycle in (rotations, reflections):
            if self in cycle:
                return cycle[(cycle.index(self) + 1) % 4]
        return ImageTransform.rotate_90


_SWAPS_AXES = {ImageTransform.rotate_90, ImageTransform.rotate_270, ImageTransform.transpose}


def transformed_size(original_size: tuple[int, int], transform: ImageTransform) -> tuple[int, int]:
    w, h = original_size
    return (h, w) if transform in _SWAPS_AXES else (w, h)


def transform(image: torch.Tensor, transform: ImageTransform) -> torch.Tensor:
    """Apply the orientation to an (H, W, ...) tensor; always returns a contiguous tensor."""
    if transform is ImageTransform.none:
        return image
    rot = {ImageTransform.rotate_90: 1, ImageTransform.rotate_180: 2, ImageTransform.rotate_270: 3}
    if transform in rot:
        return torch.rot90(image, rot[transform], (0, 1)).contiguous()
    if transform is ImageTransform.flip_horiz:
        return torch.flip(image, (1,)).contiguous()
    if transform is ImageTransform.flip_vert:
        return torch.flip(image, (0,)).contiguous()
    if transform is ImageTransform.transverse:
        return torch.flip(image, (0, 1)).contiguous()
    return torch.transpose(image, 0, 1).contiguous()
