"""Local-contrast workspaces: local Laplacian and bilateral grid (reference torch_darktable/local_contrast.py)."""

from __future__ import annotations

from dataclasses import dataclass

import torch

from .extension import extension


@dataclass
class LaplacianParams:
    """Local Laplacian settings (only num_gamma == 6 is supported, as in the reference)."""

    num_gamma: int = 6
    sigma: float = 0.2
    shadows: float = 1.0
    highlights: float = 1.0
    clarity: float = 0.0


def _expect_plane(t: torch.Tensor, ws, who: str) -> None:
    expected = (ws.height, ws.width)
    if tuple(t.shape) != expected:
        raise RuntimeError(f'{who} input shape {tuple(t.shape)} != expected {expected}')


class Laplacian:
    def __init__(self, device: torch.device, image_size: tuple[int, int], params: LaplacianParams):
        w, h = image_size
        self._laplacian = extension.Laplacian(device, w, h, params.num_gamma, params.sigma, params.shadows, params.highlights, params.clarity)

    def process(self, input_tensor: torch.Tensor) -> torch.Tensor:
        _expect_plane(input_tensor, self._laplacian, 'Laplacian')
        return self._laplacian.process(input_tensor)

    def process_rgb(self, input_image: torch.Tensor) -> torch.Tensor:
        """Filter the Lab lightness of an RGB image and put it back."""
        lum = extension.compute_luminance(input_image)
        return extension.modify_luminance(input_image, self.process(lum))

    @property
    def image_size(self) -> tuple[int, int]:
        return (self._laplacian.width, self._laplacian.height)

    @property
    def sigma(self) -> float:
        return self._laplacian.sigma

    @property
    def shadows(self) -> float:
        return self._laplacian.shadows

    @property
    def highlights(self) -> float:
        return self._laplacian.highlights

    @property
    def clarity(self) -> float:
        return self._laplacian.clarity


class Bilateral:
    def __init__(self, device: torch.device, image_size: tuple[int, int], *, sigma_s: float, sigma_r: float):
        w, h = image_size
        self._bilateral = extension.Bilateral(device, w, h, float(sigma_s), float(sigma_r))

    def process(self, luminance: torch.Tensor, detail: float) -> torch.Tensor:
        _expect_plane(luminance, self._bilateral, 'Bilateral')
        return self._bilateral.process(luminance, detail)

    def process_rgb(self, input_image: torch.Tensor, detail: float, *, luminance: torch.Tensor | None = None, metrics=None) -> torch.Tensor:
        """extract -> filter -> replace as one library call.  Optional hand-overs from / to the neighbouring pipeline
        stages (results unchanged): `luminance` = compute_luminance(input_image) if the producer already has it,
        `metrics` = a tonemap.MetricsAccumulator that collects compute_image_metrics of the result."""
        assert input_image.dim() == 3, f'image must have 3 dimensions, got {input_image.shape}'
        return self._bilateral.process_rgb(input_image, float(detail), luminance, metrics)

    def process_lab(self, luminance: torch.Tensor, chroma: torch.Tensor, detail: float, *, out_dtype: torch.dtype = torch.float32, metrics=None) -> torch.Tensor:
        """process_rgb for pixels handed over as Lab by denoise.Wiener.process_log_luminance_lab: float32 (H, W) lightness +
        float32 (H, W, 2) chroma in, (H, W, 3) RGB of `out_dtype` out."""
        return self._bilateral.process_lab(luminance, chroma, float(detail), out_dtype, metrics)

    def process_log_rgb(self, input_image: torch.Tensor, detail: float, eps: float = 1e-6, *, luminance: torch.Tensor | None = None, metrics=None) -> torch.Tensor:
        return self._bilateral.process_log_rgb(input_image, float(detail), eps, luminance, metrics)

    @property
    def image_size(self) -> tuple[int, int]:
        return (self._bilateral.width, self._bilateral.height)

    @property
    def sigma_s(self) -> float:
        return self._bilateral.sigma_s

    @property
    def sigma_r(self) -> float:
        return self._bilateral.sigma_r


__all__ = ['Bilateral', 'Laplacian', 'LaplacianParams']
