"""Tonemap operators and image statistics (reference torch_darktable/tonemap.py)."""

from __future__ import annotations

from dataclasses import dataclass

import torch

from .extension import extension


@dataclass(frozen=True)
class TonemapParameters:
    """gamma, exposure (`intensity`, in e-stops for the adaptive operators / stops for plain
    ACES), local-vs-global adaptation blend and Lab vibrance."""

    gamma: float = 1.0
    intensity: float = 0.0
    light_adapt: float = 0.8
    vibrance: float = 0.0

    def to_cpp(self) -> 'extension.TonemapParams':
        return extension.TonemapParams(self.gamma, self.intensity, self.light_adapt, self.vibrance)

    @classmethod
    def from_cpp(cls, cpp_params) -> 'TonemapParameters':
        return cls(cpp_params.gamma, cpp_params.intensity, cpp_params.light_adapt, cpp_params.vibrance)


def metrics_to_dict(metrics: torch.Tensor) -> dict:
    assert metrics.numel() == 5, f'Expected 5 elements, got {metrics.numel()}'
    m = [float(v) for v in metrics.detach().cpu().tolist()]
    return {'log_mean': m[0], 'linear_mean': m[1], 'rgb_mean': (m[2], m[3], m[4])}


def metrics_from_dict(metrics_dict: dict, device: torch.device = torch.device('cuda')) -> torch.Tensor:
    rgb = metrics_dict['rgb_mean']
    assert isinstance(rgb, tuple), 'RGB mean must be a tuple'
    return torch.tensor([metrics_dict['log_mean'], metrics_dict['linear_mean'], *rgb], device=device, dtype=torch.float32)


def print_metrics(metrics: torch.Tensor):
    d = metrics_to_dict(metrics)
    r, g, b = d['rgb_mean']
    print('Image Metrics:')
    print(f'  Log Mean: {d["log_mean"]:.4f}')
    print(f'  Linear Mean: {d["linear_mean"]:.4f}')
    print(f'  RGB Mean: ({r:.4f}, {g:.4f}, {b:.4f})')


def _check_image(image: torch.Tensor) -> None:
    assert image.dim() == 3 and image.size(2) == 3, 'Input must be (H, W, 3)'
    assert image.dtype in (torch.float32, torch.float16), 'Input must be float32'
    assert image.device.type == 'cuda', 'Input must be on CUDA device'


def _check_metrics(metrics: torch.Tensor) -> None:
    assert metrics.numel() == 5 and metrics.dtype == torch.float32 and metrics.device.type == 'cuda'


def reinhard_tonemap(image: torch.Tensor, metrics: torch.Tensor, params: TonemapParameters) -> torch.Tensor:
    """Reinhard operator -> uint8 (H, W, 3)."""
    _check_image(image)
    assert metrics.numel() == 5, 'Metrics tensor must have 5 elements'
    return extension.reinhard_tonemap(image, metrics, params.to_cpp())


def aces_tonemap(image: torch.Tensor, params: TonemapParameters, metrics: torch.Tensor | None = None) -> torch.Tensor:
    """ACES fit; with `metrics` the scene-adaptive variant."""
    _check_image(image)
    if metrics is None:
        return extension.aces_tonemap(image, params.to_cpp())
    _check_metrics(metrics)
    return extension.adaptive_aces_tonemap(image, metrics, params.to_cpp())


def linear_tonemap(image: torch.Tensor, metrics: torch.Tensor, params: TonemapParameters) -> torch.Tensor:
    """Divide by the adaptation level, gamma, clamp -> uint8 (H, W, 3)."""
    _check_image(image)
    _check_metrics(metrics)
    return extension.linear_tonemap(image, metrics, params.to_cpp())


compute_image_bounds = extension.compute_image_bounds
MetricsAccumulator = extension.MetricsAccumulator  # compute_image_metrics in two halves, so producers can feed it (see its docstring)


def compute_image_metrics(images: list, stride: int = 8, min_gray: float = 1e-4, rescale: bool = False) -> torch.Tensor:
    return extension.compute_image_metrics(images, stride, min_gray, rescale)


__all__ = ['MetricsAccumulator', 'TonemapParameters', 'aces_tonemap', 'compute_image_bounds', 'compute_image_metrics', 'linear_tonemap', 'metrics_from_dict',
           'metrics_to_dict', 'print_metrics', 'reinhard_tonemap']
