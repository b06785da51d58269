"""Seeded synthetic RAW frames for tests and benchmarks (no datasets are available offline).

Scene = smooth low-frequency colour field in [0.02, 0.9] (sum of random 2-D sinusoids per
channel) + step edges (random rectangles) + Gaussian noise sigma = 0.02, clamped to [0, 1] and
sampled on the CFA defined by the kernels' fc() function.  Structured on purpose: pure noise
turns every gradient test of PPG / RCD into a coin flip."""

from __future__ import annotations

import math

import torch

_PATTERN_WORDS = {'RGGB': 0x94949494, 'BGGR': 0x16161616, 'GRBG': 0x61616161, 'GBRG': 0x49494949}


def synthetic_rgb(height: int, width: int, seed: int = 1234, device='cpu', noise_sigma: float = 0.02, rectangles: int = 12) -> torch.Tensor:
    """(H, W, 3) float32 scene, deterministic for (size, seed) on a given device type."""
    dev = torch.device(device)
    gen = torch.Generator(device='cpu').manual_seed(seed)  # scene parameters always come from the CPU generator
    yy = torch.linspace(0.0, 1.0, height, device=dev).view(height, 1)
    xx = torch.linspace(0.0, 1.0, width, device=dev).view(1, width)
    img = torch.empty(height, width, 3, device=dev)
    for c in range(3):
        acc = torch.zeros(height, width, device=dev)
        for _ in range(4):
            fx, fy = (torch.rand(2, generator=gen) * 6.0 + 0.5).tolist()
            ph = float(torch.rand(1, generator=gen)) * 2 * math.pi
            acc += torch.sin(2 * math.pi * (fx * xx + fy * yy) + ph)
        img[:, :, c] = 0.46 + 0.11 * acc  # |acc| <= 4 -> [0.02, 0.9]
    for _ in range(rectangles):
        x0, y0, wd, ht = torch.rand(4, generator=gen).tolist()
        rx0, ry0 = int(x0 * width * 0.9), int(y0 * height * 0.9)
        rx1, ry1 = min(width, rx0 + max(4, int(wd * width * 0.3))), min(height, ry0 + max(4, int(ht * height * 0.3)))
        colour = (torch.rand(3, generator=gen) * 0.8 + 0.05).to(dev)
        img[ry0:ry1, rx0:rx1, :] = colour
    if noise_sigma > 0:
        ngen = torch.Generator(device=dev).manual_seed(seed + 7919)
        img += torch.randn(img.shape, generator=ngen, device=dev) * noise_sigma
    return img.clamp_(0.0, 1.0)


def mosaic(rgb: torch.Tensor, pattern: str = 'RGGB') -> torch.Tensor:
    """(H, W, 3) -> (H, W, 1): keep at each site the channel fc(row, col, pattern) selects."""
    word = _PATTERN_WORDS[pattern]
    h, w, _ = rgb.shape
    rows = torch.arange(h, device=rgb.device).view(h, 1)
    cols = torch.arange(w, device=rgb.device).view(1, w)
    shift = ((((rows << 1) & 14) + (cols & 1)) << 1)
    ch = (torch.tensor(word, device=rgb.device, dtype=torch.int64) >> shift) & 3
    return torch.gather(rgb, 2, ch.unsqueeze(-1))


def synthetic_bayer(height: int, width: int, seed: int = 1234, device='cpu', pattern: str = 'RGGB', noise_sigma: float = 0.02) -> torch.Tensor:
    return mosaic(synthetic_rgb(height, width, seed, device, noise_sigma), pattern).contiguous()
