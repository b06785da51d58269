"""Wiener denoiser front-end and MAD noise estimate (reference torch_darktable/denoise.py)."""

from __future__ import annotations

import torch

from .extension import extension


def check_overlap_factor(overlap_factor: int):
    if overlap_factor not in {2, 4, 8}:
        raise ValueError('overlap_factor must be 2, 4, or 8')


class Wiener:
    """Tiled-FFT Wiener shrinkage for a fixed image size; noise as a float or per-channel tensor."""

    def __init__(self, device: torch.device, image_size: tuple[int, int], overlap_factor: int = 4, tile_size: int = 32):
        width, height = image_size
        if device.type != 'cuda':
            raise ValueError(f'Device must be CUDA, got {device}')
        if width <= 0 or height <= 0:
            raise ValueError(f'Image dimensions must be positive, got {width}x{height}')
        check_overlap_factor(overlap_factor)
        if tile_size not in {16, 32}:
            raise ValueError(f'tile_size must be 16 or 32, got {tile_size}')
        try:
            self._wiener = extension.Wiener(device, width, height, overlap_factor, tile_size)
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(f'Failed to create Wiener extension: {e}') from e
        self._tile_size = tile_size
        self._device = device

    def __repr__(self):
        return f'Wiener({self._wiener.width}x{self._wiener.height},overlap_factor={self.overlap_factor}, tile_size={self._tile_size})'

    @property
    def overlap_factor(self) -> int:
        return self._wiener.overlap_factor

    def _sigmas(self, noise, channels: int) -> torch.Tensor:
        if isinstance(noise, float):
            # one small device tensor per (value, channels), kept: a fill launch per call is a latency-bound launch per frame
            cache = self.__dict__.setdefault('_sigma_cache', {})
            key = (noise, channels)
            t = cache.get(key)
            if t is None:
                if len(cache) >= 16:
                    cache.clear()
                # built on the host and copied from pageable memory: the copy has completed when .to() returns, so the tensor is
                # valid on EVERY stream that later reads it (a torch.full would be a fill queued on the stream current now)
                t = cache[key] = torch.tensor([noise] * channels, dtype=torch.float32).to(self._device)
            return t
        if isinstance(noise, torch.Tensor):
            if noise.shape != (channels,):
                raise ValueError(f'noise tensor must have {channels} elements for {channels}-channel image')
            return noise.to(dtype=torch.float32, device=self._device)
        raise ValueError(f'noise must be float, or Tensor[{channels}]')

    def process(self, image: torch.Tensor, noise) -> torch.Tensor:
        """Denoise an (H, W, C) image, C in {1, 3}."""
        assert image.dim() == 3, f'image must have 3 dimensions, got {image.shape}'
        expected = (self._wiener.height, self._wiener.width, image.size(2))
        if tuple(image.shape) != expected:
            raise RuntimeError(f'Wiener input shape {tuple(image.shape)} != expected {expected}')
        channels = image.size(2)
        if channels not in {1, 3}:
            raise ValueError(f'image channels must be 1 or 3, got {channels}')
        return self._wiener.process(image, self._sigmas(noise, channels))

    def process_luminance(self, image: torch.Tensor, noise) -> torch.Tensor:
        lum = extension.compute_luminance(image)
        return extension.modify_luminance(image, self.process(lum.unsqueeze(2), noise).squeeze(2))

    def process_log_luminance(self, image: torch.Tensor, noise, eps: float = 1e-4, *, luminance_out: torch.Tensor | None = None) -> torch.Tensor:
        """Denoise the log-lightness of an RGB image (extract -> Wiener -> replace), fused in one library call.
        luminance_out: optional float32 (H, W) tensor that also receives compute_luminance(result), for
        Bilateral.process_rgb(..., luminance=...) as the next stage."""
        expected = (self._wiener.height, self._wiener.width, 3)
        if tuple(image.shape) != expected:
            raise RuntimeError(f'Wiener input shape {tuple(image.shape)} != expected {expected}')
        return self._wiener.process_log_luminance(image, self._sigmas(noise, 1), eps, luminance_out)

    def process_log_luminance_lab(self, image: torch.Tensor, noise, eps: float = 1e-4, *, luminance_out: torch.Tensor | None = None,
                                  chroma_out: torch.Tensor | None = None, bounds: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """process_log_luminance for a consumer that takes the pixel as Lab (local_contrast.Bilateral.process_lab): returns
        (luminance, chroma) = the float32 (H, W) lightness plane of the denoised image and the float32 (H, W, 2) plane of its
        Lab (a, b), without forming the denoised RGB image.  The two stages then share ONE colour round trip; results agree
        with process_log_luminance -> process_rgb within the colour operators' tolerance.  `bounds`: normalise the image as
        pipeline.util.normalize_image(image, bounds) would, while it is read (the pipeline's step before the denoiser)."""
        expected = (self._wiener.height, self._wiener.width, 3)
        if tuple(image.shape) != expected:
            raise RuntimeError(f'Wiener input shape {tuple(image.shape)} != expected {expected}')
        return self._wiener.process_log_luminance_lab(image, self._sigmas(noise, 1), eps, luminance_out, chroma_out, bounds)

    def process_log(self, image: torch.Tensor, noise, eps: float = 1e-4) -> torch.Tensor:
        return self.process((image + eps).log(), noise).exp()


def create_wiener(device: torch.device, image_size: tuple[int, int], *, overlap: int = 4, tile_size: int = 32) -> Wiener:
    return Wiener(device, image_size, overlap_factor=overlap, tile_size=tile_size)


def estimate_channel_noise(image: torch.Tensor, stride: int = 8) -> torch.Tensor:
    """Per-channel noise sigma of an (H, W, 3) image: MAD of the 3x3 Laplacian response on a
    stride-subsampled grid, divided by 0.6745 (pure torch, as in the reference)."""
    kernel = torch.tensor([[0, -1, 0], [-1, 4, -1], [0, -1, 0]], dtype=image.dtype, device=image.device)
    weight = kernel.expand(3, 1, 3, 3).contiguous()
    response = torch.conv2d(image.permute(2, 0, 1).unsqueeze(0), weight, groups=3, padding=1)[0]
    samples = response[:, ::stride, ::stride].flatten(1)
    med = samples.median(dim=1).values
    mad = (samples - med.unsqueeze(1)).abs().median(dim=1).values
    return mad / 0.6745
