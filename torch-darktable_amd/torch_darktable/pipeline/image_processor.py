"""packed raw bytes -> uint8 RGB: decode -> [white balance] -> demosaic -> [post-process] ->
normalise -> [Wiener log-L] -> [bilateral] -> metrics -> tonemap -> orientation
(reference torch_darktable/pipeline/image_processor.py).

Everything between the byte upload and the uint8 result stays on the device: bounds / metrics
and their moving averages are device tensors, no `.item()` anywhere, so one process can keep
its GPU busy and several processes (one per GPU) scale without host-side serialisation."""

from __future__ import annotations

import torch

from .. import debayer as _debayer
from .. import tonemap as _tonemap
from ..bayer import BayerPattern, PackedFormat
from ..denoise import Wiener
from ..local_contrast import Bilateral
from ..white_balance import apply_white_balance
from .camera_settings import CameraSettings
from .config import Debayer, ImageProcessingSettings, ToneMapper
from .transform import ImageTransform, transform
from .util import lerp, normalize_image, resize_longest_edge


class ImageSizeMismatchError(Exception):
    """A raw buffer does not have the byte / pixel count the processor was built for."""

    def __init__(self, message: str, image_size: tuple[int, int], packed_format: PackedFormat, padding: int):
        super().__init__(message)
        self.image_size = image_size
        self.packed_format = packed_format
        self.padding = padding


class ImageProcessor:
    def __init__(self, image_size: tuple[int, int], bayer_pattern: BayerPattern, packed_format: PackedFormat,
                 settings: ImageProcessingSettings, device: torch.device, white_balance: tuple[float, float, float] | None,
                 transforms: ImageTransform | dict[str, ImageTransform] = ImageTransform.none, padding: int = 0,
                 storage_dtype: torch.dtype = torch.float32):
        assert device.index is not None, f'Device not fully specified: {device}'
        self.device = device
        self.settings = settings
        self.image_size = image_size
        self.bayer_pattern = bayer_pattern
        self.packed_format = packed_format
        self.transforms = transforms
        self.padding = padding
        # image storage between the stages: float32 (the reference) or float16 (fp32 arithmetic, half the HBM traffic)
        assert storage_dtype in (torch.float32, torch.float16)
        self.storage_dtype = storage_dtype
        self._lum_plane: torch.Tensor | None = None  # lightness plane handed from the denoiser to the bilateral stage
        self._ab_plane: torch.Tensor | None = None   # ... and the chroma (a, b) plane of the Lab hand-over
        self.metrics: torch.Tensor | None = None  # moving averages, device-resident
        self.bounds: torch.Tensor | None = None
        self.rcd_workspace = _debayer.RCD(device, image_size, bayer_pattern)
        self.wiener_workspace = Wiener(device, image_size)
        self._build_tunable_workspaces(settings)
        self.white_balance = torch.tensor(white_balance, device=device, dtype=torch.float32) if white_balance is not None else None

    def _build_tunable_workspaces(self, s: ImageProcessingSettings, which=('bilateral', 'ppg', 'postprocess')) -> None:
        if 'bilateral' in which:
            self.bil_workspace = Bilateral(self.device, self.image_size, sigma_s=s.bil_sigma_spatial, sigma_r=s.bil_sigma_luminance)
        if 'ppg' in which:
            self.ppg_workspace = _debayer.PPG(self.device, self.image_size, self.bayer_pattern, median_threshold=s.ppg_median_threshold)
        if 'postprocess' in which:
            self.postprocess_workspace = _debayer.PostProcess(
                self.device, self.image_size, self.bayer_pattern, color_smoothing_passes=s.color_smoothing_passes,
                green_eq_local=False, green_eq_global=True, green_eq_threshold=s.green_eq_threshold)

    def __repr__(self) -> str:
        wb = 'None' if self.white_balance is None else '({:.3f}, {:.3f}, {:.3f})'.format(*self.white_balance.tolist())
        tf = self.transforms.name if isinstance(self.transforms, ImageTransform) else '{' + ', '.join(f'{k}: {v.name}' for k, v in self.transforms.items()) + '}'
        return (f'ImageProcessor(size={self.image_size}, bayer={self.bayer_pattern.name}, format={self.packed_format.name}, device={self.device}, '
                f'wb={wb}, padding={self.padding}, transform={tf}, debayer={self.settings.debayer.name}, tonemap={self.settings.tone_mapping.name})')

    @staticmethod
    def from_camera_settings(camera_settings: CameraSettings, device: torch.device, storage_dtype: torch.dtype = torch.float32) -> 'ImageProcessor':
        return ImageProcessor(camera_settings.image_size, camera_settings.bayer_pattern, camera_settings.packed_format,
                              camera_settings.image_processing, device=device, white_balance=camera_settings.white_balance,
                              transforms=camera_settings.transform, padding=camera_settings.padding, storage_dtype=storage_dtype)

    def update_settings(self, settings: ImageProcessingSettings) -> None:
        """Swap settings; only workspaces whose parameters changed are rebuilt."""
        old, self.settings = self.settings, settings

        def changed(*names: str) -> bool:
            return any(getattr(old, n) != getattr(settings, n) for n in names)

        stale = []
        if changed('bil_sigma_spatial', 'enable_bilateral', 'bil_sigma_luminance'):
            stale.append('bilateral')
        if changed('ppg_median_threshold'):
            stale.append('ppg')
        if changed('color_smoothing_passes', 'green_eq_threshold'):
            stale.append('postprocess')
        self._build_tunable_workspaces(settings, tuple(stale))

    @property
    def final_size(self) -> tuple[int, int]:
        return resize_longest_edge(self.image_size, self.settings.resize_width)

    @property
    def expected_bytes(self) -> int:
        w, h = self.image_size
        if self.packed_format not in (PackedFormat.Packed12, PackedFormat.Packed12_IDS):
            raise ValueError(f'Unsupported packed format: {self.packed_format}')
        return (w * h * 3) // 2 + self.padding

    def _mismatch(self, message: str) -> ImageSizeMismatchError:
        return ImageSizeMismatchError(message, image_size=self.image_size, packed_format=self.packed_format, padding=self.padding)

    # ---- stages
    def load_bytes(self, bytes: torch.Tensor) -> torch.Tensor:
        """Packed raw bytes (+ trailing padding) -> (H, W) float32 mosaic."""
        if bytes.numel() != self.expected_bytes:
            raise self._mismatch(f'Image size mismatch: expected {self.expected_bytes} bytes for {self.image_size} {self.packed_format.name} '
                                 f'with {self.padding} padding, got {bytes.numel()} bytes. ')
        if self.padding > 0:
            bytes = bytes[: -self.padding]
        decoded = _debayer.decode12(bytes, output_dtype=torch.float32, format_type=self.packed_format)
        w, h = self.image_size
        if decoded.numel() != w * h:
            raise self._mismatch(f'Decoded image size mismatch: expected {w * h} pixels ({w}x{h}), got {decoded.numel()} pixels.')
        return decoded.view(h, w)

    def load_image(self, bytes: torch.Tensor) -> torch.Tensor:
        if self.settings.debayer == Debayer.rcd:
            # decode -> white balance -> RCD as one kernel (same result as load_bytes + debayer, two fp32 planes less)
            if bytes.numel() != self.expected_bytes:
                raise self._mismatch(f'Image size mismatch: expected {self.expected_bytes} bytes for {self.image_size} {self.packed_format.name} '
                                     f'with {self.padding} padding, got {bytes.numel()} bytes. ')
            payload = bytes[: bytes.numel() - self.padding] if self.padding > 0 else bytes
            rgb = self.rcd_workspace.process_packed(payload, self.white_balance, self.packed_format, self.storage_dtype)
            # (PostProcess takes the storage type as it is: fp32 between its stages, one rounding at its store -- the same bits as
            # converting to float32 around it, without the two conversion passes)
            return self.postprocess_workspace.process(rgb) if self.settings.postprocess else rgb
        return self.debayer(self.load_bytes(bytes)).to(self.storage_dtype)

    def debayer(self, bayer_image: torch.Tensor) -> torch.Tensor:
        assert bayer_image.ndim == 2, f'Bayer image must have 2 dimensions, got {bayer_image.shape}'
        if self.white_balance is not None:
            bayer_image = apply_white_balance(bayer_image, self.white_balance, self.bayer_pattern)
        mosaic = bayer_image.unsqueeze(-1)
        method = self.settings.debayer
        if method == Debayer.bilinear:
            rgb = _debayer.bilinear5x5_demosaic(mosaic, self.bayer_pattern)
        elif method == Debayer.rcd:
            rgb = self.rcd_workspace.process(mosaic)
        elif method == Debayer.ppg:
            rgb = self.ppg_workspace.process(mosaic)
        else:
            raise AssertionError(f'Invalid debayer method: {method}')
        return self.postprocess_workspace.process(rgb) if self.settings.postprocess else rgb

    def process_rgb(self, rgb_raw: torch.Tensor, bounds: torch.Tensor | None = None, metrics: '_tonemap.MetricsAccumulator | None' = None) -> torch.Tensor:
        """normalise -> [denoise] -> [local contrast].  When both stages run, the denoiser hands the lightness plane of
        its result to the bilateral (which would extract it first); with `metrics` the result is also added to that
        accumulator -- same results as the separate calls."""
        s = self.settings
        both = s.enable_denoise and s.enable_bilateral
        if bounds is not None and not both:
            rgb_raw = normalize_image(rgb_raw, bounds)
        if both:
            # both stages replace the Lab lightness of the same pixel (reference denoise.py:54-58, local_contrast.py:109-114): the
            # pixel travels between them as lightness + chroma planes and is converted back to RGB once (include/tdk_hip.h, Lab
            # hand-over; tests/test_gpu_lab_chain.py) instead of as an RGB image through two colour round trips
            hw = (self.image_size[1], self.image_size[0])
            if self._lum_plane is None or self._lum_plane.device != rgb_raw.device:
                self._lum_plane = torch.empty(hw, dtype=torch.float32, device=rgb_raw.device)
                self._ab_plane = torch.empty((*hw, 2), dtype=torch.float32, device=rgb_raw.device)
            # (normalize_image rides in the first kernel of the chain: the normalised image is never stored)
            self.wiener_workspace.process_log_luminance_lab(rgb_raw, s.denoise, luminance_out=self._lum_plane, chroma_out=self._ab_plane, bounds=bounds)
            return self.bil_workspace.process_lab(self._lum_plane, self._ab_plane, s.bilateral, out_dtype=rgb_raw.dtype, metrics=metrics)
        if s.enable_denoise:
            rgb_raw = self.wiener_workspace.process_log_luminance(rgb_raw, s.denoise)
        if s.enable_bilateral:
            rgb_raw = self.bil_workspace.process_rgb(rgb_raw, s.bilateral, metrics=metrics)
        elif metrics is not None:
            metrics.add(rgb_raw)
        return rgb_raw

    def tonemap(self, rgb_raw: torch.Tensor, metrics: torch.Tensor | None = None) -> torch.Tensor:
        s = self.settings
        params = _tonemap.TonemapParameters(s.tone_gamma, s.tone_intensity, s.light_adapt, s.vibrance)
        if metrics is None:
            metrics = _tonemap.compute_image_metrics([rgb_raw], stride=4, min_gray=1e-4)
        mapper = s.tone_mapping
        if mapper == ToneMapper.reinhard:
            return _tonemap.reinhard_tonemap(rgb_raw, metrics, params)
        if mapper == ToneMapper.linear:
            return _tonemap.linear_tonemap(rgb_raw, metrics, params)
        if mapper == ToneMapper.aces:
            return _tonemap.aces_tonemap(rgb_raw, params)
        return _tonemap.aces_tonemap(rgb_raw, params, metrics)

    def transform(self, image: torch.Tensor, image_name: str) -> torch.Tensor:
        t = self.transforms[image_name] if isinstance(self.transforms, dict) else self.transforms
        return transform(image, t)

    # ---- whole pipeline
    def process(self, bytes: torch.Tensor, image_name: str) -> torch.Tensor:
        return self.process_image_set({image_name: bytes})[image_name]

    def process_image_set(self, image_set_bytes: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        """One synchronised set of cameras: shared bounds / metrics (moving-averaged across calls)."""
        names = list(image_set_bytes.keys())
        ema = self.settings.moving_average
        rgb = [self.load_image(b) for b in image_set_bytes.values()]
        bounds = _tonemap.compute_image_bounds(rgb, stride=8)
        self.bounds = lerp(self.bounds if self.bounds is not None else bounds, bounds, ema)
        acc = _tonemap.MetricsAccumulator(self.device, stride=8)  # == compute_image_metrics(rgb, stride=8), fed by the last stage
        rgb = [self.process_rgb(img, self.bounds, acc) for img in rgb]
        metrics = acc.finish()
        self.metrics = lerp(self.metrics if self.metrics is not None else metrics, metrics, ema)
        mapped = [self.tonemap(img, self.metrics) for img in rgb]
        return {name: self.transform(img, name) for name, img in zip(names, mapped)}
