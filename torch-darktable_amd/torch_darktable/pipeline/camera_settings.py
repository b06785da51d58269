"""Per-camera raw format + processing settings, and loading packed 12-bit raw files
(reference torch_darktable/pipeline/camera_settings.py).

On-disk raw frame: width*height 12-bit samples packed 2 px -> 3 bytes (standard or IDS layout,
see csrc/codec.hip) followed by `padding` trailing bytes; a file is matched to a camera by its
directory name or, failing that, by its exact size."""

from __future__ import annotations

from pathlib import Path
import threading
from typing import Annotated, Literal
import warnings

from pydantic import BaseModel
import torch

from ..bayer import BayerPattern, PackedFormat
from ..debayer import decode12
from .config import EnumValidator, ImageProcessingSettings
from .transform import ImageTransform

warnings.filterwarnings('ignore', category=UserWarning, message='The given buffer is not writable')


class CameraSettings(BaseModel, frozen=True):
    type: Literal['camera_settings'] = 'camera_settings'

    name: str
    image_size: tuple[int, int]  # (width, height)
    padding: int = 0             # trailing bytes after the packed samples

    bayer_pattern: Annotated[BayerPattern, EnumValidator(BayerPattern, 'Bayer pattern')] = BayerPattern.RGGB
    packed_format: Annotated[PackedFormat, EnumValidator(PackedFormat, 'Packed format')] = PackedFormat.Packed12
    white_balance: tuple[float, float, float] | None = None
    image_processing: ImageProcessingSettings

    transform: Annotated[ImageTransform | dict[str, ImageTransform], EnumValidator(ImageTransform, 'Image transform')] = ImageTransform.none

    def get_image_transform(self, camera_name: str) -> ImageTransform:
        if isinstance(self.transform, dict):
            return self.transform.get(camera_name, ImageTransform.none)
        return self.transform

    @property
    def bytes(self) -> int:
        """Size of one raw file of this camera."""
        w, h = self.image_size
        return (w * h * 3) // 2 + self.padding

    def save_json(self, path: Path) -> None:
        Path(path).write_text(self.model_dump_json(indent=2))

    @classmethod
    def load_json(cls, path: Path) -> 'CameraSettings':
        return cls.model_validate_json(Path(path).read_text())


_staging_lock = threading.Lock()
_staging: dict = {}  # (device index, nbytes) -> [pinned host buffer, event of the last upload out of it]


def load_raw_bytes(filepath: Path, device: torch.device = torch.device('cuda:0')) -> torch.Tensor:
    """Whole file -> uint8 device tensor (no decoding).  The file is read straight into a cached
    pinned staging buffer (pinning 19 MB per call costs more than the upload) and copied
    asynchronously on the current stream; the buffer is reused once its previous upload has finished.
    For sustained throughput use pipeline.RawFrameStream, which also overlaps the reads."""
    path = Path(filepath)
    device = torch.device(device)
    if device.type != 'cuda':
        return torch.frombuffer(bytearray(path.read_bytes()), dtype=torch.uint8)
    nbytes = path.stat().st_size
    with _staging_lock, torch.cuda.device(device):
        key = (torch.cuda.current_device(), nbytes)
        if key not in _staging:
            if len(_staging) >= 8:  # a handful of camera formats at most; do not hoard pinned memory
                _staging.pop(next(iter(_staging)))
            _staging[key] = [torch.empty(nbytes, dtype=torch.uint8).pin_memory(), None]
        host, last = _staging[key]
        if last is not None:
            last.synchronize()
        with open(path, 'rb') as f:
            if f.readinto(host.numpy()) != nbytes:
                raise OSError(f'{path}: short read')
        out = host.to(device, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        _staging[key][1] = done
    return out


def load_raw_bytes_stripped(filepath: Path, camera_settings: CameraSettings, device: torch.device = torch.device('cuda:0')) -> torch.Tensor:
    raw = load_raw_bytes(filepath, device)
    return raw[: -camera_settings.padding] if camera_settings.padding > 0 else raw


def load_raw_bayer(filepath: Path, camera_settings: CameraSettings | None = None, device: torch.device = torch.device('cuda:0')) -> torch.Tensor:
    """Raw file -> (H, W) float32 mosaic in [0, 1]."""
    if camera_settings is None:
        camera_settings = settings_for_file(Path(filepath))
    width, _ = camera_settings.image_size
    packed = load_raw_bytes_stripped(filepath, camera_settings, device)
    return decode12(packed, output_dtype=torch.float32, format_type=camera_settings.packed_format).view(-1, width)


def get_camera_settings_dir() -> Path:
    return Path(__file__).parent.parent / 'camera_settings'


def load_camera_settings_from_dir(settings_dir: Path | None = None) -> dict[str, CameraSettings]:
    settings_dir = Path(settings_dir) if settings_dir is not None else get_camera_settings_dir()
    loaded = (CameraSettings.load_json(p) for p in sorted(settings_dir.glob('*.json')))
    return {cs.name: cs for cs in loaded}


def settings_for_file(file_path: Path, settings_dir: Path | None = None) -> CameraSettings:
    """Camera of a raw file: its directory name if that is a known camera, else a unique file-size match."""
    known = load_camera_settings_from_dir(settings_dir)
    camera_name = Path(file_path).parent.stem
    if camera_name in known:
        return known[camera_name]
    size = Path(file_path).stat().st_size
    for cs in known.values():
        if cs.bytes == size:
            return cs
    raise ValueError(
        f'Could not find camera settings for "{file_path}". Directory name "{camera_name}" not recognized and file size {size} bytes '
        f'does not match any known camera. Available cameras: {list(known.keys())}')


def validate_camera_names(settings: CameraSettings, camera_names: list[str]) -> None:
    if isinstance(settings.transform, dict):
        expected, actual = set(settings.transform.keys()), set(camera_names)
        if expected != actual:
            raise ValueError(f'Camera names mismatch: settings expects {sorted(expected)}, got {sorted(actual)}')
