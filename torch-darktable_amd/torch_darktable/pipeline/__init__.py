"""Camera -> uint8 pipeline built on the kernel ops (reference torch_darktable/pipeline/)."""

from .camera_settings import CameraSettings, load_camera_settings_from_dir, load_raw_bayer, load_raw_bytes, settings_for_file
from .config import Debayer, ImageProcessingSettings, ToneMapper
from .image_processor import ImageProcessor, ImageSizeMismatchError
from .presets import get_preset, presets
from .streaming import RawFrameStream
from .transform import ImageTransform, transform, transformed_size

__all__ = ['CameraSettings', 'Debayer', 'ImageProcessingSettings', 'ImageProcessor', 'ImageSizeMismatchError', 'ImageTransform', 'RawFrameStream',
           'ToneMapper', 'get_preset', 'load_camera_settings_from_dir', 'load_raw_bayer', 'load_raw_bytes', 'presets',
           'settings_for_file', 'transform', 'transformed_size']
