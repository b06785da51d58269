"""Named ImageProcessingSettings (reference torch_darktable/pipeline/presets.py:16-53)."""

from __future__ import annotations

from .config import ImageProcessingSettings, ToneMapper

_COMMON = dict(enable_denoise=True, enable_bilateral=True, postprocess=True, vibrance=0.5)

presets: dict[str, ImageProcessingSettings] = {
    'aces': ImageProcessingSettings(**_COMMON, tone_gamma=2.2, tone_intensity=1.0, tone_mapping=ToneMapper.aces),
    'adaptive_aces': ImageProcessingSettings(**_COMMON, tone_gamma=1.5, tone_intensity=2.0, light_adapt=0.8, tone_mapping=ToneMapper.adaptive_aces),
    'reinhard': ImageProcessingSettings(**_COMMON, tone_gamma=1.0, tone_intensity=2.5, light_adapt=0.8, tone_mapping=ToneMapper.reinhard),
}
aces, adaptive_aces, reinhard = presets['aces'], presets['adaptive_aces'], presets['reinhard']


def get_preset(name: str) -> ImageProcessingSettings:
    if name not in presets:
        raise ValueError(f'Unknown preset: {name}. Available: {list(presets.keys())}')
    return presets[name]
