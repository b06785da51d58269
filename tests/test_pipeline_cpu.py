"""CPU-only: pipeline host logic -- settings models against the reference's own camera JSON data
files (its single test, tests/test_camera_settings_serialization.py, is this round trip), presets,
transforms, file-to-camera lookup, size helpers."""

from pathlib import Path

import pytest
import torch

CAMERA_JSON = sorted((Path(__file__).parent / 'golden' / 'camera_settings').glob('*.json'))


@pytest.mark.parametrize('path', CAMERA_JSON, ids=lambda p: p.stem)
def test_reference_camera_json_roundtrip(td, path):
    from torch_darktable.pipeline import CameraSettings

    original = CameraSettings.load_json(path)
    reloaded = CameraSettings.model_validate_json(original.model_dump_json())
    assert original == reloaded
    w, h = original.image_size
    assert original.bytes == w * h * 3 // 2 + original.padding
    assert original.name == path.stem


def test_camera_json_fields(td):
    from torch_darktable.pipeline import CameraSettings, ImageTransform, ToneMapper

    pfr = CameraSettings.load_json(Path(__file__).parent / 'golden' / 'camera_settings' / 'pfr.json')
    assert pfr.image_size == (4112, 3008) and pfr.padding == 1536 and pfr.bytes == 18554880
    assert pfr.bayer_pattern is td.BayerPattern.RGGB and pfr.packed_format is td.PackedFormat.Packed12
    assert pfr.image_processing.tone_mapping is ToneMapper.aces and pfr.transform is ImageTransform.rotate_270
    beet = CameraSettings.load_json(Path(__file__).parent / 'golden' / 'camera_settings' / 'beetroot.json')
    assert isinstance(beet.transform, dict) and beet.get_image_transform('cam1') is ImageTransform.rotate_90
    assert beet.get_image_transform('unknown') is ImageTransform.none


def test_settings_validation_and_presets(td, tmp_path):
    from pydantic import ValidationError
    from torch_darktable.pipeline import Debayer, ImageProcessingSettings, ToneMapper, get_preset, presets
    from torch_darktable.pipeline.config import Float, get_validator

    s = ImageProcessingSettings()
    assert (s.tone_gamma, s.tone_intensity, s.light_adapt, s.denoise, s.bilateral) == (0.75, 2.0, 1.0, 0.075, 0.4)
    assert s.debayer is Debayer.rcd and s.tone_mapping is ToneMapper.reinhard and s.moving_average == 0.02
    with pytest.raises(ValidationError):
        ImageProcessingSettings(tone_gamma=9.0)
    with pytest.raises(ValidationError):
        ImageProcessingSettings(debayer='nearest')
    assert ImageProcessingSettings(debayer='ppg').debayer is Debayer.ppg
    p = tmp_path / 's.json'
    s.save_json(p)
    assert ImageProcessingSettings.load_json(p) == s and '"debayer": "rcd"' in p.read_text()
    v = get_validator(ImageProcessingSettings, 'tone_gamma')
    assert isinstance(v, Float) and v.range == (0.1, 5.0) and v.description == 'Gamma'
    assert set(presets) == {'aces', 'adaptive_aces', 'reinhard'} and get_preset('aces').tone_gamma == 2.2
    assert get_preset('reinhard').light_adapt == 0.8 and get_preset('adaptive_aces').vibrance == 0.5
    with pytest.raises(ValueError):
        get_preset('nope')


def test_transforms(td):
    from torch_darktable.pipeline import ImageTransform, transform, transformed_size

    img = torch.arange(24).view(2, 4, 3)
    assert transform(img, ImageTransform.none) is img
    assert torch.equal(transform(img, ImageTransform.rotate_90), torch.rot90(img, 1, (0, 1)))
    assert torch.equal(transform(transform(img, ImageTransform.rotate_90), ImageTransform.rotate_270), img)
    assert torch.equal(transform(img, ImageTransform.transverse), torch.flip(img, (0, 1)))
    assert torch.equal(transform(img, ImageTransform.transpose), img.transpose(0, 1))
    assert transform(img, ImageTransform.flip_horiz).is_contiguous()
    assert transformed_size((4, 2), ImageTransform.rotate_90) == (2, 4) and transformed_size((4, 2), ImageTransform.flip_vert) == (4, 2)
    t = ImageTransform.none
    for expect in (ImageTransform.rotate_90, ImageTransform.rotate_180, ImageTransform.rotate_270, ImageTransform.none):
        t = t.next_rotation()
        assert t is expect
    assert ImageTransform.transverse.next_rotation() is ImageTransform.transpose


def test_util_and_file_lookup(td, tmp_path):
    from torch_darktable.pipeline import CameraSettings, ImageProcessingSettings, settings_for_file
    from torch_darktable.pipeline.util import lerp, normalize_image, resize_longest_edge

    assert resize_longest_edge((4096, 3072), 0) == (4096, 3072) and resize_longest_edge((4096, 3072), 1024) == (1024, 768)
    assert resize_longest_edge((3000, 4000), 1000) == (750, 1000)
    assert torch.allclose(lerp(torch.tensor([0.0, 2.0]), torch.tensor([1.0, 4.0]), 0.25), torch.tensor([0.25, 2.5]))
    assert torch.allclose(normalize_image(torch.tensor([1.0, 3.0]), torch.tensor([1.0, 5.0])), torch.tensor([0.0, 0.5]))
    cams = tmp_path / 'cams'
    cams.mkdir()
    cs = CameraSettings(name='tiny', image_size=(8, 4), padding=16, image_processing=ImageProcessingSettings())
    cs.save_json(cams / 'tiny.json')
    raw_dir = tmp_path / 'somewhere'
    raw_dir.mkdir()
    f = raw_dir / 'frame.raw'
    f.write_bytes(bytes(cs.bytes))
    assert settings_for_file(f, cams) == cs                      # matched by file size
    named = tmp_path / 'tiny'
    named.mkdir()
    g = named / 'x.raw'
    g.write_bytes(b'123')
    assert settings_for_file(g, cams) == cs                      # matched by directory name
    f.write_bytes(bytes(7))
    with pytest.raises(ValueError):
        settings_for_file(f, cams)


def test_pipeline_helpers_match_reference_fixtures(td):
    """pipeline/util.py and pipeline/transform.py against outputs of the reference's own functions
    (tests/golden/make_golden.py -> reference_helpers.npz)."""
    import numpy as np

    from torch_darktable.pipeline.transform import ImageTransform, transform, transformed_size
    from torch_darktable.pipeline.util import lerp, normalize_image, resize, resize_longest_edge

    g = np.load(Path(__file__).parent / 'golden' / 'reference_helpers.npz')
    img, bounds = torch.from_numpy(g['pipe_img']), torch.from_numpy(g['pipe_bounds'])
    assert np.array_equal(normalize_image(img, bounds).numpy(), g['pipe_normalized'])
    assert np.array_equal(lerp(img, img.flip(0), 0.3).numpy(), g['pipe_lerp'])
    assert np.allclose(resize(img, (4, 5)).numpy(), g['pipe_resized'], rtol=0, atol=1e-6)
    sizes = [tuple(int(v) for v in s) for s in g['pipe_resize_sizes_in']]
    got = [[*resize_longest_edge(s, L)] for s in sizes for L in (0, 512, 1000)]
    assert np.array_equal(np.array(got), g['pipe_resize_sizes_out'])
    small = torch.from_numpy(g['pipe_transform_in'])
    for t in ImageTransform:
        out = transform(small, t)
        assert out.is_contiguous() and np.array_equal(out.numpy(), g[f'pipe_transform_{t.name}'])
        assert np.array_equal(np.array(transformed_size((640, 480), t)), g[f'pipe_transformed_size_{t.name}'])
        assert t.next_rotation().value == int(g[f'pipe_next_rotation_{t.name}'])
