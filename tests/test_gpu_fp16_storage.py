"""float16 storage for the ops that took float32 images only until round 5 (PostProcess, apply / estimate white balance, the nine
colour operators, Laplacian): an extension of this build -- the reference is float32-only (`TORCH_CHECK(... kFloat32)` everywhere).
Contract: fp32 arithmetic on the stored binary16 values, one rounding at the final store -- so the stencil ops equal the fp32
oracle on the same values, rounded once, BIT FOR BIT; the colour operators within their fp32 tolerance (2e-5) plus half a binary16
ulp; the Laplacian carries the same values as its float32 form (its result is binary16 by construction, laplacian.cu:92-108)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PATTERNS = ['RGGB', 'BGGR', 'GRBG', 'GBRG']


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a visible MI355X'
    return torch.device('cuda', 0)


def gpu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def npy(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('pattern', PATTERNS)
@pytest.mark.parametrize('h,w', [(64, 96), (90, 134), (131, 203)])   # 4-pixel vector path, and the scalar paths (width % 4 != 0)
@pytest.mark.parametrize('passes,local,glob', [(1, False, False), (3, True, False), (5, True, True), (0, True, True), (0, False, True), (2, False, True)])
def test_postprocess_fp16_storage_rounded_once(td, oracle, dev, scene, pattern, h, w, passes, local, glob):
    rgb16 = oracle.ppg(oracle.mosaic(scene(h, w, 3 * h + w), oracle.PATTERNS[pattern]), oracle.PATTERNS[pattern]).astype(np.float16)
    post = td.PostProcess(dev, (w, h), getattr(td.BayerPattern, pattern), color_smoothing_passes=passes, green_eq_local=local, green_eq_global=glob,
                          green_eq_threshold=4.0)
    got = npy(post.process(gpu(rgb16, dev)))
    assert got.dtype == np.float16
    x32 = rgb16.astype(np.float32)
    got32 = npy(post.process(gpu(x32, dev)))   # the float32 form on the same values
    if not glob:
        ref = oracle.postprocess(x32, oracle.PATTERNS[pattern], passes, local, glob, 4.0)
        assert np.array_equal(got32, ref)
    assert np.array_equal(got.view(np.uint16), got32.astype(np.float16).view(np.uint16))   # == the fp32 result rounded once, every stage count


@pytest.mark.parametrize('pattern', PATTERNS)
def test_white_balance_fp16_storage(td, oracle, dev, scene, pattern):
    h, w = 90, 134
    b16 = oracle.mosaic(scene(h, w, 41), oracle.PATTERNS[pattern])[:, :, 0].astype(np.float16)
    gains = np.array([1.9, 1.0, 1.45], np.float32)
    got = npy(td.apply_white_balance(gpu(b16, dev), gpu(gains, dev), getattr(td.BayerPattern, pattern)))
    ref = oracle.apply_white_balance(b16.astype(np.float32), gains, oracle.PATTERNS[pattern])
    assert got.dtype == np.float16 and np.array_equal(got.view(np.uint16), ref.astype(np.float16).view(np.uint16))
    # the estimate reads the binary16 mosaic; its samples and gains are fp32 and equal those of the same values stored as float32
    a = td.estimate_white_balance([gpu(b16, dev)], getattr(td.BayerPattern, pattern), 0.9, 4)
    b = td.estimate_white_balance([gpu(b16.astype(np.float32), dev)], getattr(td.BayerPattern, pattern), 0.9, 4)
    assert torch.equal(a, b)


def test_color_operators_fp16_storage(td, oracle, dev, scene):
    rgb16 = scene(70, 102, 8).astype(np.float16)    # 102 % 4 != 0 and 70 * 102 % 4 == 0: vector body; odd sizes below take the tail
    x32 = rgb16.astype(np.float32)
    for shape in ((70, 102), (7, 9)):
        img16 = rgb16[:shape[0], :shape[1]].copy()
        img32 = img16.astype(np.float32)
        lab32 = oracle.color_op('rgb_to_lab', img32)
        cases = [('rgb_to_xyz', img16, None), ('rgb_to_lab', img16, None), ('lab_to_rgb', lab32.astype(np.float16), None),
                 ('xyz_to_lab', oracle.color_op('rgb_to_xyz', img32).astype(np.float16), None), ('modify_hsl', img16, (0.1, 1.2, 0.95)),
                 ('modify_vibrance', img16, (0.3,))]
        for name, src16, prm in cases:
            fn = getattr(td, name)
            got = npy(fn(gpu(src16, dev), *prm) if prm else fn(gpu(src16, dev)))
            assert got.dtype == np.float16
            ref = oracle.color_op(name, src16.astype(np.float32), prm)
            ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.maximum(np.abs(ref), np.abs(got.astype(np.float32))), 2.0 ** -14))) - 10)
            assert (np.abs(got.astype(np.float32) - ref) <= 0.5 * ulp + 2e-5).all(), name
    m = np.array([[0.9, 0.1, 0.0], [0.05, 0.9, 0.05], [0.0, 0.2, 0.8]], np.float32)
    got = npy(td.color_transform_3x3(gpu(rgb16, dev), gpu(m, dev)))
    ref = np.clip(x32 @ m.T, 0.0, 1.0)
    assert got.dtype == np.float16 and np.abs(got.astype(np.float32) - ref).max() <= 2.0 ** -11 + 2e-6


@pytest.mark.parametrize('h,w,clarity', [(128, 192, 0.0), (203, 331, 0.3)])
def test_laplacian_fp16_storage_same_values(td, dev, scene, h, w, clarity):
    lum16 = gpu(scene(h, w, 77)[:, :, 1].astype(np.float16), dev)
    lap = td.Laplacian(dev, (w, h), td.LaplacianParams(6, 0.2, 1.6, 0.7, clarity))
    got16 = lap.process(lum16)
    got32 = lap.process(lum16.float())
    assert got16.dtype == torch.float16 and torch.equal(got16.float(), got32)
