"""GPU parity: every HIP op, called through the torch_darktable drop-in surface (ctypes ->
C ABI -> kernels), against the CPU oracle on the same seeded inputs.

Tolerance ladder (the reference itself is a fast-math CUDA build whose
atomics are order-nondeterministic and ships no golden vectors, so parity is "unpinned" by the
reference and anchored on the oracle's literal restatement):
  T0 bit-exact  : codec, bilinear, PPG, RCD, colour smoothing, local green eq, white balance, bilateral
  T1 sum order  : global green eq -- the ratio of two fp32 sums taken in different association orders; bound derived
                  in test_postprocess_global_green_eq from the block counts (a few 1e-6 relative)
  T2 1e-5..1e-4 : anything with powf/expf/logf/cbrtf, bilateral / Wiener (sum order)
  u8 tonemaps   : +-1 LSB on a bounded fraction of samples
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PATTERNS = ['RGGB', 'BGGR', 'GRBG', 'GBRG']


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a visible MI355X'
    return torch.device('cuda', 0)


def gpu(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def half_ulp(v):
    """One binary16 ulp of |v| (normal range)."""
    return 2.0 ** (np.floor(np.log2(np.maximum(np.abs(v), 2.0 ** -14))) - 10)


def npy(t):
    return t.detach().cpu().numpy()


def assert_f16_close(got16, ref32, what='', flips=1e-5, touched=1e-3):
    """A binary16 result computed with the approximate arithmetic flavour against the fp32 oracle rounded once: at most ONE
    binary16 ulp everywhere, except on a fraction <= `flips` of the pixels (near-tie selections that flip: counted separately,
    SURVEY.md 8c), and only a fraction <= `touched` of the values may differ at all (measured: ~5e-5)."""
    assert got16.dtype == np.float16, got16.dtype
    ref16 = ref32.astype(np.float16)
    g, r = got16.astype(np.float32), ref16.astype(np.float32)
    assert np.isfinite(g[np.isfinite(r)]).all(), what
    ok = np.isfinite(r)
    d = np.where(ok, np.abs(g - r), 0.0)
    over = (d > half_ulp(np.maximum(np.abs(g), np.abs(r)))).any(axis=-1)
    assert over.mean() <= flips, f'{what}: {over.sum()} pixels beyond one binary16 ulp ({over.mean():.2e}), first at {np.argwhere(over)[:5].tolist()}, max |d| {d.max()}'
    assert (d > 0).mean() <= touched, f'{what}: {(d > 0).mean():.2e} of the values differ'
    return over.sum(), float((d > 0).mean())


def max_ulp(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.abs(a - b).max())


# ------------------------------------------------------------------ codec
@pytest.mark.parametrize('ids', [False, True])
@pytest.mark.parametrize('n', [0, 2, 6, 8, 4096, 100002])
def test_codec_all_directions(td, oracle, dev, ids, n):
    rng = np.random.default_rng(n + ids)
    u16 = rng.integers(0, 5000, n, dtype=np.uint16)  # includes values > 4095 (clamped)
    f32 = rng.uniform(-0.1, 1.2, n).astype(np.float32)
    enc_u = npy(td.encode12_u16(gpu(u16, dev), ids_format=ids))
    assert np.array_equal(enc_u, oracle.encode12_u16(u16, ids))
    enc_f = npy(td.encode12_float(gpu(f32, dev), ids_format=ids))
    assert np.array_equal(enc_f, oracle.encode12_f32(f32, ids, True))
    enc_fu = npy(td.extension.extension.encode12_float(gpu(f32 * 4095, dev), ids_format=ids, scaled=False))
    assert np.array_equal(enc_fu, oracle.encode12_f32(f32 * 4095, ids, False))
    packed = rng.integers(0, 256, n // 2 * 3, dtype=np.uint8)
    assert np.array_equal(npy(td.decode12_u16(gpu(packed, dev), ids_format=ids)), oracle.decode12_u16(packed, ids))
    assert np.array_equal(npy(td.decode12_float(gpu(packed, dev), ids_format=ids)), oracle.decode12_f32(packed, ids, True))
    got_h = npy(td.decode12_half(gpu(packed, dev), ids_format=ids)).view(np.uint16)
    assert np.array_equal(got_h, oracle.decode12_f16(packed, ids, True).view(np.uint16))


def test_codec_unaligned_views(td, oracle, dev):
    """Slices that break the 4-B / 16-B alignment of the bulk kernels take the per-pair path."""
    rng = np.random.default_rng(5)
    packed = rng.integers(0, 256, 3 * 1001 + 1, dtype=np.uint8)
    t = gpu(packed, dev)[1:]
    assert np.array_equal(npy(td.decode12_float(t)), oracle.decode12_f32(packed[1:], False, True))


def test_codec_roundtrip_full_range(td, dev):
    x = torch.arange(4096, dtype=torch.int32).to(torch.uint16).to(dev)
    assert np.array_equal(npy(td.decode12_u16(td.encode12_u16(x))), npy(x))
    f = td.decode12_float(td.encode12_u16(x))
    assert np.array_equal(npy(td.decode12_u16(td.encode12_float(f))), npy(x))


# ------------------------------------------------------------------ demosaic
@pytest.mark.parametrize('pattern', PATTERNS)
@pytest.mark.parametrize('size', [(64, 96), (150, 202), (37, 50)])
def test_bilinear_bit_exact(td, oracle, dev, scene, pattern, size):
    h, w = size
    bayer = oracle.mosaic(scene(h, w, 11), oracle.PATTERNS[pattern])
    got = npy(td.bilinear5x5_demosaic(gpu(bayer, dev), td.BayerPattern[pattern]))
    assert np.array_equal(got, oracle.bilinear5x5(bayer, oracle.PATTERNS[pattern]))


@pytest.mark.parametrize('pattern', PATTERNS)
@pytest.mark.parametrize('size', [(64, 96), (150, 202), (33, 70)])
@pytest.mark.parametrize('median', [0.0, 1.5])
def test_ppg_bit_exact(td, oracle, dev, scene, pattern, size, median):
    h, w = size
    bayer = oracle.mosaic(scene(h, w, 12), oracle.PATTERNS[pattern])
    ws = td.PPG(dev, (w, h), td.BayerPattern[pattern], median_threshold=median)
    got = npy(ws.process(gpu(bayer, dev)))
    assert np.array_equal(got, oracle.ppg(bayer, oracle.PATTERNS[pattern], median))


@pytest.mark.parametrize('pattern', PATTERNS)
@pytest.mark.parametrize('size', [(64, 96), (150, 202), (128, 128), (70, 66), (20, 24), (12, 40)])
def test_rcd_bit_exact(td, oracle, dev, scene, pattern, size):
    h, w = size
    bayer = oracle.mosaic(scene(h, w, 13), oracle.PATTERNS[pattern])
    ws = td.RCD(dev, (w, h), td.BayerPattern[pattern])
    got = npy(ws.process(gpu(bayer, dev)))
    ref = oracle.rcd(bayer, oracle.PATTERNS[pattern])
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f'{len(bad)} mismatches, first at {bad[:5].tolist()}, max |d| {np.abs(got - ref).max()}'


@pytest.mark.parametrize('pattern', PATTERNS)
@pytest.mark.parametrize('size', [(64, 128), (203, 300), (130, 258), (321, 128)])
def test_rcd_strips_equal_tiles(td, oracle, dev, scene, pattern, size):
    """Frames of at least 128 x 64 run as column strips walked down the frame (csrc/tdk_rcd_stream.h), smaller ones as
    64 x 64 LDS tiles; both must give the oracle's bits -- here on the same input, fp32 and fp16 storage, with frames that
    hold one strip exactly, several strips and segments, an odd height, and a last strip / segment moved back onto its
    neighbour."""
    h, w = size
    bayer = oracle.mosaic(scene(h, w, 17), oracle.PATTERNS[pattern])
    ref = oracle.rcd(bayer, oracle.PATTERNS[pattern])
    b16 = bayer.astype(np.float16)
    ref16 = oracle.rcd(b16.astype(np.float32), oracle.PATTERNS[pattern]).astype(np.float16)
    ws = td.RCD(dev, (w, h), td.BayerPattern[pattern])
    from torch_darktable import torch_darktable_extension as ext
    # float16 results: TDK_RCD_EXACT = the exact flavour rounded once (the default approximate flavour of the strips has its own
    # test below)
    strips = npy(ws.process(gpu(bayer, dev)))
    with ext.verification_paths(rcd_exact=True):
        strips16 = npy(ws.process(gpu(b16, dev)))
    with ext.verification_paths(rcd_tiles=True):
        tiles, tiles16 = npy(ws.process(gpu(bayer, dev))), npy(ws.process(gpu(b16, dev)))
    with ext.concurrent_frames():  # the register-blocked strips (csrc/tdk_rcd_quad.h, TDK_RCD_CONCURRENT)
        quad = npy(ws.process(gpu(bayer, dev)))
        with ext.verification_paths(rcd_exact=True):
            quad16 = npy(ws.process(gpu(b16, dev)))
    for name, got, want in (('strips', strips, ref), ('tiles', tiles, ref), ('strips f16', strips16, ref16), ('tiles f16', tiles16, ref16),
                            ('quad strips', quad, ref), ('quad strips f16', quad16, ref16)):
        bad = np.argwhere(got != want)
        assert bad.size == 0, f'{name}: {len(bad)} mismatches, first at {bad[:5].tolist()}'


@pytest.mark.parametrize('pattern', PATTERNS)
@pytest.mark.parametrize('size', [(64, 128), (203, 300), (130, 258), (321, 128), (150, 202), (512, 640)])
def test_rcd_fp16_fast_arithmetic(td, oracle, dev, scene, pattern, size):
    """The default float16 result of the column strips is computed with the approximate arithmetic flavour (a * rcp(b)
    quotients, fused sums of products: csrc/tdk_rcd_stream.h; the reference itself is an nvcc --use_fast_math build).  Against
    the fp32 oracle rounded to binary16: at most one binary16 ulp everywhere but on <= 1e-5 of the pixels (selection flips), for
    both strip kernels, which must agree with each other bit for bit; float16 in and float32 in (the fused head's mixed form)."""
    from torch_darktable import torch_darktable_extension as ext
    h, w = size
    bayer = oracle.mosaic(scene(h, w, 19), oracle.PATTERNS[pattern])
    b16 = bayer.astype(np.float16)
    ref = oracle.rcd(b16.astype(np.float32), oracle.PATTERNS[pattern])
    ws = td.RCD(dev, (w, h), td.BayerPattern[pattern])
    strips = npy(ws.process(gpu(b16, dev)))
    with ext.concurrent_frames():
        quad = npy(ws.process(gpu(b16, dev)))
    assert np.array_equal(strips.view(np.uint16), quad.view(np.uint16)), 'the two strip kernels disagree in the approximate flavour'
    assert_f16_close(strips, ref, f'{pattern} {size}')
    # native samples pass through untouched
    for (dy, dx) in ((0, 0), (0, 1), (1, 0), (1, 1)):
        c = oracle.cfa_color(dy, dx, oracle.PATTERNS[pattern]) if hasattr(oracle, 'cfa_color') else None
        if c is not None:
            assert np.array_equal(strips[8 + dy:h - 8:2, 8 + dx:w - 8:2, c], np.maximum(b16[8 + dy:h - 8:2, 8 + dx:w - 8:2, 0], 0))


def test_rcd_strips_equal_tiles_on_many_geometries(td, dev):
    """Strip / segment geometry sweep: widths around the multiples of the 108-column strip (one strip exactly, a last strip
    moved back by 2 .. 106 columns), heights around the segment rules (64 = the minimum, odd, one row more than a multiple of
    the 8-row step), all against the tile kernel (itself tied to the oracle above) on every pixel."""
    from torch_darktable import torch_darktable_extension as ext
    rng = np.random.default_rng(2024)
    widths = [128, 130, 214, 216, 218, 234, 322, 324, 326, 1000]
    heights = [64, 65, 71, 72, 73, 127, 128, 129, 200, 333]
    cases = [(w, h) for w, h in zip(widths, heights)] + [(int(rng.integers(64, 400)) * 2, int(rng.integers(64, 500))) for _ in range(10)]
    for w, h in cases:
        bayer = torch.from_numpy(rng.random((h, w, 1), dtype=np.float32)).to(dev)
        pattern = [td.BayerPattern.RGGB, td.BayerPattern.BGGR, td.BayerPattern.GRBG, td.BayerPattern.GBRG][(w // 2 + h) % 4]
        ws = td.RCD(dev, (w, h), pattern)
        strips = ws.process(bayer)
        with ext.verification_paths(rcd_tiles=True):
            tiles = ws.process(bayer)
        with ext.concurrent_frames():
            quad = ws.process(bayer)
        assert torch.equal(strips, tiles), f'{w}x{h} {pattern}: {(strips != tiles).sum().item()} values differ'
        assert torch.equal(quad, tiles), f'{w}x{h} {pattern}, register-blocked strips: {(quad != tiles).sum().item()} values differ'


@pytest.mark.parametrize('size', [(2, 2), (2, 4), (4, 4), (4, 6), (6, 8), (10, 14), (14, 16), (16, 16), (7, 5), (3, 64), (64, 2)])
def test_tiny_images_every_stencil_op(td, oracle, dev, size):
    """Images smaller than every halo / tile / ring: all pixels are border cases (the oracle was run
    under AddressSanitizer on the same sizes).  Bit-exact ops stay bit-exact."""
    h, w = size
    rng = np.random.default_rng(h * 100 + w)
    bayer = rng.random((h, w, 1), dtype=np.float32)
    for name in PATTERNS:
        pat, opat = td.BayerPattern[name], oracle.PATTERNS[name]
        assert np.array_equal(npy(td.bilinear5x5_demosaic(gpu(bayer, dev), pat)), oracle.bilinear5x5(bayer, opat))
        for med in (0.0, 2.0):
            assert np.array_equal(npy(td.PPG(dev, (w, h), pat, median_threshold=med).process(gpu(bayer, dev))), oracle.ppg(bayer, opat, med))
        if w % 2 == 0:
            assert np.array_equal(npy(td.RCD(dev, (w, h), pat).process(gpu(bayer, dev))), oracle.rcd(bayer, opat))
    rgb = rng.random((h, w, 3), dtype=np.float32)
    cfg = dict(color_smoothing_passes=5, green_eq_local=True)
    assert np.array_equal(npy(td.PostProcess(dev, (w, h), td.BayerPattern.RGGB, **cfg).process(gpu(rgb, dev))),
                          oracle.postprocess(rgb, oracle.RGGB, **cfg))
    gains = np.array([1.7, 1.0, 1.4], np.float32)
    assert np.array_equal(npy(td.apply_white_balance(gpu(bayer[:, :, 0], dev), gpu(gains, dev), td.BayerPattern.GRBG)),
                          oracle.apply_white_balance(bayer[:, :, 0], gains, oracle.GRBG))
    lum = oracle.compute_luminance(rgb)
    assert np.abs(npy(td.compute_luminance(gpu(rgb, dev))) - lum).max() < 2e-5
    m = oracle.image_metrics([rgb], 2)
    assert np.allclose(npy(td.compute_image_metrics([gpu(rgb, dev)], stride=2)), m, rtol=2e-5, atol=2e-6)
    got = npy(td.reinhard_tonemap(gpu(rgb, dev), gpu(m, dev), td.TonemapParameters(0.75, 2.0, 1.0, 0.3))).astype(np.int32)
    assert np.abs(got - oracle.tonemap('reinhard', rgb, m, 0.75, 2.0, 1.0, 0.3).astype(np.int32)).max() <= 1


def test_rcd_negative_and_pure_function(td, oracle, dev, scene):
    """Negative raw samples are clamped; a second call on the same workspace gives the same
    result (the reference's call-history dependence is not reproduced) in a fresh tensor."""
    h, w = 96, 128
    bayer = oracle.mosaic(scene(h, w, 14), oracle.RGGB) - 0.05
    ws = td.RCD(dev, (w, h), td.BayerPattern.RGGB)
    a = ws.process(gpu(bayer, dev))
    other = ws.process(gpu(oracle.mosaic(scene(h, w, 15), oracle.RGGB), dev))
    b = ws.process(gpu(bayer, dev))
    assert a.data_ptr() != other.data_ptr()
    assert torch.equal(a, b)
    assert np.array_equal(npy(a), oracle.rcd(bayer, oracle.RGGB))


@pytest.mark.parametrize('case', ['plain', 'black_blocks', 'flat', 'wide_exponents', 'tiny_beside_huge', 'below_min', 'denormals', 'huge', 'nan'])
def test_rcd_division_flavours_bit_exact(td, oracle, dev, scene, case):
    """csrc/rcd.hip runs interior tiles whose samples all lie in {0} U [2^-24, 2^16] on the bare core of the IEEE
    division (tdk_fastdiv.h) and everything else on `/`; both must give the oracle's bits.  320 x 384 has 3 x 4
    interior tiles; the cases steer them onto either flavour and onto the per-wave fallback (zero numerators)."""
    h, w = 320, 384
    rng = np.random.default_rng(77)
    bayer = oracle.mosaic(scene(h, w, 21), oracle.RGGB)[:, :, 0].copy()
    if case == 'black_blocks':          # exact zeros inside fast tiles: numerators of steps 5.1 / 5.2 are exactly 0 there
        bayer[70:120, 70:200] = 0.0
        bayer[200:206, 100:300] = 0.0
        bayer[130:190:2, 210:260:2] = 0.0
    elif case == 'flat':                # constant patches: gradients reduce to eps, estimates cancel almost exactly
        bayer[80:140, 80:160] = 0.5
        bayer[150:250, 200:330] = 1.0
    elif case == 'wide_exponents':      # every sample in range, magnitudes spread over the whole allowed interval
        bayer = (rng.random((h, w), dtype=np.float32) + 1.0) * np.exp2(rng.integers(-24, 15, (h, w))).astype(np.float32)
        bayer[rng.random((h, w)) < 0.1] = 0.0
    elif case == 'tiny_beside_huge':    # the corner the unguarded step-3.1 divisions are closest to the bare core's limit in:
        # isolated 2^-24 samples whose same-colour neighbours are 2^16 (the low-pass ratio is then ~2^-63) while the
        # opposite-side samples are equal (that gradient is exactly eps = 1e-5): second-level numerators ~2^-80
        bayer[64:256, 64:320] = np.float32(65536.0)
        bayer[100:220:8, 100:280:8] = np.float32(2.0 ** -24)      # R sites
        bayer[101:221:8, 101:281:8] = np.float32(2.0 ** -24)      # B sites
        bayer[104:224:8, 105:285:8] = np.float32(2.0 ** -24)      # green sites
    elif case == 'below_min':           # a single sample just under 2^-24 in one tile: that tile falls back
        bayer[100, 100] = np.float32(2.0 ** -25)
        bayer[230, 300] = np.float32(2.0 ** -24)      # exactly the limit: still fast
    elif case == 'denormals':
        bayer[90:110, 90:110] = np.float32(1e-41)
        bayer[220:230, 220:330] *= np.float32(1e-36)
    elif case == 'huge':
        bayer[100, 101] = np.float32(65536.0)          # the upper limit: fast
        bayer[100, 230] = np.float32(65537.0)          # above: fallback
        bayer[240:244, 100:104] = np.float32(3e18)     # squares overflow -> inf / nan in both implementations
    elif case == 'nan':
        bayer[101, 99] = np.nan                        # max(0, nan) = 0 in both
        bayer[250, 250] = -0.0
    bayer = np.ascontiguousarray(bayer[:, :, None], dtype=np.float32)
    with np.errstate(all='ignore'):
        ref = oracle.rcd(bayer, oracle.RGGB)
    got = npy(td.RCD(dev, (w, h), td.BayerPattern.RGGB).process(gpu(bayer, dev)))
    same = (got == ref) | (np.isnan(got) & np.isnan(ref))
    bad = np.argwhere(~same)
    assert bad.size == 0, f'{case}: {len(bad)} mismatches, first at {bad[:5].tolist()}'
    if case != 'huge':
        assert np.isfinite(got).all()


def test_rcd_fp16_extremes_stay_on_the_exact_path(td, oracle, dev, scene):
    """binary16 storage spans exactly the sample range of the fast division flavour (smallest subnormal 2^-24, largest
    finite 65504 < 2^16): extreme but representable samples must still give the oracle's bits (fp32 math, one rounding
    at the store)."""
    h, w = 320, 384
    b16 = oracle.mosaic(scene(h, w, 23), oracle.RGGB)[:, :, 0].astype(np.float16)
    b16[100:104, 100:140] = np.float16(5.96e-8)      # 2^-24
    b16[200:203, 90:300] = np.float16(65504.0)
    b16[250:260:2, 200:240:2] = np.float16(0.0)
    b16 = np.ascontiguousarray(b16[:, :, None])
    from torch_darktable import torch_darktable_extension as ext
    with ext.verification_paths(rcd_exact=True):
        got = npy(td.RCD(dev, (w, h), td.BayerPattern.RGGB).process(gpu(b16, dev)))
    with np.errstate(all='ignore'):
        ref32 = oracle.rcd(b16.astype(np.float32), oracle.RGGB)
        ref = ref32.astype(np.float16)
    assert got.dtype == np.float16
    same = (got == ref) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), f'{(~same).sum()} mismatches, first at {np.argwhere(~same)[:5].tolist()}'
    # the default (approximate) flavour has no range wrapper to fall back on: the same extremes must stay finite where the
    # oracle is and within one binary16 ulp of it
    # -- away from the 65504 block within one binary16 ulp as everywhere; next to it (samples 2e5 times their neighbours) the
    # sums of products cancel sixteen orders of magnitude and both flavours carry rounding noise of ~1e-7 of the LARGEST
    # neighbour: there the bound is one binary16 ulp + 2e-7 * (largest sample within 4 pixels)
    with np.errstate(all='ignore'):
        fast = npy(td.RCD(dev, (w, h), td.BayerPattern.RGGB).process(gpu(b16, dev)))
    inside = np.abs(ref32) < 65000.0  # (beyond, the oracle's own float32 result rounds to binary16 infinity: up to 95 760 next to the 65 504 block)
    assert np.isfinite(fast[inside]).all()
    from scipy.ndimage import maximum_filter
    near = maximum_filter(b16[:, :, 0].astype(np.float32), size=9)[:, :, None]
    g, r = fast.astype(np.float32), ref.astype(np.float32)
    ok = np.isfinite(r) & inside
    with np.errstate(all='ignore'):
        d = np.where(ok, np.abs(g - r), 0.0)
    tol = half_ulp(np.maximum(np.where(ok, np.abs(g), 0.0), np.where(ok, np.abs(r), 0.0))) + np.where(near > 100.0, 2e-7 * near, 0.0)
    over = (d > tol).any(-1)
    assert over.mean() <= 1e-5, f'{over.sum()} pixels beyond the bound, first at {np.argwhere(over)[:5].tolist()}, max |d| {d.max()}'


def test_rcd_rejects_odd_width_and_wrong_shape(td, dev):
    ws = td.RCD(dev, (64, 32), td.BayerPattern.RGGB)
    with pytest.raises(RuntimeError):
        ws.process(torch.zeros(32, 60, 1, device=dev))
    odd = td.RCD(dev, (63, 32), td.BayerPattern.RGGB)
    with pytest.raises(RuntimeError):
        odd.process(torch.zeros(32, 63, 1, device=dev))


@pytest.mark.parametrize('cfg', [dict(color_smoothing_passes=1), dict(color_smoothing_passes=3), dict(green_eq_local=True),
                                 dict(color_smoothing_passes=4), dict(color_smoothing_passes=6), dict(color_smoothing_passes=9, green_eq_local=True),
                                 dict(color_smoothing_passes=2, green_eq_local=True, green_eq_threshold=4.0), dict()])
def test_postprocess_bit_exact_paths(td, oracle, dev, scene, cfg):
    h, w = 90, 134
    rgb = oracle.rcd(oracle.mosaic(scene(h, w, 16), oracle.RGGB), oracle.RGGB)
    ws = td.PostProcess(dev, (w, h), td.BayerPattern.RGGB, **cfg)
    got = npy(ws.process(gpu(rgb, dev)))
    assert np.array_equal(got, oracle.postprocess(rgb, oracle.RGGB, **cfg))


@pytest.mark.parametrize('size', [(96, 256), (50, 132), (33, 64), (16, 68), (131, 320)])
@pytest.mark.parametrize('cfg', [dict(color_smoothing_passes=3, green_eq_local=True), dict(color_smoothing_passes=4), dict(color_smoothing_passes=1),
                                 dict(green_eq_local=True, green_eq_threshold=4.0), dict(color_smoothing_passes=7, green_eq_local=True)])
def test_postprocess_vector_paths_bit_exact(td, oracle, dev, scene, size, cfg):
    """Widths that are multiples of 4 take the 16-byte staging / store paths of the smoothing and local green-equilibration kernels
    (partial tiles on the right and bottom, tiles narrower than 64, a single tile row); a scene with negative and > 1 samples
    exercises the clamps."""
    h, w = size
    rgb = oracle.rcd(oracle.mosaic(scene(h, w, 31), oracle.RGGB), oracle.RGGB)
    rgb[5:9, 7:30] -= 0.3
    rgb[h // 2:h // 2 + 3, :] *= 1.7
    ws = td.PostProcess(dev, (w, h), td.BayerPattern.RGGB, **cfg)
    got = npy(ws.process(gpu(rgb, dev)))
    ref = oracle.postprocess(rgb, oracle.RGGB, **cfg)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f'{len(bad)} mismatches, first at {bad[:5].tolist()}'
    # an input view that is not 16-byte aligned takes the scalar loads; same bits
    pad = torch.zeros(h * w * 3 + 1, device=dev)
    pad[1:] = gpu(rgb, dev).reshape(-1)
    assert np.array_equal(npy(ws.process(pad[1:].view(h, w, 3))), ref)


def test_postprocess_global_green_eq(td, oracle, dev, scene):
    h, w = 90, 134
    rgb = oracle.rcd(oracle.mosaic(scene(h, w, 17), oracle.RGGB), oracle.RGGB)
    rgb[0::2, :, 1] *= 1.03  # imbalance the two green phases
    ws = td.PostProcess(dev, (w, h), td.BayerPattern.RGGB, color_smoothing_passes=1, green_eq_global=True, green_eq_local=True)
    got = npy(ws.process(gpu(rgb, dev)))
    ref = oracle.postprocess(rgb, oracle.RGGB, 1, True, True, 0.04)
    # R and B never see the ratio; G1 sites are scaled by ratio = sum(G2) / sum(G1).  Both sums are fp32 sums of n = H W / 4
    # positive terms taken in different association orders (oracle: 16 x 16 block trees, then the B block sums in sequence,
    # postprocess.cu:193-254 + torch.sum; HIP: a tree per workgroup, then atomics): each carries a relative error of at most
    # (depth + chain length) * 2^-24, so the two ratios differ by at most 2 * [(8 + B) + (8 + 12)] * 2^-24 relative.
    blocks = ((h + 15) // 16) * ((w + 15) // 16)
    bound = 2.0 * ((8 + blocks) + (8 + 12)) * 2.0 ** -24
    assert np.array_equal(got[:, :, 0::2], ref[:, :, 0::2])
    assert np.allclose(got[:, :, 1], ref[:, :, 1], rtol=bound, atol=0), (np.abs(got[:, :, 1] / np.maximum(ref[:, :, 1], 1e-30) - 1).max(), bound)
    s32, s64 = oracle.green_eq_sums(oracle.postprocess(rgb, oracle.RGGB, 1), oracle.RGGB)
    assert abs(s64[1] / s64[0] - 1 / 1.03) < 2e-2  # the ratio the op is meant to find (scene greens differ a little)


def test_white_balance_bit_exact(td, oracle, dev, scene):
    bayer = oracle.mosaic(scene(64, 80, 18), oracle.GRBG)[:, :, 0]
    gains = np.array([1.9, 1.0, 1.4], np.float32)
    got = npy(td.apply_white_balance(gpu(bayer, dev), gpu(gains, dev), td.BayerPattern.GRBG))
    assert np.array_equal(got, oracle.apply_white_balance(bayer, gains, oracle.GRBG))


# ------------------------------------------------------------------ colour
COLOR_CASES = [('rgb_to_xyz', None), ('xyz_to_lab', None), ('lab_to_xyz', None), ('xyz_to_rgb', None), ('rgb_to_lab', None),
               ('lab_to_rgb', None), ('modify_hsl', (0.1, 0.3, -0.2)), ('modify_vibrance', (0.6,))]


@pytest.mark.parametrize('name,params', COLOR_CASES)
def test_color_ops(td, oracle, dev, scene, name, params):
    img = scene(61, 83, 19)  # odd pixel count: exercises the vec4 body and the scalar tail
    if name in ('lab_to_xyz', 'lab_to_rgb'):
        img = oracle.color_op('rgb_to_lab', img)
    fn = getattr(td, name)
    got = npy(fn(gpu(img, dev), *(params or ())))
    ref = oracle.color_op(name, img, params)
    assert np.abs(got - ref).max() < 2e-5


def test_color_transform_3x3(td, oracle, dev, scene):
    img = scene(40, 52, 20)
    m = np.array([[1.2, -0.1, -0.1], [-0.05, 1.1, -0.05], [0.0, -0.2, 1.2]], np.float32)
    got = npy(td.color_transform_3x3(gpu(img, dev), gpu(m, dev)))
    assert np.array_equal(got, oracle.color_op('color_transform_3x3', img, m))


@pytest.mark.parametrize('log', [False, True])
def test_luminance_extract_replace(td, oracle, dev, scene, log):
    img = scene(72, 100, 21)
    t = gpu(img, dev)
    lum = td.compute_log_luminance(t, 1e-4) if log else td.compute_luminance(t)
    ref_l = oracle.compute_luminance(img, log, 1e-4)
    assert np.abs(npy(lum) - ref_l).max() < 2e-5
    new_l = ref_l * 0.9 if not log else ref_l - 0.1
    out = td.modify_log_luminance(t, gpu(new_l, dev), 1e-4) if log else td.modify_luminance(t, gpu(new_l, dev))
    assert np.abs(npy(out) - oracle.modify_luminance(img, new_l, log)).max() < 5e-5


def test_color_requires_contiguous_float32_gpu(td, dev):
    with pytest.raises(RuntimeError):
        td.rgb_to_lab(torch.zeros(4, 4, 3))
    with pytest.raises(RuntimeError):
        td.rgb_to_lab(torch.zeros(4, 4, 6, device=dev)[:, :, ::2])
    with pytest.raises(RuntimeError):
        td.compute_log_luminance(torch.zeros(4, 4, 3, device=dev), 0.0)


# ------------------------------------------------------------------ statistics + tonemaps
def test_bounds_and_metrics(td, oracle, dev, scene):
    imgs = [scene(100, 140, 22), scene(100, 140, 23) * 1.3]
    ts = [gpu(i, dev) for i in imgs]
    assert np.array_equal(npy(td.compute_image_bounds(ts, 8)), oracle.image_bounds(imgs, 8))
    for rescale in (False, True):
        got = npy(td.compute_image_metrics(ts, stride=4, min_gray=1e-4, rescale=rescale))
        ref = oracle.image_metrics(imgs, 4, 1e-4, rescale)
        assert np.allclose(got, ref, rtol=2e-5, atol=2e-6)
    sat = np.ones((16, 16, 3), np.float32)
    assert np.array_equal(npy(td.compute_image_metrics([gpu(sat, dev)], 1)), np.zeros(5, np.float32))
    # one image = the single-launch, self-cleaning path: repeated calls (also on another stream) give the same answer
    for k in (0, 1):
        for rescale in (False, True):
            ref = oracle.image_metrics([imgs[k]], 2, 1e-4, rescale)
            for _ in range(3):
                assert np.allclose(npy(td.compute_image_metrics([ts[k]], stride=2, rescale=rescale)), ref, rtol=2e-5, atol=2e-6)
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        got = td.compute_image_metrics([ts[0].half()], stride=2)
    torch.cuda.synchronize()
    assert np.allclose(npy(got), oracle.image_metrics([imgs[0].astype(np.float16).astype(np.float32)], 2, 1e-4, False), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('name', ['reinhard', 'aces', 'adaptive_aces', 'linear'])
@pytest.mark.parametrize('vibrance', [0.0, 0.4])
def test_tonemaps_u8(td, oracle, dev, scene, name, vibrance):
    img = scene(96, 131, 24) * 1.5
    metrics = oracle.image_metrics([img], 8)
    p = td.TonemapParameters(0.75, 2.0 if name != 'aces' else 0.5, 1.0 if name != 'adaptive_aces' else 0.6, vibrance)
    t, m = gpu(img, dev), gpu(metrics, dev)
    if name == 'reinhard':
        got = td.reinhard_tonemap(t, m, p)
    elif name == 'linear':
        got = td.linear_tonemap(t, m, p)
    elif name == 'aces':
        got = td.aces_tonemap(t, p)
    else:
        got = td.aces_tonemap(t, p, m)
    ref_u8, ref_f = oracle.tonemap(name, img, metrics, p.gamma, p.intensity, p.light_adapt, p.vibrance, return_float=True)
    d = np.abs(npy(got).astype(np.int32) - ref_u8.astype(np.int32))
    assert got.dtype == torch.uint8 and d.max() <= 1
    # a +-1 LSB difference is only legitimate where the pre-quantisation value sits on a rounding tie
    frac = np.abs((ref_f * 255.0) - np.floor(ref_f * 255.0) - 0.5)
    assert (frac[d > 0] < 2e-3).all() and (d > 0).mean() < 2e-3


# ------------------------------------------------------------------ bilateral / Wiener / Laplacian
@pytest.mark.parametrize('sig', [(2.0, 0.2), (8.0, 0.1), (3.3, 0.05), (0.7, 0.3)])
def test_bilateral(td, oracle, dev, scene, sig):
    h, w = 150, 203
    lum = oracle.compute_luminance(scene(h, w, 25))
    ws = td.Bilateral(dev, (w, h), sigma_s=sig[0], sigma_r=sig[1])
    assert ws._bilateral.grid_size() == oracle.bilateral_grid_size(w, h, *sig)
    got = npy(ws.process(gpu(lum, dev), 0.4))
    ref = oracle.bilateral(lum, sig[0], sig[1], 0.4)
    # gather splat sums in raster order and the blurs use the reference's own formulas: bit-exact
    assert np.array_equal(got, ref), f'max |d| = {np.abs(got - ref).max()}'
    assert np.array_equal(npy(ws.process(gpu(lum, dev), 0.0)), np.maximum(lum, 0.0))


@pytest.mark.parametrize('sig', [(2.0, 0.2), (1.0, 0.25), (4.0, 0.1), (2.5, 0.07), (3.0, 0.02)])
@pytest.mark.parametrize('size', [(256, 192), (324, 130), (67, 45)])
def test_bilateral_tile_kernel(td, oracle, dev, scene, sig, size):
    """Small sigma_s takes the fused LDS tile kernel (grid never in HBM); it must equal the oracle
    AND the four-kernel path bit for bit, for planes (fp32 / fp16) and for the RGB epilogues."""
    w, h = size
    img = scene(h, w, 41)
    lum = oracle.compute_luminance(img)
    ws = td.Bilateral(dev, (w, h), sigma_s=sig[0], sigma_r=sig[1])
    got = npy(ws.process(gpu(lum, dev), 0.4))
    assert np.array_equal(got, oracle.bilateral(lum, sig[0], sig[1], 0.4))
    lum16 = lum.astype(np.float16)
    got16 = npy(ws.process(gpu(lum16, dev), 0.4))
    assert np.array_equal(got16, oracle.bilateral(lum16.astype(np.float32), sig[0], sig[1], 0.4).astype(np.float16))
    rgb_fused = npy(ws.process_rgb(gpu(img, dev), 0.4))
    log_fused = npy(ws.process_log_rgb(gpu(img, dev), 0.4))
    from torch_darktable import torch_darktable_extension as ext
    with ext.verification_paths(bilateral_general=True):  # the general four-kernel path, same parameters
        assert np.array_equal(npy(ws.process(gpu(lum, dev), 0.4)), got)
        assert np.array_equal(npy(ws.process_rgb(gpu(img, dev), 0.4)), rgb_fused)
        assert np.array_equal(npy(ws.process_log_rgb(gpu(img, dev), 0.4)), log_fused)


def test_bilateral_tile_kernel_geometry_sweep(td, dev):
    """The tile kernel's per-launch axis tables and fixed sample window over many frame sizes and sigmas (tile counts that end
    in partial tiles on either axis, widths that are / are not multiples of 4, sigma_s from 1 to 4 with non-integer values,
    short and long z columns): the tile path must equal the four-kernel path bit for bit, for planes and the RGB epilogue."""
    from torch_darktable import torch_darktable_extension as ext
    rng = np.random.default_rng(99)
    cases = [(64, 32, 2.0, 0.2), (65, 33, 2.0, 0.2), (128, 64, 1.5, 0.25), (130, 70, 4.0, 0.1), (67, 200, 3.3, 0.05), (400, 37, 1.0, 0.25), (36, 40, 2.7, 0.02)]
    cases += [(int(rng.integers(40, 500)), int(rng.integers(40, 300)), float(rng.uniform(1.0, 4.0)), float(rng.choice([0.02, 0.07, 0.1, 0.2, 0.25]))) for _ in range(12)]
    for w, h, ss, sr in cases:
        lum = torch.from_numpy(rng.random((h, w), dtype=np.float32)).to(dev)
        rgb = torch.from_numpy(rng.random((h, w, 3), dtype=np.float32)).to(dev)
        ws = td.Bilateral(dev, (w, h), sigma_s=ss, sigma_r=sr)
        a, a16, argb = ws.process(lum, 0.4), ws.process(lum.half(), 0.4), ws.process_rgb(rgb, 0.4)
        with ext.verification_paths(bilateral_general=True):  # the general four-kernel path, same parameters
            b, b16, brgb = ws.process(lum, 0.4), ws.process(lum.half(), 0.4), ws.process_rgb(rgb, 0.4)
        for name, x, y in (('plane', a, b), ('plane f16', a16, b16), ('rgb', argb, brgb)):
            assert torch.equal(x, y), f'{w}x{h} sigma ({ss:.3f}, {sr}) {name}: {(x != y).sum().item()} values differ'
    # samples outside the range of the exact-quotient shortcut (zeros, denormals, huge, negative, inf, nan): the tile kernel's
    # grouped range test must fall back to the plain division exactly where the general path divides
    w, h = 256, 96
    lum = rng.random((h, w), dtype=np.float32)
    special = np.array([0.0, -0.0, 1e-45, 1e-39, 2.0 ** -41, 2.0 ** -40, 2.0 ** 40, 2.0 ** 41, 1e30, -0.3, -1e30, np.inf, -np.inf, np.nan], np.float32)
    ys, xs = rng.integers(0, h, 200), rng.integers(0, w, 200)
    lum[ys, xs] = special[rng.integers(0, len(special), 200)]
    lum_t = torch.from_numpy(lum).to(dev)
    ws = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    a = ws.process(lum_t, 0.4)
    with ext.verification_paths(bilateral_general=True):  # the general four-kernel path, same parameters
        b = ws.process(lum_t, 0.4)
    same = (a == b) | (torch.isnan(a) & torch.isnan(b))
    assert bool(same.all()), f'special values: {(~same).sum().item()} values differ'


@pytest.mark.parametrize('K,ov', [(32, 4), (32, 2), (16, 4), (16, 8), (32, 8), (16, 2)])
@pytest.mark.parametrize('C', [1, 3])
def test_wiener(td, oracle, dev, scene, K, ov, C):
    h, w = 100, 141
    img = scene(h, w, 26)[:, :, :C].copy()
    ws = td.Wiener(dev, (w, h), overlap_factor=ov, tile_size=K)
    sig = np.array([0.05, 0.08, 0.03], np.float32)[:C]
    got = npy(ws.process(gpu(img, dev), gpu(sig, dev)))
    ref = oracle.wiener(img, sig, K, ov)
    assert np.abs(got - ref).max() < 2e-5
    ident = npy(ws.process(gpu(img, dev), 0.0))
    assert np.abs(ident - img).max() < 2e-6


@pytest.mark.parametrize('size', [(32, 32), (33, 40), (40, 32)])
def test_wiener_smallest_images(td, oracle, dev, scene, size):
    """Images barely larger than one tile: every tile is an edge tile (reflect loads on all sides)."""
    h, w = size
    img = scene(h, w, 51)
    ws = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    sig = np.array([0.05, 0.08, 0.03], np.float32)
    assert np.abs(npy(ws.process(gpu(img, dev), gpu(sig, dev))) - oracle.wiener(img, sig, 32, 4)).max() < 2e-5
    l1 = img[:, :, :1].copy()
    assert np.abs(npy(ws.process(gpu(l1, dev), gpu(sig[:1], dev))) - oracle.wiener(l1, sig[:1], 32, 4)).max() < 2e-5
    with pytest.raises(RuntimeError):
        td.Wiener(dev, (31, 40), overlap_factor=4, tile_size=32).process(gpu(scene(40, 31, 1), dev), gpu(sig, dev))


@pytest.mark.parametrize('size', [(5, 7), (9, 13), (16, 4), (3, 64)])
@pytest.mark.parametrize('sig', [(2.0, 0.2), (1.0, 0.1), (8.0, 0.1)])
def test_bilateral_tiny_images(td, oracle, dev, scene, size, sig):
    h, w = size
    lum = oracle.compute_luminance(scene(h, w, 52))
    ws = td.Bilateral(dev, (w, h), sigma_s=sig[0], sigma_r=sig[1])
    assert np.array_equal(npy(ws.process(gpu(lum, dev), 0.5)), oracle.bilateral(lum, sig[0], sig[1], 0.5))


@pytest.mark.parametrize('size', [(96, 128), (50, 77), (33, 70)])
def test_wiener_log_luminance_pipeline(td, oracle, dev, scene, size):
    """Fused extract -> tiles -> finish+modify; odd widths take the scalar (non-vector) epilogue.
    The fused call must also equal the three-call chain it replaces."""
    h, w = size
    img = scene(h, w, 27)
    ws = td.Wiener(dev, (w, h))
    x = gpu(img, dev)
    got = npy(ws.process_log_luminance(x, 0.075))
    ll = oracle.compute_luminance(img, True, 1e-4)
    ref = oracle.modify_luminance(img, oracle.wiener(ll[:, :, None], 0.075)[:, :, 0], True)
    assert np.abs(got - ref).max() < 2e-5   # measured 4.6e-6 (profiles/r03/bounds_probe.json)
    chain = td.modify_log_luminance(x, ws.process(td.compute_log_luminance(x, 1e-4).unsqueeze(2), 0.075).squeeze(2), 1e-4)
    assert np.abs(got - npy(chain)).max() < 2e-6


@pytest.mark.parametrize('size', [(120, 161), (4, 4), (7, 9), (16, 16), (33, 70), (256, 200), (301, 515), (600, 1100), (5, 301), (1000, 9), (64, 2050)])
def test_laplacian_level_schedules(td, oracle, dev, scene, size):
    """Image sizes that take every branch of the launch schedule in csrc/laplacian.hip: two levels only, everything
    inside the single-workgroup kernels, a single reduce launch, a reduce pair, tiled assembles."""
    h, w = size
    lum = oracle.compute_luminance(scene(max(h, 8), max(w, 8), 31))[:h, :w].copy()
    prm = (0.25, 1.4, 0.8, 0.2)
    got = npy(td.Laplacian(dev, (w, h), td.LaplacianParams(6, *prm)).process(gpu(lum, dev)))
    ref = oracle.laplacian(lum, *prm)
    d = np.abs(got - ref)
    # the curve is evaluated in the oracle's operation order (csrc/laplacian.hip): measured bit-identical on every schedule
    # (profiles/r04/bounds_probe.json).  The clarity term goes through the hardware exp2 where the oracle calls libm, so a rare
    # binary16 rounding flip stays allowed: at most 1 binary16 ulp of the value on at most 1e-4 of the pixels
    assert np.isfinite(got).all() and (d <= 1.0 * half_ulp(np.maximum(np.abs(got), np.abs(ref)))).all() and (d > 0).mean() <= 1e-4, (size, d.max(), (d > 0).mean())


@pytest.mark.parametrize('prm', [(0.2, 1.0, 1.0, 0.0), (0.2, 1.6, 0.7, 0.3), (0.35, 0.5, 1.5, -0.2)])
def test_laplacian(td, oracle, dev, scene, prm):
    h, w = 120, 161
    lum = oracle.compute_luminance(scene(h, w, 28))
    ws = td.Laplacian(dev, (w, h), td.LaplacianParams(6, *prm))
    got = npy(ws.process(gpu(lum, dev)))
    ref = oracle.laplacian(lum, *prm)
    # fp16 storage at every level, fp32 math in the oracle's operation order: without the clarity term the result is the
    # oracle's bit for bit; with it (hardware exp2 against libm expf) a rare binary16 rounding flip is allowed: 1 binary16 ulp
    # of the value on at most 1e-4 of the pixels (measured: none, profiles/r04/bounds_probe.json)
    d = np.abs(got - ref)
    if prm[3] == 0.0:
        assert np.array_equal(got, ref), (d.max(), (d > 0).mean())
    else:
        assert (d <= 1.0 * half_ulp(np.maximum(np.abs(got), np.abs(ref)))).all() and (d > 0).mean() <= 1e-4, (d.max(), (d > 0).mean())
    with pytest.raises(RuntimeError):
        td.Laplacian(dev, (w, h), td.LaplacianParams(num_gamma=4))


# ------------------------------------------------------------------ fp16 storage (extension; 2e-3 relative)
def test_fp16_storage_pipeline(td, oracle, dev, scene):
    h, w = 128, 160
    bayer = oracle.mosaic(scene(h, w, 29), oracle.RGGB)
    b16 = gpu(bayer, dev).half()
    rgb16 = td.RCD(dev, (w, h), td.BayerPattern.RGGB).process(b16)
    assert rgb16.dtype == torch.float16
    ref = oracle.rcd(npy(b16).astype(np.float32), oracle.RGGB)
    assert_f16_close(npy(rgb16), ref, 'RCD f16')  # small frame: the tile kernel, fp32 math and one rounding at the store
    m = td.compute_image_metrics([rgb16], 8)
    u8 = td.reinhard_tonemap(rgb16, m, td.TonemapParameters(0.75, 2.0, 1.0, 0.0))
    ref_u8 = oracle.tonemap('reinhard', npy(rgb16).astype(np.float32), npy(m), 0.75, 2.0, 1.0, 0.0)
    assert np.abs(npy(u8).astype(np.int32) - ref_u8.astype(np.int32)).max() <= 1


def test_normalize_image_kernel(td, dev):
    from torch_darktable.pipeline.util import normalize_image
    g = torch.Generator().manual_seed(5)
    for shape in [(37, 53, 3), (4, 4, 3), (1, 3, 3), (128, 256, 3)]:
        x = (torch.rand(shape, generator=g) * 3 - 0.5).to(dev)
        b = torch.tensor([-0.37, 2.11], device=dev)
        assert torch.equal(normalize_image(x, b), (x - b[0]) / (b[1] - b[0]))
        xh = x.half()
        ref = ((xh.float() - b[0]) / (b[1] - b[0])).half()
        assert torch.equal(normalize_image(xh, b), ref)


# ------------------------------------------------------------------ pipeline (caller + on-disk format)
def test_image_processor_end_to_end(td, oracle, dev, scene, tmp_path):
    """packed 12-bit raw FILE with trailing padding -> ImageProcessor -> uint8, against the oracle
    chain decode -> white balance -> RCD -> postprocess -> normalise -> Wiener -> bilateral -> ACES."""
    from torch_darktable.pipeline import CameraSettings, ImageProcessingSettings, ImageProcessor, ImageTransform, ToneMapper
    from torch_darktable.pipeline.camera_settings import load_raw_bytes

    h, w, pad = 96, 128, 64
    bayer = np.clip(oracle.mosaic(scene(h, w, 31), oracle.RGGB)[:, :, 0], 0, 1)
    packed = oracle.encode12_f32(bayer.ravel(), False, True)
    raw_file = tmp_path / 'cam' / 'frame0.raw'
    raw_file.parent.mkdir()
    raw_file.write_bytes(packed.tobytes() + bytes(pad))
    settings = ImageProcessingSettings(tone_gamma=2.2, tone_intensity=1.0, moving_average=1.0, postprocess=True, enable_bilateral=True,
                                       tone_mapping=ToneMapper.aces, vibrance=0.3)
    cam = CameraSettings(name='cam', image_size=(w, h), padding=pad, white_balance=(1.5, 1.0, 1.2), image_processing=settings,
                         transform=ImageTransform.rotate_270)
    assert raw_file.stat().st_size == cam.bytes
    proc = ImageProcessor.from_camera_settings(cam, dev)
    out = proc.process(load_raw_bytes(raw_file, dev), 'cam')
    assert out.dtype == torch.uint8 and tuple(out.shape) == (w, h, 3)  # rotated

    b = oracle.decode12_f32(packed, False, True).reshape(h, w)
    b = oracle.apply_white_balance(b, [1.5, 1.0, 1.2], oracle.RGGB)
    rgb = oracle.postprocess(oracle.rcd(b, oracle.RGGB), oracle.RGGB, 3, False, True, 0.04)
    bounds = oracle.image_bounds([rgb], 8)
    rgb = (rgb - bounds[0]) / (bounds[1] - bounds[0])
    ll = oracle.compute_luminance(rgb, True, 1e-4)
    rgb = oracle.modify_luminance(rgb, oracle.wiener(ll[:, :, None], 0.075, 32, 4)[:, :, 0], True)
    rgb = oracle.modify_luminance(rgb, oracle.bilateral(oracle.compute_luminance(rgb), 2.0, 0.2, 0.4))
    ref = oracle.tonemap('aces', rgb, None, 2.2, 1.0, 1.0, 0.3)
    ref = np.rot90(ref, 3, (0, 1))
    d = np.abs(npy(out).astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02, f'max {d.max()} frac {(d > 0).mean()}'
    # fp16 storage between the stages (fp32 arithmetic): same picture within 1 LSB, except a handful of pixels where a
    # rounded stage output flips a selection in a later stage (colour-smoothing medians, bilateral cell): measured 5 of
    # 36 864 values at 2-4 LSB
    proc16 = ImageProcessor.from_camera_settings(cam, dev, storage_dtype=torch.float16)
    out16 = proc16.process(load_raw_bytes(raw_file, dev), 'cam')
    d16 = np.abs(npy(out16).astype(np.int32) - ref.astype(np.int32))
    assert out16.dtype == torch.uint8 and d16.max() <= 4 and (d16 > 1).mean() < 5e-4, f'max {d16.max()} hist {np.bincount(d16.ravel())}'

    with pytest.raises(Exception) as ei:
        proc.process(load_raw_bytes(raw_file, dev)[:-1], 'cam')
    assert 'mismatch' in str(ei.value)
    # moving average: a second identical frame leaves the statistics unchanged
    m0 = proc.metrics.clone()
    proc.process(load_raw_bytes(raw_file, dev), 'cam')
    assert torch.allclose(proc.metrics, m0)


def test_raw_frame_stream_prefetch(td, dev, tmp_path):
    """Pinned-ring / copy-stream prefetcher: frames arrive in order and intact from
    files, bytes and tensors; wrong sizes are reported; abandoning the iterator does not hang."""
    from torch_darktable.pipeline import RawFrameStream

    n, nb = 7, 96 * 128 * 3 // 2 + 64
    rng = np.random.default_rng(7)
    blobs = [rng.integers(0, 256, nb, dtype=np.uint8) for _ in range(n)]
    paths = []
    for i, b in enumerate(blobs):
        p = tmp_path / f'frame{i}.raw'
        p.write_bytes(b.tobytes())
        paths.append(p)
    sources = [paths[0], blobs[1].tobytes(), torch.from_numpy(blobs[2]), *paths[3:]]
    for depth in (1, 2, 4):
        got = []
        for frame in RawFrameStream(sources, dev, nb, depth=depth):
            assert frame.device == dev and frame.dtype == torch.uint8
            got.append(frame.clone())  # consumer-stream work on the yielded buffer
        torch.cuda.synchronize()
        assert len(got) == n and all(np.array_equal(npy(g), b) for g, b in zip(got, blobs))
    it = iter(RawFrameStream(paths, dev, nb, depth=2))
    assert np.array_equal(npy(next(it)), blobs[0])
    it.close()  # early exit: reader thread must shut down
    with pytest.raises(ValueError):
        list(RawFrameStream([paths[0], b'short'], dev, nb))
    with pytest.raises(ValueError):
        RawFrameStream(paths, torch.device('cpu'), nb)
