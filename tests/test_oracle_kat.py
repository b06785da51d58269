"""Pins the CPU oracle (CPU-only tests).

The reference has no golden vectors or known-answer tests for any kernel op (its one test is a
pydantic round-trip), and its CUDA build cannot run here, so each op's parity is "unpinned" by
the reference.  The oracle is instead pinned by
  (a) fixtures generated from the reference's own importable pure-torch helpers
      (tests/golden/make_golden.py -> reference_helpers.npz), and
  (b) identities derived from the reference source: hand-computed byte
      triples, constant-image fixed points, identity parameter settings, closed forms.
"""

from pathlib import Path

import numpy as np
import pytest

GOLD = np.load(Path(__file__).parent / 'golden' / 'reference_helpers.npz')
PATS = ['RGGB', 'BGGR', 'GRBG', 'GBRG']


# ------------------------------------------------------------------ (a) reference-generated fixtures
@pytest.mark.parametrize('pat', PATS)
@pytest.mark.parametrize('tag', ['small', 'mid'])
def test_rgb_to_bayer_matches_reference(oracle, pat, tag):
    got = oracle.rgb_to_bayer(GOLD[f'rgb_{tag}'], oracle.PATTERNS[pat])
    assert np.array_equal(got, GOLD[f'bayer_{tag}_{pat}'])


def test_fc_agrees_with_reference_tables(oracle):
    """bayer.py pixel_order: site class (R, G1, G2, B) -> colour fc() must give 0/1/1/2."""
    colour_of_class = {0: 0, 1: 1, 2: 1, 3: 2}
    for pat in PATS:
        order = GOLD[f'pixel_order_{pat}']
        for pos in range(4):
            assert oracle.cfa_color(pos // 2, pos % 2, oracle.PATTERNS[pat]) == colour_of_class[int(order[pos])]
    # and fc() has period 2 in both axes
    r, c = np.mgrid[0:6, 0:6]
    for pat in PATS:
        m = oracle.cfa_color(r, c, oracle.PATTERNS[pat])
        assert np.array_equal(m, np.tile(m[:2, :2], (3, 3))) and set(np.unique(m)) == {0, 1, 2}


def test_mosaic_equals_reference_for_rggb_bggr(oracle):
    """The reference's channel table is right for RGGB / BGGR, where mosaic() must agree with it."""
    rgb = GOLD['rgb_mid']
    for pat in ('RGGB', 'BGGR'):
        assert np.array_equal(oracle.mosaic(rgb, oracle.PATTERNS[pat]), GOLD[f'bayer_mid_{pat}'])


# ------------------------------------------------------------------ (b1) codec
def test_codec_hand_computed_triples(oracle):
    # p0 = 0xABC, p1 = 0x123: standard b0 = 0xBC, b1 = (0x3 << 4) | 0xA, b2 = 0x12 (packed.cu:8-12)
    assert oracle.encode12_u16(np.array([0xABC, 0x123], np.uint16)).tolist() == [0xBC, 0x3A, 0x12]
    assert oracle.decode12_u16(np.array([0xBC, 0x3A, 0x12], np.uint8)).tolist() == [0xABC, 0x123]
    # IDS encode: b0 = p0 >> 4, b1 = p1 >> 4, b2 = (p0 & 0xf) << 4 | (p1 & 0xf) (packed.cu:20-24)
    assert oracle.encode12_u16(np.array([0xABC, 0x123], np.uint16), ids=True).tolist() == [0xAB, 0x12, 0xC3]
    # IDS decode takes p0's nibble from the LOW half of byte 2 (packed.cu:27-31)
    assert oracle.decode12_u16(np.array([0xAB, 0x12, 0xC3], np.uint8), ids=True).tolist() == [0xAB3, 0x12C]


def test_codec_roundtrip_and_clamps(oracle):
    x = np.arange(4096, dtype=np.uint16)
    assert np.array_equal(oracle.decode12_u16(oracle.encode12_u16(x)), x)
    assert np.array_equal(oracle.decode12_u16(oracle.encode12_u16(np.array([5000, 65535], np.uint16))), [4095, 4095])
    f = oracle.decode12_f32(oracle.encode12_u16(x))
    assert f[0] == 0.0 and f[4095] == np.float32(4095) * np.float32(1.0 / 4095.0)
    assert np.array_equal(oracle.decode12_u16(oracle.encode12_f32(f)), x)  # scaled round trip is exact
    assert oracle.decode12_u16(oracle.encode12_f32(np.array([-1.0, 2.0], np.float32))).tolist() == [0, 4095]
    assert np.array_equal(oracle.decode12_f16(oracle.encode12_u16(x)), f.astype(np.float16))
    assert oracle.encode12_u16(np.zeros(0, np.uint16)).size == 0


# ------------------------------------------------------------------ (b2) demosaic
@pytest.mark.parametrize('pat', PATS)
def test_demosaic_of_constant_cfa(oracle, pat):
    p = oracle.PATTERNS[pat]
    for c in (0.25, 0.5):
        bayer = np.full((40, 48, 1), c, np.float32)
        assert np.array_equal(oracle.bilinear5x5(bayer, p), np.full((40, 48, 3), c, np.float32))
        assert np.array_equal(oracle.ppg(bayer, p), np.full((40, 48, 3), c, np.float32))
        rcd = oracle.rcd(bayer, p)
        assert np.abs(rcd - c).max() < 4e-6 * c / 0.25 + 2e-6   # eps = 1e-5 in the ratio estimates
        assert np.abs(rcd[:7] - c).max() <= np.spacing(np.float32(c))  # border ring: 3-sample averages


@pytest.mark.parametrize('pat', PATS)
def test_native_sample_is_preserved(oracle, scene, pat):
    p = oracle.PATTERNS[pat]
    rgb = scene(48, 64, 3)
    bayer = oracle.mosaic(rgb, p)
    rows, cols = np.mgrid[0:48, 0:64]
    ch = oracle.cfa_color(rows, cols, p)
    for fn in (oracle.ppg, oracle.rcd):
        out = fn(bayer, p)
        native = np.take_along_axis(out, ch[:, :, None], 2)
        assert np.array_equal(native, np.maximum(bayer, 0))
    out = oracle.bilinear5x5(bayer, p)
    if pat in ('RGGB', 'GRBG'):  # BGGR / GBRG: the reference's site-class table swaps the green rows
        assert np.array_equal(np.take_along_axis(out, ch[:, :, None], 2), bayer)


def _pq_taps(r, c):
    """P and Q taps of step 4.2 at R/B site (r, c) in 2-D coordinates (derived from the
    flat `idx / 2` slot arithmetic of rcd.cu:157,173-180 for an even width): an even idx shares its slot with
    idx + 1 and step 4.1 only writes odd columns, so the three taps are NOT a symmetric diagonal."""
    if c % 2 == 0:
        return [(r - 1, c - 1), (r, c + 1), (r + 1, c + 1)], [(r - 1, c + 1), (r, c + 1), (r + 1, c - 1)]
    return [(r - 1, c), (r, c), (r + 1, c + 2)], [(r - 1, c + 2), (r, c), (r + 1, c)]


def _pq_dir_2d(cfa, r, c):
    """Independent 2-D restatement of steps 4.1 + 4.2 for one site (rcd.cu:149-182), float32, same operation order."""
    f = np.float32

    def p_diff(y, x):  # NW-SE diagonal high-pass, squared
        d = (cfa[y - 3, x - 3] - cfa[y - 1, x - 1] - cfa[y + 1, x + 1] + cfa[y + 3, x + 3]) - f(3) * (cfa[y - 2, x - 2] + cfa[y + 2, x + 2]) + f(6) * cfa[y, x]
        return d * d

    def q_diff(y, x):  # NE-SW diagonal
        d = (cfa[y - 3, x + 3] - cfa[y - 1, x + 1] - cfa[y + 1, x - 1] + cfa[y + 3, x - 3]) - f(3) * (cfa[y - 2, x + 2] + cfa[y + 2, x - 2]) + f(6) * cfa[y, x]
        return d * d

    pt, qt = _pq_taps(r, c)
    P = np.maximum(f(1e-10), p_diff(*pt[0]) + p_diff(*pt[1]) + p_diff(*pt[2]))
    Q = np.maximum(f(1e-10), q_diff(*qt[0]) + q_diff(*qt[1]) + q_diff(*qt[2]))
    return P / (P + Q)


@pytest.mark.parametrize('W', [16, 24])
def test_rcd_slot_aliasing_known_answer(oracle, W):
    """Known-answer test of the one RCD quirk a clean reimplementation would silently 'fix'
    (rcd.cu:166-182): PQ_dir at an R/B site is built from p/q slots addressed as flat idx / 2.
    For W = 16, site (4,4): P <- (3,3),(4,5),(5,5), Q <- (3,5),(4,5),(5,3); site (5,5):
    P <- (4,5),(5,5),(6,7), Q <- (4,7),(5,5),(6,5).  The expected values come from an
    independent 2-D restatement with those tap tables, not from flat-index code."""
    H = 20
    rng = np.random.default_rng(W)
    cfa = rng.uniform(0.05, 0.95, (H, W)).astype(np.float32)
    _, pq, _, _ = oracle.rcd_planes(cfa, oracle.RGGB)
    if W == 16:
        assert _pq_taps(4, 4) == ([(3, 3), (4, 5), (5, 5)], [(3, 5), (4, 5), (5, 3)])
        assert _pq_taps(5, 5) == ([(4, 5), (5, 5), (6, 7)], [(4, 7), (5, 5), (6, 5)])
    checked = 0
    for r in range(4, H - 4):
        for c in range(4 + (r & 1), W - 4, 2):  # RGGB: R at even/even, B at odd/odd
            pt, qt = _pq_taps(r, c)
            if not all(3 <= y <= H - 4 and 3 <= x <= W - 4 for y, x in pt + qt):
                continue  # a tap outside step 4.1's write region reads stale v/h_diff (covered by the GPU == oracle tests)
            assert pq[(r * W + c) // 2] == _pq_dir_2d(cfa, r, c), (r, c)
            checked += 1
    assert checked >= 20

    # perturbation: cfa(7,8) lies on the NW-SE diagonal of (4,5) only among the taps of site (4,4) --
    # PQ_dir(4,4) must move (aliased tap (4,5)); a symmetric P diagonal (3,3),(4,4),(5,5) would not see it
    bumped = cfa.copy()
    bumped[7, 8] += np.float32(0.25)
    _, pq2, _, _ = oracle.rcd_planes(bumped, oracle.RGGB)
    assert pq2[(4 * W + 4) // 2] != pq[(4 * W + 4) // 2]
    assert pq2[(4 * W + 4) // 2] == _pq_dir_2d(bumped, 4, 4)
    # ...cfa(8,8), on the diagonal of tap (5,5), also moves it; cfa(10,2), on none of the six taps' diagonals
    # (y - x in {0, -1}, y + x in {8, 9} within 3 steps of a tap), does not
    for (y, x), moves in (((8, 8), True), ((10, 2), False)):
        b2 = cfa.copy()
        b2[y, x] += np.float32(0.25)
        _, pq3, _, _ = oracle.rcd_planes(b2, oracle.RGGB)
        assert (pq3[(4 * W + 4) // 2] != pq[(4 * W + 4) // 2]) == moves, (y, x)
    with pytest.raises(ValueError):
        oracle.rcd(np.zeros((16, 15, 1), np.float32), oracle.RGGB)


def test_border_interpolate_ring_only(oracle, scene):
    bayer = oracle.mosaic(scene(20, 24, 4), oracle.RGGB)
    out = oracle.border_interpolate(bayer, oracle.RGGB, 3)
    assert np.all(out[3:-3, 3:-3] == 0) and np.all(out[:3].sum(-1) > 0)
    # corner pixel (0,0) is red: R = own, G = mean of (0,1),(1,0), B = (1,1)
    b = bayer[:, :, 0]
    assert out[0, 0, 0] == b[0, 0] and out[0, 0, 2] == b[1, 1]
    assert out[0, 0, 1] == np.float32(np.float32(b[0, 1] + b[1, 0]) / np.float32(2))


# ------------------------------------------------------------------ (b3) postprocess
def test_postprocess_identities(oracle, scene):
    rgb = scene(32, 40, 5)
    assert np.array_equal(oracle.postprocess(rgb, oracle.RGGB), rgb)
    grey = np.repeat(rgb[:, :, 1:2], 3, 2)
    assert np.array_equal(oracle.postprocess(grey, oracle.RGGB, 3), grey)  # R-G = B-G = 0 everywhere
    s32, s64 = oracle.green_eq_sums(rgb, oracle.RGGB)
    g = rgb[:, :, 1].astype(np.float64)
    assert np.isclose(s64[0], g[0::2, 1::2].sum(), rtol=1e-6) and np.isclose(s64[1], g[1::2, 0::2].sum(), rtol=1e-6)
    assert np.allclose(s32, s64, rtol=1e-5)


# ------------------------------------------------------------------ (b4) Wiener
@pytest.mark.parametrize('K,ov', [(16, 4), (32, 4), (16, 2), (32, 8)])
def test_wiener_identities(oracle, scene, K, ov):
    img = scene(72, 88, 6)[:, :, :1]
    assert np.abs(oracle.wiener(img, 0.0, K, ov) - img).max() < 2e-6          # sigma = 0: identity
    const = np.full((72, 88, 3), 0.3, np.float32)
    assert np.abs(oracle.wiener(const, 0.5, K, ov) - 0.3).max() < 5e-7       # constant: fixed point
    big = oracle.wiener(img, 1e3, K, ov)                                       # sigma -> inf: blend of tile means
    assert big.std() < img.std() * 0.5 and abs(big.mean() - img.mean()) < 0.02


def test_wiener_window_values(oracle):
    w = oracle.wiener_window(32)
    assert abs(float(w[0]) - 0.013216) < 1e-6 and abs(float(w[15]) - 0.300795) < 1e-6
    assert abs(float((w.astype(np.float64) ** 2).sum()) - 1.0) < 1e-6 and np.array_equal(w, w[::-1])


# ------------------------------------------------------------------ (b5) bilateral
def test_bilateral_identities(oracle, scene):
    lum = oracle.compute_luminance(scene(60, 84, 7))
    assert np.array_equal(oracle.bilateral(lum, 2.0, 0.2, 0.0), np.maximum(lum, 0))
    const = np.full((60, 84), 0.4, np.float32)
    interior = oracle.bilateral(const, 2.0, 0.2, 0.4)[12:-12, 12:-12]
    assert np.abs(interior - interior[0, 0]).max() < 1e-6   # translation invariant away from the edges
    assert oracle.bilateral_grid_size(4096, 3072, 2.0, 0.2) == (2049, 1537, 6)
    assert oracle.bilateral_grid_size(4096, 3072, 8.0, 0.1) == (513, 385, 11)
    assert oracle.bilateral_grid_size(8192, 6144, 2.0, 0.2) == (3001, 2251, 6)
    assert oracle.bilateral_grid_size(512, 512, 2.0, 0.2) == (257, 257, 6)


# ------------------------------------------------------------------ (b6) Laplacian
def test_laplacian_identity_up_to_fp16(oracle, scene):
    lum = oracle.compute_luminance(scene(64, 80, 8))
    out = oracle.laplacian(lum, 0.2, 1.0, 1.0, 0.0)
    assert np.abs(out - lum).max() < 1.5e-3   # binary16 storage at every level
    assert oracle.lib().oracle_laplacian_levels(4096, 3072) == 11 and oracle.lib().oracle_laplacian_levels(512, 512) == 9


# ------------------------------------------------------------------ (b7) colour
def test_colour_identities(oracle, scene):
    rgb = scene(24, 32, 9)
    back = oracle.color_op('lab_to_rgb', oracle.color_op('rgb_to_lab', rgb))
    assert np.abs(back - rgb).max() < 5e-6
    white = oracle.color_op('rgb_to_lab', np.ones((1, 1, 3), np.float32))[0, 0]
    assert np.abs(white - [1, 0, 0]).max() < 1e-4
    grey = np.full((4, 4, 3), 0.35, np.float32)
    L = oracle.color_op('rgb_to_lab', grey)[0, 0, 0]
    assert abs(oracle.compute_luminance(grey)[0, 0] - L) < 1e-6
    assert np.allclose(oracle.compute_luminance(grey, log=True, eps=1e-4), np.log(L), atol=1e-6)
    assert np.array_equal(oracle.color_op('color_transform_3x3', rgb, np.eye(3, dtype=np.float32)), np.clip(rgb, 0, 1))
    assert np.abs(oracle.color_op('modify_vibrance', rgb, [0.0]) - rgb).max() < 5e-6
    assert np.abs(oracle.color_op('modify_hsl', rgb, [0, 0, 0]) - rgb).max() < 5e-6
    same = oracle.modify_luminance(rgb, oracle.compute_luminance(rgb))
    assert np.abs(same - rgb).max() < 1e-5


# ------------------------------------------------------------------ (b8) tonemaps
def test_tonemap_closed_forms(oracle):
    px = np.array([[[0.2, 0.4, 0.8]]], np.float32)
    metrics = np.array([0.0, 0.5, 0.3, 0.3, 0.3], np.float32)  # log_mean 0 -> map_key 0.3
    # light_adapt = 0, intensity = 0: adapt = global_mean ^ 0.3
    _, f = oracle.tonemap('reinhard', px, metrics, 1.0, 0.0, 0.0, 0.0, return_float=True)
    adapt = np.float32(0.3) ** np.float32(0.3)
    assert np.allclose(f[0, 0], px[0, 0] / (adapt + px[0, 0]), atol=3e-6)
    _, f = oracle.tonemap('linear', px * 10, metrics, 1.0, 0.0, 0.0, 0.0, return_float=True)
    assert np.all(f <= 1.0) and f[0, 0, 2] == 1.0
    # u8 rounding: roundf half away from zero, clamp at 255 (device_math.h:347-349)
    u8 = oracle.tonemap('linear', np.array([[[0.0, 0.5 / 255 * adapt, 5.0]]], np.float32), metrics, 1.0, 0.0, 0.0, 0.0)
    assert u8[0, 0, 0] == 0 and u8[0, 0, 2] == 255 and u8[0, 0, 1] in (0, 1)
    assert oracle.tonemap('aces', np.zeros((1, 1, 3), np.float32), None, 1.0, 0.0, 0.8, 0.0).max() == 0


# ------------------------------------------------------------------ (b9) metrics
def test_metrics_closed_forms(oracle):
    img = np.empty((32, 32, 3), np.float32)
    img[:] = [0.2, 0.4, 0.6]
    m = oracle.image_metrics([img], 8)
    gray = np.float32(0.2) * np.float32(0.299) + np.float32(0.4) * np.float32(0.587) + np.float32(0.6) * np.float32(0.114)
    assert np.allclose(m, [np.log(gray), gray, 0.2, 0.4, 0.6], atol=1e-6)
    assert np.array_equal(oracle.image_metrics([np.ones((16, 16, 3), np.float32)], 4), np.zeros(5, np.float32))
    assert oracle.image_bounds([img], 8).tolist() == [np.float32(0.2), np.float32(0.6)]
    assert oracle.apply_white_balance(np.full((4, 4), 0.5, np.float32), [2.5, 1.0, 0.5], oracle.RGGB)[:2, :2].tolist() == [[1.0, 0.5], [0.5, 0.25]]
