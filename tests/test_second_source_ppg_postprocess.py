"""CPU-only: independent restatements of PPG and of the PostProcess passes on 2-D numpy arrays, against the C oracle.

oracle/src/ppg.c and postprocess.c restate reference csrc/debayer/ppg.cu and postprocess.cu pixel by pixel in C.  This file
restates them a SECOND time from the reference sources in another style -- whole-image numpy arrays, taps as (d_row, d_col)
shifts of a zero-padded plane, the 3x3 medians by `numpy.sort` instead of a sorting network -- so that a misreading shared
by the oracle and the HIP kernels (which share an author) has to survive a second derivation.  numpy float32 arithmetic is
IEEE and uncontracted like the oracle's, and every expression keeps the reference's association: the comparisons are
BIT FOR BIT, except the global green equilibration, whose sum order the reference leaves to torch.sum (bound derived below).

What is restated (reference lines):
  * the CFA colour function fc() for the four BayerPattern words (debayer/bayer_device.h:10-12, demosaic.h:7-12);
  * border_interpolate, border 3 (ppg.cu:342-389): 3x3 same-colour averages of max(0, sample), own colour kept;
  * pre_median (ppg.cu:21-108): nine taps on the 5x5 diamond, out-of-threshold taps lifted by 64, sort, pick, clamped move
    -- green sites only, threshold = median_threshold / 100 (ppg.cu:441);
  * green at red / blue sites (ppg.cu:115-221), red / blue fill (ppg.cu:227-335), with the pass-through of the outermost ring;
  * colour smoothing: 3x3 median of R-G and B-G over zero-extended neighbours (postprocess.cu:25-76, reduction.h:131-144);
  * local green equilibration at odd-row greens (postprocess.cu:82-157), global equilibration (postprocess.cu:163-262, 335-357).
"""

import numpy as np
import pytest

f32 = np.float32
WORDS = {'RGGB': 0x94949494, 'BGGR': 0x16161616, 'GRBG': 0x61616161, 'GBRG': 0x49494949}


def fc_map(h, w, word):
    r, c = np.mgrid[0:h, 0:w]
    return (word >> ((((r << 1) & 14) + (c & 1)) << 1)) & 3


def padded(a, n):
    """zero extension by n pixels; taps are then plain slices"""
    return np.pad(a, ((n, n), (n, n)) + ((0, 0),) * (a.ndim - 2))


def tap(p, n, dr, dc, h, w):
    return p[n + dr:n + dr + h, n + dc:n + dc + w]


# ------------------------------------------------------------------------------------------------ PPG
def border_interpolate(raw, color, border):
    h, w = raw.shape
    pos = np.maximum(f32(0), raw)
    pp, pc = padded(pos, 1), padded(color + 1, 1)  # colour code + 1; 0 marks "outside the frame"
    s = [np.zeros((h, w), f32) for _ in range(4)]
    n = [np.zeros((h, w), np.int32) for _ in range(4)]
    for dr in (-1, 0, 1):          # the reference's loop order: j (rows) outer, i (columns) inner
        for dc in (-1, 0, 1):
            v, k = tap(pp, 1, dr, dc, h, w), tap(pc, 1, dr, dc, h, w)
            for f in range(4):
                hit = k == f + 1
                s[f] = np.where(hit, s[f] + v, s[f]).astype(f32)
                n[f] = n[f] + hit
    with np.errstate(all='ignore'):
        r = np.where(n[0] > 0, s[0] / n[0].astype(f32), pos)
        g = np.where(n[1] + n[3] > 0, (s[1] + s[3]) / (n[1] + n[3]).astype(f32), pos)
        b = np.where(n[2] > 0, s[2] / n[2].astype(f32), pos)
    out = np.stack([np.where(color == 0, pos, r), np.where((color == 1) | (color == 3), pos, g), np.where(color == 2, pos, b)], -1).astype(f32)
    rr, cc = np.mgrid[0:h, 0:w]
    ring = ~((cc >= border) & (cc < w - border) & (rr >= border) & (rr < h - border))
    return out, ring


def pre_median(raw, color, thr):
    h, w = raw.shape
    p = padded(raw, 2)
    taps = [(-2, 0), (-1, -1), (-1, 1), (0, -2), (0, 0), (0, 2), (1, -1), (1, 1), (2, 0)]  # i = 0 .. 4, j = -lim .. lim step 2
    center = raw
    vals = np.stack([tap(p, 2, dr, dc, h, w) for dr, dc in taps], -1)
    near = np.abs(vals - center[:, :, None]) < thr
    cnt = near.sum(-1)
    med = np.sort(np.where(near, vals, f32(64) + vals).astype(f32), -1)
    pick = np.take_along_axis(med, ((cnt - 1) // 2).clip(0, 8)[:, :, None], -1)[:, :, 0]
    target = np.where(cnt == 1, med[:, :, 4] - f32(64), pick).astype(f32)
    moved = center + np.minimum(np.maximum(target - center, -thr), thr)
    return np.maximum(np.where((color & 1) == 1, moved, center), f32(0)).astype(f32)


def ppg_green(src, color, temp):
    """writes the interior (margin 3) of temp: own sample in its channel, green interpolated at red / blue sites"""
    h, w = src.shape
    p = padded(src, 3)
    t = lambda dr, dc: tap(p, 3, dr, dc, h, w)
    pc = src
    two, three = f32(2), f32(3)

    def guess_diff(d):
        m1, m2, m3 = t(-d[0], -d[1]), t(-2 * d[0], -2 * d[1]), t(-3 * d[0], -3 * d[1])
        M1, M2, M3 = t(d[0], d[1]), t(2 * d[0], 2 * d[1]), t(3 * d[0], 3 * d[1])
        guess = (m1 + pc + M1) * two - M2 - m2
        diff = (np.abs(m2 - pc) + np.abs(M2 - pc) + np.abs(m1 - M1)) * three + (np.abs(M3 - M1) + np.abs(m3 - m1)) * two
        return guess, diff, np.minimum(m1, M1), np.maximum(m1, M1)

    gx, dx, mx, Mx = guess_diff((0, 1))
    gy, dy, my, My = guess_diff((1, 0))
    use_y = dx > dy
    green = np.where(use_y, np.maximum(np.minimum(gy * f32(0.25), My), my), np.maximum(np.minimum(gx * f32(0.25), Mx), mx)).astype(f32)
    col = np.zeros((h, w, 3), f32)
    col[:, :, 0] = np.where(color == 0, pc, 0)
    col[:, :, 1] = np.where((color == 1) | (color == 3), pc, np.where((color == 0) | (color == 2), green, 0))
    col[:, :, 2] = np.where(color == 2, pc, 0)
    rr, cc = np.mgrid[0:h, 0:w]
    inner = (cc >= 3) & (cc < w - 3) & (rr >= 3) & (rr < h - 3)
    return np.where(inner[:, :, None], np.maximum(col, f32(0)), temp).astype(f32)


def ppg_redblue(temp, color, word):
    h, w, _ = temp.shape
    p = padded(temp, 1)
    t = lambda dr, dc: tap(p, 1, dr, dc, h, w)
    c = temp
    two, half, quarter = f32(2), f32(0.5), f32(0.25)
    nt, nb, nl, nr = t(-1, 0), t(1, 0), t(0, -1), t(0, 1)
    ntl, ntr, nbl, nbr = t(-1, -1), t(-1, 1), t(1, -1), t(1, 1)
    g = c[:, :, 1]
    rr, cc = np.mgrid[0:h, 0:w]
    red_right = ((word >> ((((rr << 1) & 14) + ((cc + 1) & 1)) << 1)) & 3) == 0  # fc(row, col + 1) == 0

    def cross(a, b, k):  # (a.k + b.k + 2 g - a.y - b.y) / 2
        return (a[:, :, k] + b[:, :, k] + two * g - a[:, :, 1] - b[:, :, 1]) * half

    def star(k):
        d1 = np.abs(ntl[:, :, k] - nbr[:, :, k]) + np.abs(ntl[:, :, 1] - g) + np.abs(nbr[:, :, 1] - g)
        g1 = ntl[:, :, k] + nbr[:, :, k] + two * g - ntl[:, :, 1] - nbr[:, :, 1]
        d2 = np.abs(ntr[:, :, k] - nbl[:, :, k]) + np.abs(ntr[:, :, 1] - g) + np.abs(nbl[:, :, 1] - g)
        g2 = ntr[:, :, k] + nbl[:, :, k] + two * g - ntr[:, :, 1] - nbl[:, :, 1]
        return np.where(d1 > d2, g2 * half, np.where(d1 < d2, g1 * half, (g1 + g2) * quarter))

    is_green = (color == 1) | (color == 3)
    red = np.where(is_green, np.where(red_right, cross(nl, nr, 0), cross(nt, nb, 0)), np.where(color == 2, star(0), c[:, :, 0]))
    blue = np.where(is_green, np.where(red_right, cross(nt, nb, 2), cross(nl, nr, 2)), np.where(color == 0, star(2), c[:, :, 2]))
    out = np.stack([red, g, blue], -1).astype(f32)
    edge = (cc == 0) | (rr == 0) | (cc == w - 1) | (rr == h - 1)
    return np.maximum(np.where(edge[:, :, None], c, out), f32(0)).astype(f32)


def ppg_2d(bayer, name, median_threshold):
    raw = bayer.astype(f32)
    h, w = raw.shape
    word = WORDS[name]
    color = fc_map(h, w, word)
    ring_rgb, ring = border_interpolate(raw, color, 3)
    temp = np.where(ring[:, :, None], ring_rgb, f32(0)).astype(f32)  # temp_buffer_ starts as zeros (ppg.cu:402)
    src = pre_median(raw, color, f32(median_threshold) / f32(100)) if median_threshold > 0 else raw
    temp = ppg_green(src, color, temp)
    return ppg_redblue(temp, color, word)


# ------------------------------------------------------------------------------------------------ PostProcess
def color_smoothing(rgb):
    h, w, _ = rgb.shape
    p = padded(rgb, 1)
    out = rgb.copy()
    for k in (0, 2):
        d = np.stack([tap(p, 1, dr, dc, h, w)[:, :, k] - tap(p, 1, dr, dc, h, w)[:, :, 1] for dr in (-1, 0, 1) for dc in (-1, 0, 1)], -1)
        out[:, :, k] = np.maximum(np.sort(d.astype(f32), -1)[:, :, 4] + rgb[:, :, 1], f32(0))
    return np.maximum(out, f32(0)).astype(f32)


def green_eq_local(rgb, color, thr):
    h, w, _ = rgb.shape
    g = rgb[:, :, 1]
    p = padded(g, 2)
    t = lambda dr, dc: tap(p, 2, dr, dc, h, w)
    o1 = [t(-1, -1), t(-1, 1), t(1, -1), t(1, 1)]
    o2 = [t(-2, 0), t(2, 0), t(0, -2), t(0, 2)]
    four, six = f32(4), f32(6)
    m1 = (o1[0] + o1[1] + o1[2] + o1[3]) / four
    m2 = (o2[0] + o2[1] + o2[2] + o2[3]) / four

    def spread(o):
        return (np.abs(o[0] - o[1]) + np.abs(o[0] - o[2]) + np.abs(o[0] - o[3]) + np.abs(o[1] - o[2]) + np.abs(o[2] - o[3]) + np.abs(o[1] - o[3])) / six

    rr = np.mgrid[0:h, 0:w][0]
    with np.errstate(all='ignore'):
        ratio = m1 / m2
        ok = (color == 1) & ((rr & 1) == 1) & (m2 > 0) & (m1 > 0) & (ratio < f32(2)) & (g < f32(0.95)) & (spread(o1) < thr) & (spread(o2) < thr)
        out = rgb.copy()
        out[:, :, 1] = np.maximum(np.where(ok, g * ratio, g), f32(0))
    return out.astype(f32)


def green_sums(rgb, color):
    h, w, _ = rgb.shape
    rr, cc = np.mgrid[0:h, 0:w]
    inimg = (cc < 2 * (w // 2)) & (rr < 2 * (h // 2))
    g = rgb[:, :, 1].astype(np.float64)
    return g[inimg & (color == 1) & ((rr & 1) == 0)].sum(), g[inimg & (color == 1) & ((rr & 1) == 1)].sum()


# ------------------------------------------------------------------------------------------------ tests
@pytest.mark.parametrize('pattern', list(WORDS))
@pytest.mark.parametrize('size', [(40, 52), (37, 50), (9, 12), (6, 8)])
@pytest.mark.parametrize('median', [0.0, 1.5, 30.0])
def test_ppg_2d_restatement_matches_the_oracle_bit_for_bit(oracle, scene, pattern, size, median):
    h, w = size
    bayer = oracle.mosaic(scene(h, w, 61), oracle.PATTERNS[pattern])[:, :, 0]
    if median == 30.0:  # a threshold wide enough that most taps count: exercises the (cnt - 1) / 2 pick, not only cnt == 1
        bayer = (bayer + np.random.default_rng(5).normal(0, 0.05, bayer.shape)).astype(np.float32)
    got = ppg_2d(bayer, pattern, median)
    ref = oracle.ppg(bayer[:, :, None], oracle.PATTERNS[pattern], median)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f'{len(bad)} mismatches, first at {bad[:5].tolist()}, max |d| {np.abs(got - ref).max()}'


def test_fc_codes_are_0_1_2_and_green_is_always_1(oracle):
    """The four BayerPattern words never produce code 3: the `c == 1 && (y & 1)` tests of postprocess.cu see every green."""
    for name, word in WORDS.items():
        codes = fc_map(4, 4, word)
        assert set(np.unique(codes)) == {0, 1, 2}
        assert (codes == 1).sum() == 8
        assert np.array_equal(codes, oracle.cfa_color(*np.mgrid[0:4, 0:4], oracle.PATTERNS[name]))


@pytest.mark.parametrize('pattern', list(WORDS))
@pytest.mark.parametrize('size', [(40, 52), (33, 47), (5, 7)])
def test_postprocess_2d_restatement_matches_the_oracle(oracle, scene, pattern, size):
    h, w = size
    rgb = scene(h, w, 62)
    rgb[:, :, 1] *= np.where(np.mgrid[0:h, 0:w][0] & 1, np.float32(1.03), np.float32(1.0))  # a green imbalance to equilibrate
    rgb = rgb.astype(np.float32)
    word = oracle.PATTERNS[pattern]
    color = fc_map(h, w, WORDS[pattern])
    # colour smoothing, 1 and 3 passes: bit for bit
    x = rgb
    for passes in (1, 2, 3):
        x = color_smoothing(x)
        assert np.array_equal(x, oracle.postprocess(rgb, word, color_smoothing_passes=passes)), f'{passes} smoothing passes'
    # local equilibration (threshold / 100, postprocess.cu:365): bit for bit, alone and behind the smoothing
    for thr in (0.04, 4.0):
        assert np.array_equal(green_eq_local(rgb, color, np.float32(thr / 100.0)), oracle.postprocess(rgb, word, green_eq_local=True, green_eq_threshold=thr))
    assert np.array_equal(green_eq_local(x, color, np.float32(0.04 / 100.0)),
                          oracle.postprocess(rgb, word, color_smoothing_passes=3, green_eq_local=True))
    # global equilibration: pixel.y *= sum2 / sum1 at even-row greens.  The reference adds the sums in a workgroup tree and
    # then with torch.sum (order unspecified): compare the ratio against an fp64 sum with the bound of fp32 summation.
    s1, s2 = green_sums(rgb, color)
    ref = oracle.postprocess(rgb, word, green_eq_global=True)
    rr = np.mgrid[0:h, 0:w][0]
    g1 = (color == 1) & ((rr & 1) == 0)
    want = np.maximum(rgb, 0).astype(np.float64)
    want[:, :, 1] = np.where(g1, want[:, :, 1] * (s2 / s1 if s1 > 0 and s2 > 0 else 1.0), want[:, :, 1])
    rel = h * w * 2.0 ** -24 + 2.0 ** -22  # each fp32 sum within n * 2^-24 relative; quotient and product one rounding each
    assert np.all(np.abs(ref - want) <= rel * np.abs(want) + 1e-30)
    # with the oracle's own ratio the apply pass is one multiplication: bit for bit
    ratio = np.float32(1.0173)
    want32 = np.maximum(rgb, np.float32(0)).copy()
    want32[:, :, 1] = np.where(g1, np.maximum(rgb[:, :, 1] * ratio, np.float32(0)), want32[:, :, 1])
    assert np.array_equal(want32, oracle.postprocess(rgb, word, green_eq_global=True, ratio_override=float(ratio)))


# ------------------------------------------------------------------------------------------------ image statistics
def metrics64(images, stride, min_gray, rescale, gray_w):
    """compute_image_bounds / compute_image_metrics (tonemap/color_adaption.cu:11-36, 39-84, 90-166) on the stride grid, in fp64"""
    lo, hi = 0.0, 1.0
    if rescale:
        lo = min(float(im[::stride, ::stride].min()) for im in images)
        hi = max(float(im[::stride, ::stride].max()) for im in images)
    acc, cnt = np.zeros(5), 0.0
    for im in images:
        s = (im[::stride, ::stride].astype(np.float64).reshape(-1, 3) - lo) / (np.float64(np.float32(hi) - np.float32(lo)) + 1e-6)
        ok = ~(s >= 0.99).any(1)
        gray = s @ np.asarray(gray_w, np.float64)
        acc += np.array([np.log(np.maximum(gray, min_gray))[ok].sum(), gray[ok].sum(), *s[ok].sum(0)])
        cnt += ok.sum()
    return acc / max(cnt, 1.0), (lo, hi)


@pytest.mark.parametrize('stride', [1, 3, 8])
@pytest.mark.parametrize('rescale', [False, True])
def test_image_statistics_second_source(oracle, scene, stride, rescale):
    imgs = [scene(70, 93, 63), (scene(41, 50, 64) * np.float32(1.3)).astype(np.float32)]  # the second one holds saturated samples
    got = oracle.image_metrics(imgs, stride, 1e-4, rescale)
    want, (lo, hi) = metrics64(imgs, stride, 1e-4, rescale, (0.299, 0.587, 0.114))  # device_math.h rgb_to_gray weights
    if rescale:
        assert np.array_equal(oracle.image_bounds(imgs, stride), np.array([lo, hi], np.float32))
    assert np.allclose(got, want, rtol=3e-6, atol=1e-7), (got, want)
