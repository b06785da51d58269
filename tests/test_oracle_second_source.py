"""CPU-only: second-source checks of the CPU oracle.

The reference pins no kernel op with vectors, and the HIP kernels and
oracle/src/*.c share one author and one reading of the reference.  These tests restate three
ops a SECOND time, independently of the C oracle's code structure -- float64 numpy with library
FFTs / convolutions instead of hand-written butterflies and line walkers -- directly from the
formulas in SURVEY.md Appendix A, and require the oracle to agree.  A misreading shared by both
restatements is still possible; an implementation slip in either is not.
"""

import numpy as np
import pytest
from scipy import ndimage


# ------------------------------------------------------------------ Wiener (denoise.cu:84-242)
def wiener_fp64(img, sigma, K, ov):
    """float64 / numpy.fft restatement: tile origins (g - ov) * s, asymmetric reflect, L2-normalised Gaussian
    windows (weight 0.3), per-tile mean, gain max(|X|^2 + 1e-15 - sigma^2, 0) / (|X|^2 + 1e-15), overlap-add of
    (y + mean * w2d) * w2d into a (H + 2K, W + 2K) buffer at origin + K (high-side bounds check only), / (mask + 1e-15)."""
    H, W = img.shape
    s = K // ov
    r = -K / 2 + 0.5 + np.arange(K)
    w = np.exp(-(r * r) / (0.3 * (K / 2) ** 2))
    w /= np.sqrt((w * w).sum())
    w2d = np.outer(w, w)

    def reflect(x, L):
        x = np.where(x < 0, -x, x)
        return np.where(x >= L, 2 * L - x - 1, x)

    Hp, Wp = H + 2 * K, W + 2 * K
    acc, mask = np.zeros((Hp, Wp)), np.zeros((Hp, Wp))
    grid_h, grid_w = -(-(H + K) // s) + ov, -(-(W + K) // s) + ov
    x64 = img.astype(np.float64)
    for gy in range(grid_h):
        oy = (gy - ov) * s
        ys = reflect(oy + np.arange(K), H)
        if (ys < 0).any() or (ys >= H).any():
            continue  # far outside the frame: only cropped-away samples (reflect is single-bounce)
        for gx in range(grid_w):
            ox = (gx - ov) * s
            xs = reflect(ox + np.arange(K), W)
            if (xs < 0).any() or (xs >= W).any():
                continue
            tile = x64[np.ix_(ys, xs)]
            mean = tile.mean()
            X = np.fft.fft2((tile - mean) * w2d)
            P = np.abs(X) ** 2 + 1e-15
            y = np.real(np.fft.ifft2(np.maximum(P - sigma * sigma, 0.0) / P * X))
            py, px = oy + K, ox + K
            if py < 0 or px < 0:
                continue
            hy, hx = min(K, Hp - py), min(K, Wp - px)
            if hy <= 0 or hx <= 0:
                continue
            acc[py:py + hy, px:px + hx] += ((y + mean * w2d) * w2d)[:hy, :hx]
            mask[py:py + hy, px:px + hx] += (w2d * w2d)[:hy, :hx]
    return acc[K:K + H, K:K + W] / (mask[K:K + H, K:K + W] + 1e-15)


@pytest.mark.parametrize('K,ov,shape', [(16, 4, (40, 56)), (32, 4, (72, 88)), (16, 2, (48, 48)), (32, 8, (64, 72)), (32, 2, (80, 64))])
@pytest.mark.parametrize('sigma', [0.05, 0.5])
def test_wiener_oracle_vs_fp64_numpy_fft(oracle, scene, K, ov, shape, sigma):
    h, w = shape
    img = scene(h, w, 11 + K + ov)[:, :, 1].copy()
    got = oracle.wiener(img[:, :, None], sigma, K, ov)[:, :, 0]
    ref = wiener_fp64(img, sigma, K, ov)
    assert np.abs(got - ref).max() < 1e-5


def test_wiener_oracle_vs_fp64_on_log_luminance_range(oracle, scene):
    """The pipeline feeds log-luminance (values in [-9.2, 0]); same check on that range, tolerance scaled by the range."""
    lum = oracle.compute_luminance(scene(72, 88, 5), log=True, eps=1e-4)
    got = oracle.wiener(lum[:, :, None], 0.075, 32, 4)[:, :, 0]
    assert np.abs(got - wiener_fp64(lum, 0.075, 32, 4)).max() < 2e-5


def test_wiener_fp64_restatement_is_self_consistent(scene):
    img = scene(40, 56, 3)[:, :, 0].copy()
    assert np.abs(wiener_fp64(img, 0.0, 16, 4) - img).max() < 1e-7          # sigma = 0: identity (3e-8)
    assert np.abs(wiener_fp64(np.full((40, 56), 0.3), 0.7, 16, 4) - 0.3).max() < 1e-12  # constant image: fixed point


# ------------------------------------------------------------------ bilateral grid (bilateral.cu:71-228,273-299)
def bilateral_fp64(L, sigma_s, sigma_r, detail):
    """float64 dense restatement: trilinear splat (np.add.at), [1 4 6 4 1]/16 blurs along x and y with zero
    extension (scipy correlate1d, mode='constant'), [-2 -4 0 4 2]/16 derivative along z with zero extension,
    trilinear slice, out = max(0, L - detail * sigma_r * 4 * value)."""
    H, W = L.shape
    f = np.float32  # the grid SIZE is defined by float32 arithmetic (roundf = half away from zero, ceilf of fp32 quotients)
    ss = max(f(sigma_s), f(0.5))
    gx = np.clip(np.floor(f(W) / ss + f(0.5)), 4, 3000).astype(f)
    gy = np.clip(np.floor(f(H) / ss + f(0.5)), 4, 3000).astype(f)
    gz = np.clip(np.floor(f(1.0) / f(sigma_r) + f(0.5)), 4, 50).astype(f)
    s_s = max(f(H) / gy, f(W) / gx)
    s_r = f(1.0) / gz
    sx, sy, sz = int(np.ceil(f(W) / s_s)) + 1, int(np.ceil(f(H) / s_s)) + 1, int(np.ceil(f(1.0) / s_r)) + 1

    yy, xx = np.mgrid[0:H, 0:W]
    cx = np.clip(xx / sigma_s, 0, sx - 1)
    cy = np.clip(yy / sigma_s, 0, sy - 1)
    cz = np.clip(L.astype(np.float64) / sigma_r, 0, sz - 1)
    ix, iy, iz = np.minimum(cx.astype(int), sx - 2), np.minimum(cy.astype(int), sy - 2), np.minimum(cz.astype(int), sz - 2)
    fx, fy, fz = cx - ix, cy - iy, cz - iz

    grid = np.zeros((sz, sy, sx))
    for dz, wz in ((0, 1 - fz), (1, fz)):
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                np.add.at(grid, (iz + dz, iy + dy, ix + dx), wx * wy * wz / (sigma_s * sigma_s))
    k5 = np.array([1, 4, 6, 4, 1]) / 16.0
    grid = ndimage.correlate1d(grid, k5, axis=2, mode='constant')
    grid = ndimage.correlate1d(grid, k5, axis=1, mode='constant')
    grid = ndimage.correlate1d(grid, np.array([-2, -4, 0, 4, 2]) / 16.0, axis=0, mode='constant')

    val = np.zeros((H, W))
    for dz, wz in ((0, 1 - fz), (1, fz)):
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                val += grid[iz + dz, iy + dy, ix + dx] * wx * wy * wz
    return np.maximum(0.0, L - detail * sigma_r * 4.0 * val), (sx, sy, sz)


@pytest.mark.parametrize('sigma_s,sigma_r', [(2.0, 0.2), (8.0, 0.1), (3.0, 0.15)])
@pytest.mark.parametrize('shape', [(60, 84), (33, 47)])
def test_bilateral_oracle_vs_fp64_dense(oracle, scene, sigma_s, sigma_r, shape):
    h, w = shape
    lum = oracle.compute_luminance(scene(h, w, 21))
    ref, size = bilateral_fp64(lum, sigma_s, sigma_r, 0.4)
    assert oracle.bilateral_grid_size(w, h, sigma_s, sigma_r) == size
    got = oracle.bilateral(lum, sigma_s, sigma_r, 0.4)
    assert np.abs(got - ref).max() < 2e-6


# ------------------------------------------------------------------ white balance estimate (white_balance.cu:57-161)
def test_wb_collect_hand_computed(oracle):
    """4 x 4 cells of 2 x 2 (stride 2): cell (i, j) skipped when i + 1 >= sh or j + 1 >= sw (white_balance.cu:69)."""
    b = np.arange(64, dtype=np.float32).reshape(8, 8) / 100.0
    chroma, inten, mask = oracle.wb_collect_samples(b, oracle.RGGB, stride=2)
    assert chroma.shape == (16, 2) and mask.reshape(4, 4)[:3, :3].all() and not mask.reshape(4, 4)[3].any() and not mask.reshape(4, 4)[:, 3].any()
    # cell (1, 2): quad at (2, 4): p00 = .20, p01 = .21, p10 = .28, p11 = .29 -> r = .20, g = .245, b = .29
    f = np.float32
    s = f(0.20) + (f(0.21) + f(0.28)) * f(0.5) + f(0.29)
    n = 1 * 4 + 2
    assert inten[n] == s and chroma[n, 0] == f(0.20) / s and chroma[n, 1] == ((f(0.21) + f(0.28)) * f(0.5)) / s
    # GBRG: r = p10, g = (p00 + p11) / 2, b = p01 (bayer_device.h:40)
    chroma2, inten2, _ = oracle.wb_collect_samples(b, oracle.GBRG, stride=2)
    s2 = f(0.28) + (f(0.20) + f(0.29)) * f(0.5) + f(0.21)
    assert inten2[n] == s2 and chroma2[n, 0] == f(0.28) / s2
    # a saturated sample in the quad invalidates the cell
    b2 = b.copy()
    b2[3, 5] = 1.0
    assert not oracle.wb_collect_samples(b2, oracle.RGGB, stride=2)[2][n]
    # literal reference positions (pos * 2) vs the intended pos * stride differ for stride 4 ...
    lit = oracle.wb_collect_samples(b, oracle.RGGB, stride=4, literal_positions=True)
    fix = oracle.wb_collect_samples(b, oracle.RGGB, stride=4, literal_positions=False)
    assert lit[1][0] == fix[1][0] and fix[1][0] == inten[0]        # cell (0,0) is the same quad either way
    b3 = np.random.default_rng(0).uniform(0, 0.9, (16, 16)).astype(np.float32)
    lit, fix = (oracle.wb_collect_samples(b3, oracle.RGGB, 4, lp) for lp in (True, False))
    assert lit[1][1 * 4 + 1] == oracle.wb_collect_samples(b3, oracle.RGGB, 2)[1][1 * 8 + 1]   # literal: quad at (2, 2)
    assert fix[1][1 * 4 + 1] == oracle.wb_collect_samples(b3, oracle.RGGB, 2)[1][2 * 8 + 2]   # intended: quad at (4, 4)


def test_wb_estimate_recovers_known_gains(oracle, scene):
    """The estimate is the chroma RATIO of the bright samples, (mean r / mean g, 1, mean b / mean g)
    (white_balance.cu:155-160): a grey scene scaled per channel by (1/1.8, 1, 1/1.4) must return those factors."""
    rng = np.random.default_rng(5)
    grey = rng.uniform(0.2, 0.8, (96, 128)).astype(np.float32)
    grey = ndimage.uniform_filter(grey, 9)  # smooth, so a 2x2 quad sees one grey level
    gains = np.array([1 / 1.8, 1.0, 1 / 1.4], np.float32)
    rgb = np.stack([grey * gains[0], grey, grey * gains[2]], -1).astype(np.float32)
    bayer = oracle.mosaic(rgb, oracle.RGGB)[:, :, 0]
    est = oracle.estimate_white_balance([bayer], oracle.RGGB, quantile=0.9, stride=4)
    assert est[1] == 1.0 and np.abs(est - gains).max() < 0.03
    assert np.array_equal(oracle.estimate_white_balance([np.ones((32, 32), np.float32)], oracle.RGGB), np.ones(3, np.float32))  # all saturated
