"""CPU-only: float64 numpy closed forms of the colour operators and the four tone mappers, against the C oracle.

The oracle (oracle/src/color.h, color_ops.c, tonemap.c) restates the reference's two colour headers and tone-map kernels in
fp32 C.  These tests restate them again as vectorised float64 numpy, straight from the reference's formulas:
  header A  csrc/device_conversions.h        (public colour ops, luminance extract / replace): piecewise sRGB, lab_f with
            powf(t, 1/3) above 0.008856 and 7.787 t + 16/116 below, lab_f_inv on t^3 > 0.008856, power-law HSL adjust
  header B  csrc/device_color_conversions.h  (vibrance inside the tone mappers): cbrt / delta = 6/29 constants, no clamps
  tone maps csrc/tonemap/color_adaption.h:17-59, reinhard.cu:38-43, aces.cu:13-34,57-84, linear.cu:33-38
on a few thousand random pixels (in and out of gamut, both sides of every branch threshold).  Agreement to 2e-6 (fp32 libm vs
float64) means the oracle's constants, branch conditions and operation order are the reference's -- a slip in either
restatement shows up as 1e-3 or more."""

import numpy as np
import pytest

M_RGB2XYZ = np.array([[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]])
M_XYZ2RGB = np.array([[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]])
D65 = np.array([0.95047, 1.0, 1.08883])


def powp(x, y):
    """pow on the branch that is selected; the other branch may see a negative base (its value is discarded)."""
    return np.power(np.maximum(x, 0.0), y)


# ---------------------------------------------------------------- header A (device_conversions.h)
class A:
    @staticmethod
    def srgb_to_linear(c):
        return np.where(c > 0.04045, powp((c + 0.055) / 1.055, 2.4), c / 12.92)

    @staticmethod
    def linear_to_srgb(c):
        return np.where(c > 0.0031308, 1.055 * powp(c, 1 / 2.4) - 0.055, c * 12.92)

    @staticmethod
    def lab_f(t):
        return np.where(t > 0.008856, powp(t, 1 / 3), 7.787 * t + 16 / 116)

    @staticmethod
    def lab_f_inv(t):
        return np.where(t ** 3 > 0.008856, t ** 3, (t - 16 / 116) / 7.787)

    @classmethod
    def rgb_to_xyz(cls, rgb):
        return cls.srgb_to_linear(rgb) @ M_RGB2XYZ.T

    @classmethod
    def xyz_to_lab(cls, xyz):
        f = cls.lab_f(xyz / D65)
        return np.stack([(116 / 100) * f[:, 1] - 16 / 100, (500 / 128) * (f[:, 0] - f[:, 1]), (200 / 128) * (f[:, 1] - f[:, 2])], -1)

    @classmethod
    def lab_to_xyz(cls, lab):
        fy = lab[:, 0] * (100 / 116) + 16 / 116
        f = np.stack([lab[:, 1] * (128 / 500) + fy, fy, fy - lab[:, 2] * (128 / 200)], -1)
        return cls.lab_f_inv(f) * D65

    @classmethod
    def xyz_to_rgb(cls, xyz):
        return cls.linear_to_srgb(xyz @ M_XYZ2RGB.T)

    @classmethod
    def rgb_to_lab(cls, rgb):
        return cls.xyz_to_lab(cls.rgb_to_xyz(rgb))

    @classmethod
    def lab_to_rgb(cls, lab):
        return cls.xyz_to_rgb(cls.lab_to_xyz(lab))

    @classmethod
    def lab_l(cls, rgb):
        y = cls.srgb_to_linear(rgb) @ M_RGB2XYZ[1]
        return np.maximum(0.0, (116 / 100) * cls.lab_f(y) - 16 / 100)

    @classmethod
    def modify_vibrance(cls, rgb, amount):
        lab = cls.rgb_to_lab(rgb)
        chroma = np.hypot(lab[:, 1], lab[:, 2])
        ls, ss = 1 - amount * chroma * 0.25, 1 + amount * chroma
        return np.clip(cls.lab_to_rgb(np.stack([lab[:, 0] * ls, lab[:, 1] * ss, lab[:, 2] * ss], -1)), 0, 1)

    @staticmethod
    def rgb_to_hsl(rgb):
        mx, mn = rgb.max(1), rgb.min(1)
        d = mx - mn
        l = (mx + mn) * 0.5
        ok = d > 1e-6
        dd = np.where(ok, d, 1.0)
        s = np.where(ok, np.where(l < 0.5, d / np.where(ok, mx + mn, 1.0), d / np.where(ok, 2.0 - mx - mn, 1.0)), 0.0)
        r, g, b = rgb.T
        h = np.where(mx == r, (g - b) / dd + np.where(g < b, 6.0, 0.0), np.where(mx == g, (b - r) / dd + 2.0, (r - g) / dd + 4.0)) / 6.0
        return np.stack([np.where(ok, h, 0.0), s, l], -1)

    @staticmethod
    def hue(p, q, t):
        t = np.where(t < 0, t + 1, t)
        t = np.where(t > 1, t - 1, t)
        return np.where(t < 1 / 6, p + (q - p) * 6 * t, np.where(t < 1 / 2, q, np.where(t < 2 / 3, p + (q - p) * (2 / 3 - t) * 6, p)))

    @classmethod
    def hsl_to_rgb(cls, hsl):
        h, s, l = hsl.T
        q = np.where(l < 0.5, l * (1 + s), l + s - l * s)
        p = 2 * l - q
        rgb = np.stack([cls.hue(p, q, h + 1 / 3), cls.hue(p, q, h), cls.hue(p, q, h - 1 / 3)], -1)
        return np.where((s < 1e-6)[:, None], l[:, None], rgb)

    @classmethod
    def modify_hsl(cls, rgb, dh, ds, dl_):
        hsl = cls.rgb_to_hsl(rgb)
        h = hsl[:, 0] + dh
        h = np.where(h < 0, h + 1, h)
        h = np.where(h > 1, h - 1, h)
        return np.clip(cls.hsl_to_rgb(np.stack([h, powp(hsl[:, 1], 1 / (1 + ds)), powp(hsl[:, 2], 1 / (1 + dl_))], -1)), 0, 1)


# ---------------------------------------------------------------- header B (device_color_conversions.h)
class B:
    DELTA = 6 / 29

    @staticmethod
    def srgb_to_linear(c):
        return np.where(c <= 0.04045, c / 12.92, powp((c + 0.055) / 1.055, 2.4))

    @staticmethod
    def linear_to_srgb(c):
        return np.where(c <= 0.0031308, 12.92 * c, 1.055 * powp(c, 1 / 2.4) - 0.055)

    @classmethod
    def rgb_to_lab(cls, rgb):
        n = (cls.srgb_to_linear(rgb) @ M_RGB2XYZ.T) / D65
        f = np.where(n > cls.DELTA ** 3, np.cbrt(n), n / (3 * cls.DELTA ** 2) + 4 / 29)
        return np.stack([(116 * f[:, 1] - 16) / 100, 500 * (f[:, 0] - f[:, 1]) / 128, 200 * (f[:, 1] - f[:, 2]) / 128], -1)

    @classmethod
    def lab_to_rgb(cls, lab):
        fy = (lab[:, 0] * 100 + 16) / 116
        f = np.stack([lab[:, 1] * 128 / 500 + fy, fy, fy - lab[:, 2] * 128 / 200], -1)
        xyz = np.where(f > cls.DELTA, f ** 3, 3 * cls.DELTA ** 2 * (f - 4 / 29)) * D65
        return cls.linear_to_srgb(xyz @ M_XYZ2RGB.T)

    @classmethod
    def vibrance(cls, rgb, amount):
        lab = cls.rgb_to_lab(rgb)
        chroma = np.hypot(lab[:, 1], lab[:, 2])
        ls, ss = 1 - amount * chroma * 0.25, 1 + amount * chroma
        return np.clip(cls.lab_to_rgb(np.stack([lab[:, 0] * ls, lab[:, 1] * ss, lab[:, 2] * ss], -1)), 0, 1)


def pixels(n, lo=0.0, hi=1.0, seed=0):
    rng = np.random.default_rng(seed)
    px = rng.uniform(lo, hi, (n, 3))
    px[: n // 8] *= 0.05                       # dark: the linear branches of sRGB / lab_f
    px[n // 8: n // 4, :] = px[n // 8: n // 4, :1]  # greys: HSL's delta <= 1e-6 branch
    return px.astype(np.float32)


TOL = 2e-6


@pytest.mark.parametrize('op,fn,lo,hi', [
    ('rgb_to_xyz', A.rgb_to_xyz, -0.1, 1.2), ('xyz_to_lab', A.xyz_to_lab, 0.0, 1.1), ('lab_to_xyz', A.lab_to_xyz, None, None),
    ('xyz_to_rgb', A.xyz_to_rgb, 0.0, 1.0), ('rgb_to_lab', A.rgb_to_lab, 0.0, 1.0), ('lab_to_rgb', A.lab_to_rgb, None, None)])
def test_colour_space_ops_closed_form(oracle, op, fn, lo, hi):
    if lo is None:  # Lab input: L in [0, 1], a / b in [-0.6, 0.6]
        px = pixels(4096, 0.0, 1.0, 3)
        px[:, 1:] = (px[:, 1:] - 0.5) * 1.2
    else:
        px = pixels(4096, lo, hi, 3)
    got = oracle.color_op(op, px[None])[0]
    ref = fn(px.astype(np.float64))
    assert np.abs(got - ref).max() <= TOL * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


@pytest.mark.parametrize('adj', [(0.1, 0.3, -0.2), (-0.25, -0.4, 0.5), (0.0, 0.0, 0.0)])
def test_modify_hsl_closed_form(oracle, adj):
    px = pixels(4096, 0.0, 1.0, 5)
    got = oracle.color_op('modify_hsl', px[None], adj)[0]
    ref = A.modify_hsl(px.astype(np.float64), *adj)
    d = np.abs(got - ref)
    # the hue wheel is piecewise linear with slope <= 6 (q - p): fp32 hue error is amplified by that much
    assert d.max() <= 2e-5 and np.quantile(d, 0.999) <= 4e-6, (d.max(), np.quantile(d, 0.999))


@pytest.mark.parametrize('amount', [0.0, 0.5, -0.3])
def test_modify_vibrance_closed_form(oracle, amount):
    px = pixels(4096, 0.0, 1.0, 7)
    got = oracle.color_op('modify_vibrance', px[None], [amount])[0]
    assert np.abs(got - A.modify_vibrance(px.astype(np.float64), amount)).max() <= 3 * TOL


def test_luminance_closed_form(oracle):
    px = pixels(4096, -0.05, 1.1, 9)
    p64 = np.clip(px.astype(np.float64), 0, 1)   # compute_luminance clips its input (color_conversions.cu:24)
    assert np.abs(oracle.compute_luminance(px[None])[0] - A.lab_l(p64)).max() <= TOL
    ll = oracle.compute_luminance(px[None], True, 1e-4)[0]
    L = np.maximum(1e-4, A.lab_l(p64))
    assert (np.abs(ll - np.log(L)) <= 1e-6 + TOL / L).all()   # d log L = dL / L: the log amplifies dark values' error


# ---------------------------------------------------------------- tone mappers
def adaptation(rgb, metrics, light_adapt, intensity):
    norm = np.clip(-metrics[0] / 9.21034, 0.0, 1.0)
    map_key = 0.3 + 0.7 * norm ** 1.4
    mean = metrics[2:5][None] + light_adapt * (rgb - metrics[2:5][None])   # lerp(t, global, pixel) = a + t (b - a)
    return powp(mean / np.exp(intensity), map_key)


def aces(rgb):
    m_in = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
    m_out = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
    v = rgb @ m_in.T
    return ((v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081)) @ m_out.T


def tonemap64(name, rgb, metrics, gamma, intensity, light_adapt, vibrance):
    if name == 'reinhard':
        a = adaptation(rgb, metrics, light_adapt, intensity)
        t = rgb / (a + rgb)
    elif name == 'linear':
        t = rgb / adaptation(rgb, metrics, light_adapt, intensity)
    elif name == 'aces':
        t = aces(rgb * 2.0 ** intensity)
    else:
        t = aces(rgb / adaptation(rgb, metrics, light_adapt, intensity))
    g = powp(t, 1 / gamma)
    return B.vibrance(g, vibrance), np.maximum(1.0, g.max(1, keepdims=True))   # modify_rgb_vibrance_dt clips, also for amount = 0


@pytest.mark.parametrize('name', ['reinhard', 'aces', 'adaptive_aces', 'linear'])
@pytest.mark.parametrize('prm', [(0.75, 2.0, 1.0, 0.0), (2.2, 0.5, 0.6, 0.4), (1.0, -1.0, 0.0, -0.3)])
def test_tonemaps_closed_form(oracle, name, prm):
    gamma, intensity, light_adapt, vibrance = prm
    px = pixels(4096, 0.005, 1.0, 11)
    metrics = np.array([-2.3, 0.18, 0.21, 0.19, 0.15], np.float32)
    u8, got = oracle.tonemap(name, px[None], metrics, gamma, intensity, light_adapt, vibrance, return_float=True)
    ref, scale = tonemap64(name, px.astype(np.float64), metrics.astype(np.float64), gamma, intensity, light_adapt, vibrance)
    d = np.abs(got[0] - ref)
    # fp32 error is relative to the largest intermediate: an over-range pixel (the linear mapper reaches ~50 before the final
    # clip) goes through the Lab round trip at that magnitude
    # ... and a dark channel of such a pixel comes back through x^(1/2.4), whose slope near 0.003 is ~12
    r = d / scale
    assert np.quantile(r, 0.99) <= 2e-6 and r.max() <= 5e-5, (np.quantile(r, 0.99), r.max())
    # quantisation: round(clip(v) * 255); a pixel may sit on a rounding tie
    q = np.floor(np.clip(ref, 0, 1) * 255 + 0.5)
    dq = np.abs(u8[0].astype(np.float64) - q)
    assert dq.max() <= 1 and (dq > 0).mean() < 2e-3
