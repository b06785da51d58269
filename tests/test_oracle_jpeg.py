"""The CPU restatement of JPEG encoding (oracle/src/jpeg.c: ITU-T T.81 + JFIF, the algorithm behind the reference's nvjpeg calls,
csrc/jpeg_encoder.cu:118-173) pinned by an independent implementation: libjpeg, through Pillow.

nvjpeg's own bytes are not available (closed library, the reference holds no JPEG output): parity with nvjpeg is unpinned.  What
is pinned: the stream is valid (libjpeg decodes every flavour), its quantisation tables are libjpeg's for the same quality (the IJG
scaling nvjpeg documents), the decoded image is as close to the source as libjpeg's own encoding, the optimised Huffman stream is
not larger than libjpeg's optimised baseline stream, and the DCT + quantiser agrees with a float64 DCT-II to within one
quantisation step at rounding ties."""

import io

import numpy as np
import pytest
from PIL import Image

SUBS = {0: '4:4:4', 1: '4:2:2', 2: 'gray'}


def sample_image(h, w, seed=0, noise=6.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0), 127 + 90 * np.sin((xx + yy) / 31.0), 127 + 80 * np.cos(xx / 11.0 - yy / 7.0)], -1)
    return np.clip(img + rng.normal(0, noise, img.shape), 0, 255).astype(np.uint8)


def psnr(a, b):
    return 10 * np.log10(255.0**2 / max(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2), 1e-12))


def pil_encode(img, quality, sub, progressive):
    buf = io.BytesIO()
    src = Image.fromarray(img)
    if sub == 2:
        src.convert('L').save(buf, 'JPEG', quality=quality, optimize=True, progressive=progressive)
    else:
        src.save(buf, 'JPEG', quality=quality, optimize=True, progressive=progressive, subsampling=SUBS[sub])
    return buf.getvalue()


def decode(stream):
    im = Image.open(io.BytesIO(bytes(stream)))
    im.load()
    return im


@pytest.mark.parametrize('sub', [0, 1, 2])
@pytest.mark.parametrize('progressive', [False, True])
@pytest.mark.parametrize('quality', [25, 75, 94, 100])
def test_stream_decodes_and_matches_libjpeg(oracle, sub, progressive, quality):
    img = sample_image(203, 331)  # neither dimension a multiple of the MCU
    s = oracle.jpeg_encode(img, quality, 3, sub, progressive)
    assert bytes(s[:4]) == b'\xff\xd8\xff\xe0' and bytes(s[-2:]) == b'\xff\xd9'
    im = decode(s)
    assert im.size == (331, 203) and im.mode == ('L' if sub == 2 else 'RGB')
    assert im.info.get('progressive', 0) == (1 if progressive else 0)
    ref_stream = pil_encode(img, quality, sub, progressive)
    ref = decode(ref_stream)
    assert {k: list(v) for k, v in im.quantization.items()} == {k: list(v) for k, v in ref.quantization.items()}   # IJG quality scaling
    target = np.asarray(Image.fromarray(img).convert('L')) if sub == 2 else img
    ours, theirs = psnr(np.asarray(im), target), psnr(np.asarray(ref), target)
    assert ours > theirs - 0.1, (ours, theirs)
    assert psnr(np.asarray(im), np.asarray(ref)) > 40.0      # the two decodes agree far better than either agrees with the source
    if not progressive:
        assert len(s) <= len(pil_encode(img, quality, sub, False)) * 1.01   # optimal tables: not larger than libjpeg's optimised stream


def test_input_formats_agree(oracle):
    img = sample_image(64, 96, 3)
    base = oracle.jpeg_encode(img, 90, 3, 1)
    assert np.array_equal(oracle.jpeg_encode(img[:, :, ::-1], 90, 2, 1), base)                       # BGRI
    assert np.array_equal(oracle.jpeg_encode(img.transpose(2, 0, 1), 90, 1, 1), base)                # RGB planar
    assert np.array_equal(oracle.jpeg_encode(img[:, :, ::-1].transpose(2, 0, 1), 90, 0, 1), base)    # BGR planar


def test_coefficients_against_float64_dct(oracle):
    """AAN flow graph + folded scale factors == the DCT-II of T.81 A.3.3 (scipy, float64) followed by division by the table."""
    from scipy.fft import dctn

    img = sample_image(64, 64, 5, noise=20.0)
    quality = 90
    s, coefs = oracle.jpeg_encode(img, quality, 3, 2, return_coefs=True)   # gray: one plane of 8 x 8 blocks
    q = np.array(decode(s).quantization[0]).reshape(-1)   # Pillow >= 8.3 returns the tables in natural order
    R, G, B = (img[..., k].astype(np.float32) for k in range(3))
    y = np.minimum(255, np.rint((np.float32(0.299) * R + np.float32(0.587) * G) + np.float32(0.114) * B)).astype(np.float64) - 128.0
    zz = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49,
                   56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
    coefs = coefs.reshape(8, 8, 64)
    worst = 0.0
    for by in range(8):
        for bx in range(8):
            exact = dctn(y[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8], norm='ortho').reshape(-1)[zz] / q[zz]
            d = np.abs(coefs[by, bx] - exact)
            worst = max(worst, d.max())
    assert worst <= 0.5 + 1e-3   # the nearest integer, up to float32 rounding right at a tie


def test_optimal_tables_are_valid_prefix_codes(oracle):
    """Every DHT segment of the stream: lengths <= 16, Kraft sum < 1 (the all-ones code stays free), no symbol twice."""
    img = sample_image(120, 160, 7, noise=25.0)
    for progressive in (False, True):
        s = bytes(oracle.jpeg_encode(img, 97, 3, 1, progressive))
        i, seen = 2, 0
        while i < len(s):
            assert s[i] == 0xFF
            marker, ln = s[i + 1], int.from_bytes(s[i + 2:i + 4], 'big')
            if marker == 0xC4:
                counts = list(s[i + 5:i + 21])
                vals = list(s[i + 21:i + 2 + ln])
                assert sum(counts) == len(vals) == len(set(vals))
                kraft = sum(c / 2.0 ** (l + 1) for l, c in enumerate(counts))
                assert kraft < 1.0
                seen += 1
            if marker == 0xDA:   # skip the entropy-coded segment
                j = i + 2 + ln
                while not (s[j] == 0xFF and s[j + 1] not in (0x00,) and not 0xD0 <= s[j + 1] <= 0xD7):
                    j += 1
                i = j
                continue
            i += 2 + ln
            if marker == 0xD9:
                break
        assert seen == (4 if not progressive else 2 + 3)


def test_extreme_images(oracle):
    """Constant, saturated and maximum-entropy images at the quality extremes: still valid streams."""
    rng = np.random.default_rng(11)
    for img in (np.full((1, 1, 3), 77, np.uint8), rng.integers(0, 256, (3, 2, 3), dtype=np.uint8), np.zeros((17, 9, 3), np.uint8), np.full((8, 8, 3), 255, np.uint8), rng.integers(0, 256, (40, 56, 3), dtype=np.uint8),
                (rng.integers(0, 2, (33, 47, 3)) * 255).astype(np.uint8)):
        for quality in (1, 100):
            for sub in (0, 1, 2):
                for progressive in (False, True):
                    im = decode(oracle.jpeg_encode(img, quality, 3, sub, progressive))
                    assert im.size == (img.shape[1], img.shape[0])
                    if quality == 100:
                        target = np.asarray(Image.fromarray(img).convert('L')) if sub == 2 else img
                        if sub != 1:  # 4:2:2 loses chroma detail of the noise images by construction
                            assert psnr(np.asarray(im), target) > 30.0
