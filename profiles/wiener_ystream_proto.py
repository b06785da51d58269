"""numpy prototype of the y-streaming Wiener tile kernel's data flow (K = 32, ov = 4), fp32.

Checks the algebra the HIP kernel (csrc/wiener_ystream.hip) relies on, against the CPU oracle:
  * the row FFT of an 8-row block is computed ONCE and shared by the 4 tile rows that contain it
    (the per-tile mean and the row factor of the analysis window are applied in the kx domain);
  * the tile rows are held in circular row order m = padded_row mod 32: a circular shift only multiplies the
    spectrum by a unit phase, the Wiener gain depends on |X| alone, so the shift needs no un-doing;
  * the overlap-add ACROSS tile rows happens in the (row, kx) domain, before the inverse row FFT, which then runs once
    per finished 8-row block instead of once per tile row;
  * two adjacent tiles ride one complex transform (Hermitian split).

  python profiles/wiener_ystream_proto.py [H W]
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'oracle'))

K, OV, S = 32, 4, 8
f32, c64 = np.float32, np.complex64


def reflect(x, limit):
  x = np.where(x < 0, -x, x)
  return np.where(x >= limit, 2 * limit - x - 1, x)


def window():
  half = K / 2.0
  r = -half + 0.5 + np.arange(K)
  v = np.exp(-(r * r) / (0.3 * half * half))
  return (v / np.sqrt((v * v).sum())).astype(f32)


def wiener_ystream(img: np.ndarray, sigma: float, use_ref: bool = True) -> np.ndarray:
  H, W = img.shape
  img = img.astype(f32)
  wx = window()
  What = np.fft.fft(wx.astype(c64)).astype(c64)  # per-kx constant
  ntx, nty = (W - 1) // S + OV, (H - 1) // S + OV
  npairs = (ntx + 1) // 2
  PH_, PW_ = 8 * (nty + 3), 8 * (ntx + 3)
  acc = np.zeros((PH_, PW_), f32)  # padded coordinates u = pixel + 24
  sig2 = f32(sigma * sigma)
  kneg = (-np.arange(K)) % K
  for p in range(npairs):
    ox = (2 * p - (OV - 1)) * S
    cols = reflect(ox + np.arange(K + S), W)
    act_b = 2 * p + 1 < ntx
    win = np.zeros((K, K), c64)
    carry = np.zeros((K, K), c64)
    Sa, Sb, ref = np.zeros(4, f32), np.zeros(4, f32), np.zeros(4, f32)

    def emit(slot, t_block):
      rows_k = carry[8 * slot:8 * slot + 8].copy()
      carry[8 * slot:8 * slot + 8] = 0
      z = np.fft.ifft(rows_k, axis=1).astype(c64)
      s_ = np.zeros((8, K + S), f32)
      s_[:, :K] += wx * z.real
      s_[:, S:] += wx * z.imag
      acc[8 * t_block:8 * t_block + 8, 16 * p:16 * p + K + S] += s_

    for q in range(nty + 3):  # 8-row block index in padded rows
      rows = reflect(8 * q + np.arange(8) - 24, H)
      blk = img[rows][:, cols]
      slot = q & 3
      Sa[slot] = blk[:, :K].sum(dtype=f32)
      Sb[slot] = blk[:, S:].sum(dtype=f32)
      ref[slot] = Sa[slot] / f32(256) if use_ref else f32(0)
      re = (blk[:, :K] - ref[slot]) * wx
      im = (blk[:, S:] - ref[slot]) * wx if act_b else np.zeros_like(re)
      win[8 * slot:8 * slot + 8] = np.fft.fft((re + 1j * im).astype(c64), axis=1).astype(c64)
      t = q - 3
      if 0 <= t < nty:
        ph = t & 3
        mean_a = Sa.sum(dtype=f32) / f32(1024)
        mean_b = Sb.sum(dtype=f32) / f32(1024) if act_b else f32(0)
        m = np.arange(K)
        y = (m - 8 * ph) & 31
        da = ref[m >> 3] - mean_a
        db = (ref[m >> 3] - mean_b) if act_b else np.zeros(K, f32)
        D = (da[:, None] + 1j * db[:, None]).astype(c64) * What[None, :]
        Z = (wx[y][:, None] * (win + D)).astype(c64)
        X = np.fft.fft(Z, axis=0).astype(c64)
        Xc = np.conj(X[kneg][:, kneg])
        A, B = (X + Xc) / 2, (X - Xc) / 2j
        pa = (A.real ** 2 + A.imag ** 2 + f32(1e-15)).astype(f32)
        pb = (B.real ** 2 + B.imag ** 2 + f32(1e-15)).astype(f32)
        ga = np.maximum(pa - sig2, 0) / pa
        gb = np.maximum(pb - sig2, 0) / pb
        v = np.fft.ifft((ga * A + 1j * gb * B).astype(c64), axis=0).astype(c64)
        Mhat = ((mean_a + 1j * mean_b) * What).astype(c64)
        carry += (wx[y][:, None] * (v + wx[y][:, None] * Mhat[None, :])).astype(c64)
        emit(ph, t)
    for f in range(3):  # flush the three unfinished blocks
      emit((nty + f) & 3, nty + f)
  m1 = np.array([sum(wx[r + k * S] * wx[r + k * S] for k in range(OV)) for r in range(S)], f32)
  yy, xx = np.arange(H), np.arange(W)
  mask = m1[yy & 7][:, None] * m1[xx & 7][None, :]
  return acc[24:24 + H, 24:24 + W] / (mask + f32(1e-15))


if __name__ == '__main__':
  import tdk_oracle as O

  H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (104, 136)
  rng = np.random.default_rng(3)
  yy, xx = np.mgrid[0:H, 0:W]
  base = 0.4 + 0.3 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + 0.2 * (xx > W // 2)
  for name, img, sigma in [
    ('unit range', (base + 0.02 * rng.standard_normal((H, W))).clip(0, 1).astype(f32), 0.05),
    ('log-luminance', np.log(np.maximum(1e-4, (base * 0.1 + 0.002 * rng.standard_normal((H, W))).clip(0, 1))).astype(f32), 0.075),
    ('dark log (-9)', np.log(np.maximum(1e-4, 2e-4 + 1e-5 * rng.standard_normal((H, W)))).astype(f32), 0.075),
  ]:
    ref = O.wiener(img[:, :, None], [sigma], 32, 4)[:, :, 0]
    for use_ref in (True, False):
      got = wiener_ystream(img, sigma, use_ref)
      print(f'{name:16s} ref-level={use_ref!s:5s} max|d| = {np.abs(got - ref).max():.3e}   (|img| max {np.abs(img).max():.2f})')
