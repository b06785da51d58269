"""CPU-only: an independent restatement of RCD on 2-D arrays, against the C oracle, bit for bit.

oracle/src/rcd.c restates reference csrc/debayer/rcd.cu literally, flat `idx / 2` slot arithmetic included.  This file
restates the same algorithm a SECOND time with none of that machinery: every plane is a 2-D (row, col) array, every tap
is a (d_row, d_col) shift, and the half-density planes are indexed by the SITE that owns a slot.  The reference's slot
aliasing is written down as the coordinate map SURVEY.md Appendix A.2 derives:

  * a half-density value computed at site (r, c) lives in slot (r, c // 2);
  * step 4.1 runs at every ODD column of every row, so slot (r, j) of the p / q planes holds the value of site (r, 2 j + 1);
  * step 4.2 at site (r, c) therefore sums, for even c, P <- sites (r-1, c-1), (r, c+1), (r+1, c+1) and
    Q <- (r-1, c+1), (r, c+1), (r+1, c-1); for odd c, P <- (r-1, c), (r, c), (r+1, c+2) and Q <- (r-1, c+2), (r, c), (r+1, c)
    (rcd.cu:166-182) -- not a symmetric diagonal;
  * the p / q planes share their memory with the full-density v / h planes of step 1.1 (rcd.cu:637-652): a slot step 4.1
    does not write still holds the v_diff / h_diff value whose flat position it aliases, i.e. that of site
    (r // 2, (r % 2) * W / 2 + j) -- values from the same call, read by step 4.2 near the frame border.

numpy float32 arithmetic is IEEE and uncontracted, like the oracle's (gcc -ffp-contract=off), and every expression keeps the
reference's association, so the interior (margin 7, rcd.cu:48-60) must agree BIT FOR BIT.  A misreading shared with the
oracle would have to survive two different index algebras.
"""

import numpy as np
import pytest

f32 = np.float32
TABLES = {'RGGB': [[0, 1], [1, 2]], 'BGGR': [[2, 1], [1, 0]], 'GRBG': [[1, 0], [2, 1]], 'GBRG': [[1, 2], [0, 1]]}


def sh(a, dr, dc):
    """out[r, c] = a[r + dr, c + dc] (cyclic: the wrapped rim is never used by a step's write region)."""
    return np.roll(a, (-dr, -dc), (0, 1))


def rcd_2d(bayer, pattern):
    H, W = bayer.shape
    tab = np.array(TABLES[pattern])
    rr, cc = np.mgrid[0:H, 0:W]
    color = tab[rr & 1, cc & 1]
    cfa = np.maximum(f32(0), bayer.astype(f32))
    rgb = [np.where(color == k, cfa, f32(0)).astype(f32) for k in range(3)]
    rb_parity = (tab[np.arange(H) & 1, 0] & 1)[:, None]          # column parity of the R/B sites of each row: fc(row, 0) & 1
    is_rb = (cc & 1) == rb_parity
    is_green52 = (cc & 1) == (tab[np.arange(H) & 1, 1] & 1)[:, None]  # step 5.2's sites: fc(row, 1) & 1

    def region(r0, r1, c0, c1):  # inclusive bounds
        return (rr >= r0) & (rr <= r1) & (cc >= c0) & (cc <= c1)

    # ---- 1.1: squared high-pass of the colour differences, vertical / horizontal (rcd.cu:63-75)
    def hp(d):
        a = lambda k: sh(cfa, k * d[0], k * d[1])
        v = a(-3) - f32(3) * a(-2) - a(-1) + f32(6) * cfa - a(1) - f32(3) * a(2) + a(3)
        return v * v
    m = region(3, H - 4, 3, W - 4)
    v_diff = np.where(m, hp((1, 0)), f32(0)).astype(f32)
    h_diff = np.where(m, hp((0, 1)), f32(0)).astype(f32)

    # ---- 1.2: vertical / horizontal discrimination (rcd.cu:78-90)
    eps10 = f32(1e-10)
    v_stat = np.maximum(eps10, sh(v_diff, -1, 0) + v_diff + sh(v_diff, 1, 0))
    h_stat = np.maximum(eps10, sh(h_diff, 0, -1) + h_diff + sh(h_diff, 0, 1))
    vh_dir = np.where(region(2, H - 3, 2, W - 3), v_stat / (v_stat + h_stat), f32(0)).astype(f32)

    # ---- 2.1: low-pass at the R/B sites, rows [2, h-2], cols <= w-2 (rcd.cu:93-104).  lpf[r, c] is meaningful at R/B sites only
    lp = cfa + f32(0.5) * (sh(cfa, -1, 0) + sh(cfa, 1, 0) + sh(cfa, 0, -1) + sh(cfa, 0, 1)) \
        + f32(0.25) * (sh(cfa, -1, -1) + sh(cfa, -1, 1) + sh(cfa, 1, -1) + sh(cfa, 1, 1))
    lpf = np.where(is_rb & region(2, H - 2, 2, W - 2), lp, f32(0)).astype(f32)

    # ---- 3.1: green at the R/B sites, rows [4, h-5], cols [4, w-5] (rcd.cu:107-146)
    def refined(dirp):  # central value, or the mean of the 4 diagonal ones when that is further from 0.5
        nb = f32(0.25) * (sh(dirp, -1, -1) + sh(dirp, -1, 1) + sh(dirp, 1, -1) + sh(dirp, 1, 1))
        return np.where(np.abs(f32(0.5) - dirp) < np.abs(f32(0.5) - nb), nb, dirp)
    eps5 = f32(1e-5)
    vh_disc = refined(vh_dir)
    c = lambda dr, dc: sh(cfa, dr, dc)
    n_grad = eps5 + np.abs(c(-1, 0) - c(1, 0)) + np.abs(cfa - c(-2, 0)) + np.abs(c(-1, 0) - c(-3, 0)) + np.abs(c(-2, 0) - c(-4, 0))
    s_grad = eps5 + np.abs(c(1, 0) - c(-1, 0)) + np.abs(cfa - c(2, 0)) + np.abs(c(1, 0) - c(3, 0)) + np.abs(c(2, 0) - c(4, 0))
    w_grad = eps5 + np.abs(c(0, -1) - c(0, 1)) + np.abs(cfa - c(0, -2)) + np.abs(c(0, -1) - c(0, -3)) + np.abs(c(0, -2) - c(0, -4))
    e_grad = eps5 + np.abs(c(0, 1) - c(0, -1)) + np.abs(cfa - c(0, 2)) + np.abs(c(0, 1) - c(0, 3)) + np.abs(c(0, 2) - c(0, 4))
    with np.errstate(divide='ignore', invalid='ignore'):
        two = lpf + lpf
        n_est = c(-1, 0) * two / (eps5 + lpf + sh(lpf, -2, 0))   # the same-colour site two rows up owns slot lidx - width
        s_est = c(1, 0) * two / (eps5 + lpf + sh(lpf, 2, 0))
        w_est = c(0, -1) * two / (eps5 + lpf + sh(lpf, 0, -2))
        e_est = c(0, 1) * two / (eps5 + lpf + sh(lpf, 0, 2))
        v_est = (s_grad * n_est + n_grad * s_est) / (n_grad + s_grad)
        h_est = (w_grad * e_est + e_grad * w_est) / (e_grad + w_grad)
        g_at_rb = (f32(1) - vh_disc) * v_est + vh_disc * h_est
    rgb[1] = np.where(is_rb & region(4, H - 5, 4, W - 5), g_at_rb, rgb[1]).astype(f32)

    # ---- 4.1: squared diagonal high-pass at every ODD column, rows [3, h-4], cols [3, w-4] (rcd.cu:149-163).
    # p_site[r, c] (c odd) = content of slot (r, c // 2); what step 4.1 leaves unwritten still holds the step-1.1 plane
    odd = (cc & 1) == 1
    slot_j = cc // 2
    alias_r, alias_c = rr // 2, (rr % 2) * (W // 2) + slot_j      # full-density site whose flat position slot (r, j) aliases
    p_site = v_diff[alias_r, alias_c]
    q_site = h_diff[alias_r, alias_c]
    p_new = (c(-3, -3) - c(-1, -1) - c(1, 1) + c(3, 3)) - f32(3) * (c(-2, -2) + c(2, 2)) + f32(6) * cfa
    q_new = (c(-3, 3) - c(-1, 1) - c(1, -1) + c(3, -3)) - f32(3) * (c(-2, 2) + c(2, -2)) + f32(6) * cfa
    m41 = odd & region(3, H - 4, 3, W - 4)
    p_site = np.where(m41, p_new * p_new, p_site).astype(f32)
    q_site = np.where(m41, q_new * q_new, q_site).astype(f32)

    # ---- 4.2: diagonal discrimination at the R/B sites, rows [2, h-3], cols [2, w-3] (rcd.cu:166-182): the asymmetric taps
    even = ~odd
    p_stat = np.where(even, sh(p_site, -1, -1) + sh(p_site, 0, 1) + sh(p_site, 1, 1), sh(p_site, -1, 0) + p_site + sh(p_site, 1, 2))
    q_stat = np.where(even, sh(q_site, -1, 1) + sh(q_site, 0, 1) + sh(q_site, 1, -1), sh(q_site, -1, 2) + q_site + sh(q_site, 1, 0))
    p_stat, q_stat = np.maximum(eps10, p_stat), np.maximum(eps10, q_stat)
    # PQ_dir shares its plane with lpf: an R/B site outside step 4.2's region keeps its lpf value (never read by 5.1's region)
    pq_dir = np.where(is_rb & region(2, H - 3, 2, W - 3), p_stat / (p_stat + q_stat), lpf).astype(f32)

    # ---- 5.1: the opposite colour at the R/B sites, rows [4, h-4], cols [4, w-4] (rcd.cu:185-226)
    pq_disc = refined(pq_dir)  # the 4 neighbours of pqidx2 / pqidx3 are the diagonal R/B sites, summed NW, NE, SW, SE
    g = rgb[1]
    for k in (0, 2):           # sites of colour 2 - k receive colour k
        ck = rgb[k]
        s = lambda dr, dc: sh(ck, dr, dc)
        gs = lambda dr, dc: sh(g, dr, dc)
        nw_grad = eps5 + np.abs(s(-1, -1) - s(1, 1)) + np.abs(s(-1, -1) - s(-3, -3)) + np.abs(g - gs(-2, -2))
        ne_grad = eps5 + np.abs(s(-1, 1) - s(1, -1)) + np.abs(s(-1, 1) - s(-3, 3)) + np.abs(g - gs(-2, 2))
        sw_grad = eps5 + np.abs(s(-1, 1) - s(1, -1)) + np.abs(s(1, -1) - s(3, -3)) + np.abs(g - gs(2, -2))
        se_grad = eps5 + np.abs(s(-1, -1) - s(1, 1)) + np.abs(s(1, 1) - s(3, 3)) + np.abs(g - gs(2, 2))
        nw_est, ne_est, sw_est, se_est = s(-1, -1) - gs(-1, -1), s(-1, 1) - gs(-1, 1), s(1, -1) - gs(1, -1), s(1, 1) - gs(1, 1)
        p_est = (nw_grad * se_est + se_grad * nw_est) / (nw_grad + se_grad)
        q_est = (ne_grad * sw_est + sw_grad * ne_est) / (ne_grad + sw_grad)
        val = g + ((f32(1) - pq_disc) * p_est + pq_disc * q_est)
        rgb[k] = np.where((color == 2 - k) & region(4, H - 4, 4, W - 4), val, ck).astype(f32)

    # ---- 5.2: red and blue at the green sites, rows [4, h-4], cols [4, w-4] (rcd.cu:229-282)
    vh_disc = refined(vh_dir)
    gs = lambda dr, dc: sh(g, dr, dc)
    n1, s1 = eps5 + np.abs(g - gs(-2, 0)), eps5 + np.abs(g - gs(2, 0))
    w1, e1 = eps5 + np.abs(g - gs(0, -2)), eps5 + np.abs(g - gs(0, 2))
    new = {}
    for k in (0, 2):
        ck = rgb[k]
        s = lambda dr, dc: sh(ck, dr, dc)
        sn_abs, ew_abs = np.abs(s(-1, 0) - s(1, 0)), np.abs(s(0, -1) - s(0, 1))
        n_grad = n1 + sn_abs + np.abs(s(-1, 0) - s(-3, 0))
        s_grad = s1 + sn_abs + np.abs(s(1, 0) - s(3, 0))
        w_grad = w1 + ew_abs + np.abs(s(0, -1) - s(0, -3))
        e_grad = e1 + ew_abs + np.abs(s(0, 1) - s(0, 3))
        n_est, s_est, w_est, e_est = s(-1, 0) - gs(-1, 0), s(1, 0) - gs(1, 0), s(0, -1) - gs(0, -1), s(0, 1) - gs(0, 1)
        v_est = (n_grad * s_est + s_grad * n_est) / (n_grad + s_grad)
        h_est = (e_grad * w_est + w_grad * e_est) / (e_grad + w_grad)
        new[k] = g + ((f32(1) - vh_disc) * v_est + vh_disc * h_est)
    m52 = is_green52 & region(4, H - 4, 4, W - 4)
    for k in (0, 2):
        rgb[k] = np.where(m52, new[k], rgb[k]).astype(f32)
    return np.maximum(np.stack(rgb, -1), f32(0))


@pytest.mark.parametrize('pattern', ['RGGB', 'BGGR', 'GRBG', 'GBRG'])
@pytest.mark.parametrize('size', [(32, 32), (41, 58), (64, 48)])
def test_rcd_2d_restatement_matches_the_oracle_bit_for_bit(oracle, scene, pattern, size):
    h, w = size
    bayer = oracle.mosaic(scene(h, w, 90 + h), oracle.PATTERNS[pattern])[:, :, 0]
    bayer[h // 2, w // 2] = -0.25          # negative sample: max(0, .) at the input
    bayer[h // 2 + 1, 9:12] = 0.0          # a flat zero run: eps-dominated ratios
    got = rcd_2d(bayer, pattern)
    ref = oracle.rcd(bayer, oracle.PATTERNS[pattern])
    assert got.dtype == np.float32
    inner = (slice(7, h - 7), slice(7, w - 7))   # write_output's region (rcd.cu:48-60); the ring is the border kernels' work
    assert np.array_equal(got[inner], ref[inner]), f'{(got[inner] != ref[inner]).sum()} of {got[inner].size} values differ'


def test_rcd_2d_asymmetric_taps_are_what_the_slot_arithmetic_does(oracle):
    """The coordinate map itself, against the oracle's exposed planes (oracle_rcd_planes: the reference's flat layout):
    slot (r, j) of the p plane == p_site[r, 2 j + 1], stale step-1.1 values included."""
    h, w = 24, 32
    rng = np.random.default_rng(11)
    bayer = rng.uniform(0.05, 0.9, (h, w)).astype(np.float32)
    _, pq, p, q = oracle.rcd_planes(bayer, oracle.RGGB)
    # rebuild p_site the 2-D way (steps 1.1 and 4.1 only)
    cfa = bayer
    rr, cc = np.mgrid[0:h, 0:w]
    c = lambda dr, dc: sh(cfa, dr, dc)
    a = lambda k: sh(cfa, k, 0)
    v = a(-3) - f32(3) * a(-2) - a(-1) + f32(6) * cfa - a(1) - f32(3) * a(2) + a(3)
    in11 = (rr >= 3) & (rr <= h - 4) & (cc >= 3) & (cc <= w - 4)
    v_diff = np.where(in11, v * v, f32(0)).astype(np.float32)
    p_site = v_diff[rr // 2, (rr % 2) * (w // 2) + cc // 2]
    p_new = (c(-3, -3) - c(-1, -1) - c(1, 1) + c(3, 3)) - f32(3) * (c(-2, -2) + c(2, 2)) + f32(6) * cfa
    p_site = np.where(((cc & 1) == 1) & in11, p_new * p_new, p_site).astype(np.float32)
    slots = np.asarray(p).reshape(-1)[: h * w // 2].reshape(h, w // 2)   # the flat plane's first half, as (row, slot)
    assert np.array_equal(slots, p_site[:, 1::2])
    # slots step 4.1 does not write hold same-call step-1.1 values: the last rows (aliasing rows (h - 3) // 2 ...) and, on odd
    # rows, slot 0 (site column 1 is outside [3, w - 4]; its flat position is column w / 2 of full-density row r // 2)
    assert (slots[h - 2, 3:-3] != 0).all() and slots[7, 0] == v_diff[3, w // 2] != 0
