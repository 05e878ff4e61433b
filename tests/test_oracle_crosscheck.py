"""A SECOND, independently written restatement of the integer / small-matrix parts of the OpenCV 4.4 arithmetic in
numpy / scipy.ndimage, compared with oracle/vo_oracle.c on the BASELINE shapes: pyrDown, Scharr derivatives, Sobel + 31x31 box +
minimum eigenvalue, the filled-circle rasteriser, two-view DLT, and (round 5) the LK iteration itself, the greedy minimum-distance corner
selection and the bilateral pre-filter.  It does NOT pin OpenCV (no cv2 and no OpenCV source in this image:
that boundary stays unpinned); what it rules out is a transcription slip in the single C file that a bit-exact
GPU-vs-oracle comparison could never see, because both sides would share it.  The formulations differ on purpose: library
correlations with `mirror` boundaries instead of per-pixel index reflection, a closed-form circle instead of the incremental
midpoint loop, numpy's SVD instead of the hand-written Jacobi.

Also here: the C oracle rebuilt with -fsanitize=address,undefined and driven through its whole API on odd shapes."""
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy import ndimage

import vo_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(376, 1241), (94, 311), (47, 156), (33, 40)]            # BASELINE level 0 / 2 / 3 of config A and a tiny odd one


def _img(shape, seed):
    from vo_mi355x import synthetic as syn
    h, w = shape
    return np.clip(np.rint(syn.make_texture(h, w, seed)), 0, 255).astype(np.uint8)


# ---- pyrDown: [1 4 6 4 1] x [1 4 6 4 1] / 256 with rounding, BORDER_REFLECT_101, output ((w + 1) / 2, (h + 1) / 2) -------------
def pyr_down_np(img):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    a = ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")       # scipy 'mirror' = d c b | a b c d | c b a = REFLECT_101
    a = ndimage.correlate1d(a, k, axis=0, mode="mirror")
    return ((a[::2, ::2] + 128) >> 8).astype(np.uint8)


@pytest.mark.parametrize("shape", SHAPES)
def test_pyr_down_independent(shape):
    img = _img(shape, 11)
    assert np.array_equal(o.pyr_down(img), pyr_down_np(img))
    lv = img
    for _ in range(3):                                   # a whole pyramid: errors would compound
        a, b = o.pyr_down(lv), pyr_down_np(lv)
        assert a.shape == ((lv.shape[0] + 1) // 2, (lv.shape[1] + 1) // 2) and np.array_equal(a, b)
        lv = a
        if min(lv.shape) < 8:
            break


# ---- Scharr: Ix = [3 10 3]^T (x) [-1 0 1], Iy = [-1 0 1]^T (x) [3 10 3], int16, REFLECT_101 ---------------------------------
def scharr_np(img):
    a = img.astype(np.int32)
    sm, df = np.array([3, 10, 3], np.int32), np.array([-1, 0, 1], np.int32)
    ix = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    iy = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=1, mode="mirror"), df, axis=0, mode="mirror")
    return np.stack([ix, iy], -1).astype(np.int16)


@pytest.mark.parametrize("shape", SHAPES)
def test_scharr_independent(shape):
    img = _img(shape, 12)
    d = o.scharr(img)
    assert d.dtype == np.int16 and np.array_equal(d, scharr_np(img))
    assert np.abs(d).max() <= 16 * 255                   # what the 4x derivative store of the GPU relies on (|.| <= 4080 -> x 4 fits int16)


# ---- cornerMinEigenVal: Sobel 3x3 scaled by 1 / (4 * 31 * 255), products, 31x31 box SUM (reflect-101), eigenvalue --------------
def min_eig_np(img, block=31):
    a = img.astype(np.int64)
    sm, df = np.array([1, 2, 1], np.int64), np.array([-1, 0, 1], np.int64)
    dx = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=1, mode="mirror"), df, axis=0, mode="mirror")
    box = np.ones(block, np.int64)

    def boxsum(p):                                       # exact integer window sums
        return ndimage.correlate1d(ndimage.correlate1d(p, box, axis=1, mode="mirror"), box, axis=0, mode="mirror")
    sxx, sxy, syy = boxsum(dx * dx), boxsum(dx * dy), boxsum(dy * dy)
    s = 1.0 / (4.0 * block * 255.0)
    A, B, C = sxx * s * s * 0.5, sxy * s * s, syy * s * s * 0.5
    return (A + C) - np.sqrt((A - C) ** 2 + B * B)       # float64


@pytest.mark.parametrize("shape", SHAPES[:3])
def test_min_eig_independent(shape):
    img = _img(shape, 13)
    ref = min_eig_np(img)
    e_int = o.min_eig(img, 31, exact_int=True)
    e_flt = o.min_eig(img, 31, exact_int=False)
    scale = np.abs(ref).max()
    assert np.abs(e_int - ref).max() <= 2e-6 * scale     # float32 evaluation of the same integer sums
    assert np.abs(e_flt - ref).max() <= 2e-5 * scale     # OpenCV's float running sums (SURVEY ST-1: rel 1e-5 of the map maximum)


# ---- calcOpticalFlowPyrLK, the iteration itself: a second formulation --------------------------------
# Written against the description, not against vo_oracle.c: whole-window numpy arithmetic on pre-padded level arrays (np.pad 'reflect' =
# BORDER_REFLECT_101 for the images, zeros for the derivatives) instead of per-pixel border functions, `np.rint` weights, `>>` descale on
# int64 arrays, exact integer window sums, np.float32 scalars for the 2 x 2 solve in OpenCV's expression order.  The pyramid and the
# derivatives come from the independent `pyr_down_np` / `scharr_np` above.  (Round-4 review: pyrDown, Scharr, min-eig, circle and DLT had
# a second restatement; the LK iteration and the greedy corner selection -- the intricate parts -- had not.)
def klt_np(im0, im1, p0, win=31, max_level=3, max_count=30, eps=0.03, min_eig_thr=1e-4):
    F = np.float32
    lv0, lv1 = [im0], [im1]
    while len(lv0) <= max_level:
        h, w = lv0[-1].shape
        if (w + 1) // 2 <= win or (h + 1) // 2 <= win:
            break
        lv0.append(pyr_down_np(lv0[-1])); lv1.append(pyr_down_np(lv1[-1]))
    top = len(lv0) - 1
    pad = win + 2
    I = [np.pad(a.astype(np.int64), pad, mode="reflect") for a in lv0]
    J = [np.pad(a.astype(np.int64), pad, mode="reflect") for a in lv1]
    D = [np.pad(scharr_np(a).astype(np.int64), ((pad, pad), (pad, pad), (0, 0))) for a in lv0]
    n = len(p0)
    p1 = np.zeros((n, 2), np.float32)
    status = np.ones(n, np.uint8)
    err = np.zeros(n, np.float32)
    iters = np.full((n, max_level + 1), -1, np.int32)
    half, scale20, eps2 = F((win - 1) * 0.5), F(1.0 / (1 << 20)), float(eps) * float(eps)

    def weights(a, b):
        w00 = int(np.rint((F(1) - a) * (F(1) - b) * F(1 << 14)))
        w01 = int(np.rint(a * (F(1) - b) * F(1 << 14)))
        w10 = int(np.rint((F(1) - a) * b * F(1 << 14)))
        return w00, w01, w10, (1 << 14) - w00 - w01 - w10

    def sample(P, wts, shift):                     # P: (win + 1, win + 1[, c]) int64
        s = P[:-1, :-1] * wts[0] + P[:-1, 1:] * wts[1] + P[1:, :-1] * wts[2] + P[1:, 1:] * wts[3]
        return (s + (1 << (shift - 1))) >> shift

    def window(A, ix, iy):
        return A[iy + pad: iy + pad + win + 1, ix + pad: ix + pad + win + 1]

    for level in range(top, -1, -1):
        rows, cols = lv0[level].shape
        for pt in range(n):
            prevx, prevy = F(p0[pt, 0]) * F(1.0 / (1 << level)), F(p0[pt, 1]) * F(1.0 / (1 << level))
            if level == top:
                nextx, nexty = prevx, prevy
            else:
                nextx, nexty = p1[pt, 0] * F(2), p1[pt, 1] * F(2)
            p1[pt] = (nextx, nexty)
            prevx, prevy = prevx - half, prevy - half
            ipx, ipy = int(np.floor(prevx)), int(np.floor(prevy))
            if ipx < -win or ipx >= cols or ipy < -win or ipy >= rows:
                if level == 0:
                    status[pt], err[pt] = 0, 0
                continue
            wts = weights(prevx - F(ipx), prevy - F(ipy))
            Iw = sample(window(I[level], ipx, ipy), wts, 14 - 5)
            dI = sample(window(D[level], ipx, ipy), wts, 14)
            gx, gy = dI[..., 0], dI[..., 1]
            A11, A12, A22 = F(int((gx * gx).sum())) * scale20, F(int((gx * gy).sum())) * scale20, F(int((gy * gy).sum())) * scale20
            det = A11 * A22 - A12 * A12
            min_eig = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + F(4) * A12 * A12)) / F(2 * win * win)
            if min_eig < F(min_eig_thr) or det < F(1.1920929e-07):
                if level == 0:
                    status[pt] = 0
                continue
            det = F(1) / det
            nextx, nexty = nextx - half, nexty - half
            pdx = pdy = F(0)
            j = 0
            while j < max_count:
                inx, iny = int(np.floor(nextx)), int(np.floor(nexty))
                if inx < -win or inx >= cols or iny < -win or iny >= rows:
                    if level == 0:
                        status[pt] = 0
                    break
                diff = sample(window(J[level], inx, iny), weights(nextx - F(inx), nexty - F(iny)), 14 - 5) - Iw
                b1, b2 = F(int((diff * gx).sum())) * scale20, F(int((diff * gy).sum())) * scale20
                dx = (A12 * b2 - A22 * b1) * det
                dy = (A12 * b1 - A11 * b2) * det
                nextx, nexty = nextx + dx, nexty + dy
                p1[pt] = (nextx + half, nexty + half)
                if float(dx) * float(dx) + float(dy) * float(dy) <= eps2:
                    j += 1
                    break
                if j > 0 and abs(float(dx + pdx)) < 0.01 and abs(float(dy + pdy)) < 0.01:
                    p1[pt, 0] -= dx * F(0.5); p1[pt, 1] -= dy * F(0.5)
                    j += 1
                    break
                pdx, pdy = dx, dy
                j += 1
            iters[pt, level] = j
            if status[pt] and level == 0:
                nx, ny = p1[pt, 0] - half, p1[pt, 1] - half
                inx, iny = int(np.floor(nx)), int(np.floor(ny))
                if inx < -win or inx >= cols or iny < -win or iny >= rows:
                    status[pt] = 0
                    continue
                diff = sample(window(J[level], inx, iny), weights(nx - F(inx), ny - F(iny)), 14 - 5) - Iw
                err[pt] = F(int(np.abs(diff).sum())) * F(1) / F(32 * win * win)
    return p1, status, err, iters


def _klt_points(w, h, n, seed):
    """points inside, on and beyond the border (windows that hang over the edge, points whose window leaves the image on the way)"""
    rng = np.random.default_rng(seed)
    p = np.stack([rng.uniform(-20, w + 20, n), rng.uniform(-20, h + 20, n)], 1)
    p[: n // 2] = np.stack([rng.uniform(20, w - 20, n // 2), rng.uniform(20, h - 20, n // 2)], 1)
    p[n // 2] = (0.0, 0.0); p[n // 2 + 1] = (w - 1.0, h - 1.0); p[n // 2 + 2] = (w - 0.5, 3.25)
    return p.astype(np.float32)


@pytest.mark.parametrize("case", ["baseline_1241x376", "truncated_pyramid_320x240", "flat_and_noisy_200x120"])
def test_lk_iteration_independent(case):
    """positions, status, err and the per-level iteration counts of oracle/vo_oracle.c == the numpy formulation, bit for bit"""
    from vo_mi355x import synthetic as syn
    if case == "baseline_1241x376":
        fr, _ = syn.make_sequence(2)
        p0 = _klt_points(1241, 376, 160, 3)
    elif case == "truncated_pyramid_320x240":
        fr, _ = syn.make_sequence(2, w=320, h=240, seed=77, margin=64)
        p0 = _klt_points(320, 240, 120, 4)
    else:
        rng = np.random.default_rng(5)                       # half of the image is flat (min-eigenvalue rejections), the other half noise
        a = np.full((120, 200), 90, np.uint8)                # (no convergence within 30 iterations, the oscillation rule)
        a[:, 100:] = rng.integers(0, 256, (120, 100))
        b = np.roll(a, (1, 2), (0, 1)); b[:, 100:] = np.clip(b[:, 100:].astype(int) + rng.integers(-30, 31, (120, 100)), 0, 255)
        fr = np.stack([a, b.astype(np.uint8)])
        p0 = _klt_points(200, 120, 120, 6)
    q1, qs, qe, qi = o.klt(fr[0], fr[1], p0, return_iters=True)
    r1, rs, re_, ri = klt_np(fr[0], fr[1], p0)
    assert np.array_equal(qs, rs) and np.array_equal(qi, ri)
    assert np.array_equal(q1.view(np.uint32), r1.view(np.uint32)) and np.array_equal(qe.view(np.uint32), re_.view(np.uint32))
    assert 0 < qs.sum() < len(qs)                            # both outcomes occur
    if case == "flat_and_noisy_200x120":
        assert (qi == 30).any() or (qi[:, 0] > 5).any()      # long runs occur: the exit rules inside the loop are exercised


# ---- goodFeaturesToTrack after the eigenvalue map: a second formulation ------------------------------
# threshold, 3 x 3 non-maximum test (scipy's maximum_filter instead of nine compares), ordering by one lexsort, and the greedy minimum
# distance rule by BRUTE FORCE against every corner accepted so far (float32 squared distance compared as float64, OpenCV's types) in
# place of the cell grid: the grid only prunes the search -- two pixels closer than minDistance lie in adjacent cells of that size.
def good_features_np(eig, mask, max_corners=1000, quality=0.03, min_distance=7.0):
    h, w = eig.shape
    m = np.ones((h, w), bool) if mask is None else mask != 0
    thr = np.float32(float(eig[m].max()) * quality)
    te = np.where(eig > thr, eig, np.float32(0))
    loc = (te != 0) & (te == ndimage.maximum_filter(te, size=3, mode="nearest")) & m
    loc[0, :] = loc[-1, :] = False; loc[:, 0] = loc[:, -1] = False      # the scan covers the interior only
    idx = np.flatnonzero(loc)
    order = np.lexsort((-idx, -te.ravel()[idx]))                        # value descending, then address descending
    idx = idx[order]
    ys, xs = (idx // w).astype(np.float32), (idx % w).astype(np.float32)
    out = np.zeros((0, 2), np.float32)
    md2 = float(min_distance) * float(min_distance)
    for x, y in zip(xs, ys):
        if len(out):
            dx, dy = x - out[:, 0], y - out[:, 1]
            if ((dx * dx + dy * dy).astype(np.float64) < md2).any():
                continue
        out = np.concatenate([out, np.array([[x, y]], np.float32)])
        if len(out) == max_corners:
            break
    return out, len(idx)


@pytest.mark.parametrize("shape,seed,n_discs", [((376, 1241), 21, 2000), ((376, 1241), 22, 0), ((94, 311), 23, 150), ((60, 64), 24, 6)])
def test_corner_selection_independent(shape, seed, n_discs):
    """ordered corner list and candidate count of vo_oracle_good_features == the brute-force formulation on the oracle's own eigenvalue map
    (the map has its own cross-check above), with the reference's exclusion discs (extractor.py:104-107) and without a mask"""
    img = _img(shape, seed)
    h, w = shape
    mask = None
    if n_discs:
        rng = np.random.default_rng(seed)
        mask = np.full(shape, 255, np.uint8)
        for x, y in zip(rng.integers(0, w, n_discs), rng.integers(0, h, n_discs)):
            o.circle_mask(mask, (int(x), int(y)), 7, 0)
    got, eig, nc = o.good_features(img, mask, return_aux=True)
    want, nc_np = good_features_np(eig, mask)
    assert nc == nc_np and got.shape == want.shape and np.array_equal(got, want)
    assert len(got) > 10
    # a tight budget and a plateau: maxCorners cuts the greedy loop; equal eigenvalues are ordered by address
    got5 = o.good_features(img, mask, maxCorners=5)
    assert np.array_equal(got5, want[:5])


def test_corner_selection_plateau_ties_independent():
    img = np.zeros((80, 96), np.uint8)
    img[20:60, 24:72] = 200                                   # a rectangle: its four corners give equal eigenvalue peaks by symmetry
    got, eig, nc = o.good_features(img, None, minDistance=3, return_aux=True)
    want, nc_np = good_features_np(eig, None, min_distance=3.0)
    assert nc == nc_np and np.array_equal(got, want) and len(got) >= 4


# ---- cv2.bilateralFilter(d = 5, sigmaColor = sigmaSpace = 1.5), the loader's pre-filter (loader.py:16-20, 86): whole-image form ---------
def bilateral_np(img, d=5, sigma_color=1.5, sigma_space=1.5):
    """taps of the disc r <= d / 2 in row-major order; float32 weights from float64 exponentials; per pixel sum += val * (space * colour),
    wsum += ..., in tap order, all float32; round half to even -- as array arithmetic over shifted views of a reflect-101 padded image"""
    F = np.float32
    radius = d // 2
    pad = np.pad(img, radius, mode="reflect").astype(np.int32)
    h, w = img.shape
    cw = np.exp(np.arange(256, dtype=np.float64) ** 2 * (-0.5 / (sigma_color * sigma_color))).astype(F)
    v0 = img.astype(np.int32)
    acc, wsum = np.zeros((h, w), F), np.zeros((h, w), F)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            r = np.sqrt(float(i * i + j * j))
            if r > radius:
                continue
            sw = F(np.exp(r * r * (-0.5 / (sigma_space * sigma_space))))
            v = pad[radius + i: radius + i + h, radius + j: radius + j + w]
            wt = sw * cw[np.abs(v - v0)]
            acc = acc + v.astype(F) * wt
            wsum = wsum + wt
    return np.rint(acc / wsum).astype(np.uint8)


@pytest.mark.parametrize("shape", [(376, 1241), (47, 156), (33, 40)])
def test_bilateral_independent(shape):
    img = _img(shape, 31)
    assert np.array_equal(o.bilateral(img), bilateral_np(img))
    rng = np.random.default_rng(32)
    noisy = rng.integers(0, 256, shape).astype(np.uint8)       # every colour distance occurs
    assert np.array_equal(o.bilateral(noisy), bilateral_np(noisy))


# ---- cornerHarris (goodFeaturesToTrack(useHarrisDetector=True, k)): the same box sums, response a c - b^2 - k (a + c)^2 ---------------
def harris_np(img, block=31, k=0.04):
    a = img.astype(np.int64)
    sm, df = np.array([1, 2, 1], np.int64), np.array([-1, 0, 1], np.int64)
    dx = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=1, mode="mirror"), df, axis=0, mode="mirror")
    box = np.ones(block, np.int64)

    def boxsum(p):
        return ndimage.correlate1d(ndimage.correlate1d(p, box, axis=1, mode="mirror"), box, axis=0, mode="mirror")
    s = 1.0 / (4.0 * block * 255.0)
    A, B, C = boxsum(dx * dx) * s * s, boxsum(dx * dy) * s * s, boxsum(dy * dy) * s * s
    return A * C - B * B - k * (A + C) ** 2                 # float64


@pytest.mark.parametrize("shape", SHAPES[:3])
def test_harris_independent(shape):
    img = _img(shape, 14)
    ref = harris_np(img)
    scale = np.abs(ref).max()
    assert np.abs(o.harris(img, 31, 0.04, exact_int=True) - ref).max() <= 3e-6 * scale
    assert np.abs(o.harris(img, 31, 0.04, exact_int=False) - ref).max() <= 5e-5 * scale
    got, resp, nc = o.good_features(img, None, return_aux=True, useHarrisDetector=True, k=0.04)
    want, nc_np = good_features_np(resp, None)              # the selection after the map is the same code path for both responses
    assert nc == nc_np and np.array_equal(got, want) and len(got) > 10
    assert not np.array_equal(got, o.good_features(img, None))      # ... and the two responses do rank differently


# ---- cv2.circle(mask, c, r, 0, -1): closed form of the midpoint circle instead of its incremental loop --------------------------
def circle_rows_np(r):
    """half-width of every row 0..r of the filled circle: the rasteriser walks the first octant with dx(dy) = floor(sqrt(r^2 - dy^2)) while
    dx >= dy and draws rows +-dy with half-width dx and rows +-dx with half-width dy (imgproc/drawing.cpp Circle, fill branch)"""
    import math
    hw = np.full(r + 1, -1, np.int64)
    for dy in range(r + 1):
        dx = math.isqrt(r * r - dy * dy)
        if dx < dy:
            break
        hw[dy] = max(hw[dy], dx)
        hw[dx] = max(hw[dx], dy)
    return hw


@pytest.mark.parametrize("r", list(range(0, 32)))
def test_circle_rows_independent(r):
    got = o.circle_rows(r)
    want = circle_rows_np(r)
    assert np.array_equal(np.asarray(got, np.int64), want), (r, got, want)


def test_circle_mask_clipping_independent():
    h, w, r = 40, 50, 7
    hw = circle_rows_np(r)
    for cx, cy in [(0, 0), (49, 39), (25, 20), (-3, 5), (52, 41), (3, -6), (-20, -20)]:
        m = np.full((h, w), 255, np.uint8)
        o.circle_mask(m, (cx, cy), r, 0)
        ys, xs = np.mgrid[0:h, 0:w]
        ady = np.abs(ys - cy)
        inside = (ady <= r) & (np.abs(xs - cx) <= hw[np.minimum(ady, r)])
        assert np.array_equal(m == 0, inside), (cx, cy)


# ---- cv2.triangulatePoints: last right-singular vector of the 4x4 DLT matrix ---------------------------------------------------------
def test_dlt_independent():
    from vo_mi355x import synthetic as syn
    s = syn.make_ba_scene(n_pts=400, n_slots=5, seed=4)
    K = s["K"]
    H = []
    for i in (4, 0):
        Hm = np.eye(4); Hm[:3, :3] = syn.rodrigues(s["poses_gt"][i, :3]); Hm[:3, 3] = s["poses_gt"][i, 3:]
        H.append(Hm)
    P0, P1 = np.float32(K @ H[0][:3]), np.float32(K @ H[1][:3])
    uv0, uv1 = s["obs"][4].astype(np.float32), s["obs"][0].astype(np.float32)
    X4 = o.triangulate(P0, P1, uv0, uv1)
    X = (X4[:3] / X4[3]).T.astype(np.float64)
    ref = np.empty_like(X)
    Q0, Q1, a0, a1 = P0.astype(np.float64), P1.astype(np.float64), uv0.astype(np.float64), uv1.astype(np.float64)   # float32 values, double arithmetic
    for i in range(len(uv0)):
        A = np.stack([a0[i, 0] * Q0[2] - Q0[0], a0[i, 1] * Q0[2] - Q0[1], a1[i, 0] * Q1[2] - Q1[0], a1[i, 1] * Q1[2] - Q1[1]])
        v = np.linalg.svd(A)[2][-1]
        ref[i] = v[:3] / v[3]
    assert (np.linalg.norm(X - ref, axis=1) <= 2e-6 * np.linalg.norm(ref, axis=1)).all()      # the output is rounded to float32
    assert np.abs(np.linalg.norm(X4, axis=0) - 1).max() <= 1e-6                                # unit-norm homogeneous vectors, like OpenCV


# ---- the C oracle under AddressSanitizer + UndefinedBehaviorSanitizer ----------------------------------------------------------------
def test_c_oracle_under_sanitizers(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan for this gcc")
    so = str(tmp_path / "libvo_oracle_san.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-std=c11", "-shared", "-o", so, os.path.join(ROOT, "oracle", "vo_oracle.c"), "-lm"])
    driver = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r]
import vo_oracle as o
o._SO = %r
o.build = lambda force=False: o._SO
from vo_mi355x import synthetic as syn
for (w, h) in ((161, 97), (64, 48), (33, 35)):
    fr, _ = syn.make_sequence(2, w=w, h=h, seed=5, margin=40)
    p0 = np.float32([[5, 5], [w - 2, h - 2], [w / 2, h / 2], [-3, 4], [w + 5, h + 9], [0, 0], [w, h]])
    p1, st, err = o.klt(fr[0], fr[1], p0)
    assert p1.shape == p0.shape
    m = np.full((h, w), 255, np.uint8)
    for x, y in np.int32(p1):
        o.circle_mask(m, (int(x), int(y)), 7, 0)
    c = o.good_features(fr[1], m)
    o.min_eig(fr[0], 31, True); o.min_eig(fr[0], 31, False)
    o.bilateral(fr[0])
    lv = o.build_pyramid(fr[0])
    for l in lv:
        o.scharr(l)
P = np.float32(np.hstack([np.eye(3), np.zeros((3, 1))])); P1 = P.copy(); P1[0, 3] = -1
o.triangulate(P, P1, np.float32([[0.1, 0.2], [0, 0]]), np.float32([[0.0, 0.2], [0, 0]]))
print("sanitized-ok")
''' % (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "visual-odom-pipeline_amd"), so)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
