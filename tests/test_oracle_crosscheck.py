"""A SECOND, independently written restatement of the integer / small-matrix parts of the OpenCV 4.4 arithmetic in
numpy / scipy.ndimage, compared with oracle/vo_oracle.c on the BASELINE shapes: pyrDown, Scharr derivatives, Sobel + 31x31 box +
minimum eigenvalue, the filled-circle rasteriser, two-view DLT.  It does NOT pin OpenCV (no cv2 and no OpenCV source in this image:
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
