"""Live-OpenCV comparisons of the hot path (SURVEY.md 8a' KLT-1/2/3, ST-1/2, DLT-1; 8d item 2; BASELINE.md 3 path B).

`cv2` is not installable in the build image (no network) and the reference holds no vectors of its own, so the OpenCV
boundary of the path -- the reference's call sites extractor.py:44-45,65-66 (calcOpticalFlowPyrLK), :107 (circle), :111
(goodFeaturesToTrack), :270 (triangulatePoints), loader.py:86 (bilateralFilter), bundle_adjuster.py:48 (Rodrigues) -- is
"parity unpinned" until a box with a real OpenCV runs these checks.  Everything here is written against the cv2 API only; the
device side is anything with VoContext's array API.  tests/test_gpu_live_cv2.py runs it on the GPU box when `import cv2`
finds a real OpenCV there; a CPU self-test runs the same code with the stub + oracle standing in for cv2, so that the
harness itself is known to work the day a real cv2 is present."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "oracle", "ref_stub")

LK = dict(winSize=(31, 31), maxLevel=3, criteria=(3, 30, 0.03))                     # reference extractor.py:16-19
ST = dict(maxCorners=1000, qualityLevel=0.03, minDistance=7, blockSize=31)          # reference extractor.py:21-24


def find_real_cv2():
    """a real OpenCV build or None.  The stub package (oracle/ref_stub/cv2) is kept off the search path while looking and a
    stub that is already imported is not mistaken for OpenCV."""
    saved_path = list(sys.path)
    saved_mod = sys.modules.pop("cv2", None)
    try:
        sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != STUB_DIR]
        try:
            mod = importlib.import_module("cv2")
        except Exception:
            return None
        if not hasattr(mod, "getBuildInformation") or not hasattr(mod, "__version__"):
            return None
        return mod
    finally:
        sys.path[:] = saved_path
        if saved_mod is not None:
            sys.modules["cv2"] = saved_mod
        elif "cv2" in sys.modules and not hasattr(sys.modules["cv2"], "getBuildInformation"):
            del sys.modules["cv2"]


def compare_pyramid(cv2, ctx, img):
    """KLT-1: padded pyramid levels and Scharr derivatives, bit-exact (integer arithmetic).  ctx holds `img` as its current frame."""
    n, pyr = cv2.buildOpticalFlowPyramid(img, LK["winSize"], LK["maxLevel"], withDerivatives=True)
    out = {}
    for lvl in range(n + 1):
        pad = LK["winSize"][0]
        im = np.asarray(pyr[2 * lvl])
        de = np.asarray(pyr[2 * lvl + 1])
        # OpenCV returns views into the padded buffers when the ROI carries its border; either way the interior is [h, w]
        g_im, g_de = ctx.pyramid_read(1, lvl)
        h, w = g_im.shape
        if im.shape != (h, w):
            im = im[pad:pad + h, pad:pad + w]
            de = de[pad:pad + h, pad:pad + w]
        out[lvl] = (bool(np.array_equal(im, g_im)), bool(np.array_equal(de.reshape(h, w, 2), g_de)))
    return out


def compare_klt(cv2, ctx, im0, im1, p0):
    """KLT-2/3: positions of status = 1 points <= 0.01 px for >= 99 %, status equal, err rel 1e-4.  ctx holds (im0, im1)."""
    q1, qs, qe = cv2.calcOpticalFlowPyrLK(im0, im1, p0.reshape(-1, 1, 2).astype(np.float32), None, **LK)
    q1, qs, qe = q1.reshape(-1, 2), qs.reshape(-1).astype(np.uint8), qe.reshape(-1)
    p1, st, err = ctx.klt_track(p0)
    ok = (qs == 1) & (st == 1)
    d = np.abs(p1 - q1).max(axis=1)
    rel_err = np.abs(err - qe)[ok] / np.maximum(np.abs(qe[ok]), 1e-6)
    return dict(n=len(p0), status_equal=float((qs == st).mean()), frac_within_0p01=float((d[ok] <= 0.01).mean()) if ok.any() else 1.0,
                max_diff_px=float(d[ok].max()) if ok.any() else 0.0, frac_bit_equal=float((d[ok] == 0).mean()) if ok.any() else 1.0,
                err_rel_p99=float(np.quantile(rel_err, 0.99)) if ok.any() else 0.0)


def compare_shi_tomasi(cv2, ctx, img, cur_pts, radius=7):
    """ST-1 (min-eigenvalue map, rel 1e-5 of its maximum), the exclusion mask (exact), ST-2 (ordered corner list: identical
    except swaps among near-ties |d eig| < 1e-6 max).  ctx holds `img` as its current frame."""
    eig_cv = cv2.cornerMinEigenVal(img, ST["blockSize"], ksize=3)
    mask = np.full(img.shape, 255, np.uint8)
    for uv in cur_pts:                                       # reference extractor.py:104-107
        cv2.circle(mask, (int(np.int32(uv[0])), int(np.int32(uv[1]))), radius, 0, -1)
    corners_cv = cv2.goodFeaturesToTrack(img, mask=mask, **ST)
    corners_cv = np.zeros((0, 2), np.float32) if corners_cv is None else np.asarray(corners_cv).reshape(-1, 2)
    corners = ctx.shi_tomasi(cur_pts if len(cur_pts) else None, mask_radius=radius)
    rep = dict(n_cv=len(corners_cv), n_ours=len(corners))
    if hasattr(ctx, "shi_tomasi_read"):
        eig, gmask, _ = ctx.shi_tomasi_read()
        rep["eig_rel"] = float(np.abs(eig - eig_cv).max() / max(float(eig_cv.max()), 1e-30))
        rep["mask_equal"] = bool(np.array_equal(gmask, mask))
    same_set = set(map(tuple, corners_cv.astype(int).tolist())) == set(map(tuple, np.asarray(corners).astype(int).tolist()))
    rep["same_set"] = bool(same_set)
    m = min(len(corners_cv), len(corners))
    rep["order_mismatches"] = int((np.asarray(corners)[:m] != corners_cv[:m]).any(axis=1).sum()) + abs(len(corners_cv) - len(corners))
    return rep


def compare_dlt(cv2, ctx, P0, P1, uv0, uv1):
    """DLT-1: dehomogenised points, |dX| / |X| <= 1e-4"""
    X4 = np.asarray(cv2.triangulatePoints(P0, P1, uv0.reshape(-1, 1, 2), uv1.reshape(-1, 1, 2))).reshape(4, -1)
    G4 = np.asarray(ctx.triangulate(P0, P1, uv0, uv1)).reshape(4, -1)
    X, G = (X4[:3] / X4[3]).T.astype(np.float64), (G4[:3] / G4[3]).T.astype(np.float64)
    return dict(max_rel=float((np.linalg.norm(X - G, axis=1) / np.linalg.norm(X, axis=1)).max()))


def compare_bilateral(cv2, ctx, img):
    """the loader's pre-filter (loader.py:16-20,86); OpenCV's SIMD rows may fuse the multiply-add -> allow 1 grey level"""
    a = cv2.bilateralFilter(img, 5, 1.5, 1.5)
    b = ctx.bilateral(img, 5, 1.5, 1.5)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return dict(max_diff=int(d.max()), frac_equal=float((d == 0).mean()))


def compare_rodrigues(cv2, vec_to_mat, mat_to_vec, seed=0):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for r in rng.normal(0, 1.0, (32, 3)):
        R = np.asarray(cv2.Rodrigues(r.reshape(3, 1))[0])
        worst = max(worst, float(np.abs(R - vec_to_mat(r)).max()))
        back = np.asarray(cv2.Rodrigues(R)[0]).reshape(3)
        worst = max(worst, float(np.abs(back - np.asarray(mat_to_vec(R)).reshape(3)).max()))
    return dict(max_abs=worst)


def time_reference_call_pattern(cv2, im0, im1, p_cand, p_land, cur_pts, P0, P1, uv0, uv1, repeats=3):
    """seconds per frame of the reference's OpenCV calls on one frame pair, in its call pattern: 4 x calcOpticalFlowPyrLK
    (extractor.py:44-45 on the candidates, :65-66 on the landmark keypoints -- each rebuilds both pyramids), the mask loop +
    goodFeaturesToTrack (:104-111) and triangulatePoints (:270).  -> dict of per-stage seconds (best of `repeats`)."""
    def best(f):
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        return min(ts)

    def klt4():
        for p in (p_cand, p_land):
            q, _, _ = cv2.calcOpticalFlowPyrLK(im0, im1, p.reshape(-1, 1, 2), None, **LK)
            cv2.calcOpticalFlowPyrLK(im0, im1, q, None, **LK)

    def redetect():
        mask = np.full(im1.shape, 255, np.uint8)
        for uv in cur_pts:
            cv2.circle(mask, (int(np.int32(uv[0])), int(np.int32(uv[1]))), 7, 0, -1)
        cv2.goodFeaturesToTrack(im1, mask=mask, **ST)

    def dlt():
        cv2.triangulatePoints(P0, P1, uv0.reshape(-1, 1, 2), uv1.reshape(-1, 1, 2))
    out = dict(klt_x4_s=best(klt4), redetect_s=best(redetect), dlt_s=best(dlt))
    out["frame_s"] = out["klt_x4_s"] + out["redetect_s"] + out["dlt_s"]
    try:
        out["threads"] = int(cv2.getNumThreads())
    except Exception:
        out["threads"] = None
    return out
