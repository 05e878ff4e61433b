"""Minimal stand-in for the `cv2` module -- TEST INFRASTRUCTURE ONLY.

OpenCV 4.4.0 (the reference's pin, /root/reference/setup/conda_env.yml:57,78,87)
is not installed in this image and cannot be fetched.  The reference's
`bundle_adjuster.py` (line 3) and `extractor/triangulate.py` (line 1) import
`cv2` at module scope, so to import those reference modules *in this container*
(to generate golden vectors, tests/golden/gen_golden.py) we put this package on
sys.path first.  It is our own code; nothing here is copied from OpenCV or the
reference.

What it provides:
  * Rodrigues            closed form, follows the published algorithm of
                         calib3d/calibration.cpp
  * the few constants / factory names Extractor.__init__ touches
    (/root/reference/src/extractor/extractor.py:16-36)
  * calcOpticalFlowPyrLK / goodFeaturesToTrack / circle / triangulatePoints /
    KeyPoint_convert / solvePnPRansac, which forward to a pluggable backend (`set_backend`) --
    the tests plug in the CPU oracle (oracle/vo_oracle.py) so that the
    reference's *glue* code (list bookkeeping, filters, grouping) can be run
    unmodified on top of our restatement of the OpenCV arithmetic.

It never ships in the product path and never travels as part of the product:
only tests/ and tests/golden/gen_golden.py import it.
"""
import numpy as np

TERM_CRITERIA_COUNT = 1
TERM_CRITERIA_MAX_ITER = 1
TERM_CRITERIA_EPS = 2
RANSAC = 8
IMREAD_GRAYSCALE = 0

_backend = None


def set_backend(b):
    """b must offer klt(im0, im1, p0, winSize, maxLevel, criteria),
    good_features(img, mask, maxCorners, qualityLevel, minDistance, blockSize),
    circle_mask(mask, center, radius) and triangulate(P0, P1, uv0, uv1)."""
    global _backend
    _backend = b


def _need_backend(name):
    if _backend is None:
        raise RuntimeError("cv2 stub: %s needs a backend (cv2.set_backend)" % name)
    return _backend


# ----------------------------------------------------------------------------
# Rodrigues
# ----------------------------------------------------------------------------
def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rodrigues_vec_to_mat(r):
    r = np.asarray(r, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    c, s = np.cos(theta), np.sin(theta)
    return c * np.eye(3) + (1.0 - c) * np.outer(k, k) + s * _skew(k)


def rodrigues_mat_to_vec(R):
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    # project onto SO(3) first, as OpenCV does (R <- U Vt)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(0.25 * (v @ v))
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5
    c = min(1.0, max(-1.0, c))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        rx = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        ry = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        rz = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        w = np.array([rx, ry, rz])
        n = np.linalg.norm(w)
        return w * (theta / n) if n > 0 else np.zeros(3)
    return v * (0.5 * theta / s)


def Rodrigues(src, dst=None, jacobian=None):
    a = np.asarray(src, dtype=np.float64)
    if a.size == 3:
        return rodrigues_vec_to_mat(a), None
    if a.shape == (3, 3):
        return rodrigues_mat_to_vec(a).reshape(3, 1), None
    raise ValueError("Rodrigues: bad input shape %r" % (a.shape,))


# ----------------------------------------------------------------------------
# names touched by Extractor.__init__ (never exercised on the hot path)
# ----------------------------------------------------------------------------
class _Unavailable:
    def __init__(self, what):
        self._what = what

    def __getattr__(self, name):
        raise RuntimeError("cv2 stub: %s.%s is outside the hot path" % (self._what, name))


def SIFT_create(*a, **k):
    return _Unavailable("SIFT")


def ORB_create(*a, **k):
    return _Unavailable("ORB")


def BFMatcher(*a, **k):
    return _Unavailable("BFMatcher")


# ----------------------------------------------------------------------------
# backend-forwarded OpenCV arithmetic
# ----------------------------------------------------------------------------
def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, winSize=(21, 21), maxLevel=3,
                         criteria=(3, 30, 0.01), flags=0, minEigThreshold=1e-4):
    if nextPts is not None or flags != 0:
        raise NotImplementedError("cv2 stub: only nextPts=None, flags=0")
    p1, st, err = _need_backend("calcOpticalFlowPyrLK").klt(
        prevImg, nextImg, np.asarray(prevPts, np.float32).reshape(-1, 2),
        winSize, maxLevel, criteria, minEigThreshold)
    n = p1.shape[0]
    return p1.reshape(n, 1, 2), st.reshape(n, 1), err.reshape(n, 1)


def goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, mask=None,
                        blockSize=3, useHarrisDetector=False, k=0.04):
    be = _need_backend("goodFeaturesToTrack")
    pts = (be.good_features(image, mask, maxCorners, qualityLevel, minDistance, blockSize, useHarrisDetector=True, k=k) if useHarrisDetector
           else be.good_features(image, mask, maxCorners, qualityLevel, minDistance, blockSize))
    if pts.shape[0] == 0:
        return None
    return pts.reshape(-1, 1, 2)


def circle(img, center, radius, color, thickness=1, lineType=8, shift=0):
    if thickness >= 0 or shift != 0:
        raise NotImplementedError("cv2 stub: filled circles only")
    cx, cy = int(np.asarray(center[0]).reshape(-1)[0]), int(np.asarray(center[1]).reshape(-1)[0])
    _need_backend("circle").circle_mask(img, (cx, cy), int(radius), int(np.asarray(color).reshape(-1)[0]))
    return img


def KeyPoint_convert(kp):
    # the reference only round-trips goodFeaturesToTrack output through this
    # (/root/reference/src/extractor/extractor.py:112); pass it through.
    return kp


def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2):
    return _need_backend("triangulatePoints").triangulate(
        np.asarray(projMatr1), np.asarray(projMatr2),
        np.asarray(projPoints1).reshape(-1, 2), np.asarray(projPoints2).reshape(-1, 2))


def buildOpticalFlowPyramid(img, winSize, maxLevel, withDerivatives=True, pyrBorder=4, derivBorder=0, tryReuseInputImage=True):
    """(maxLevel actually built, [level 0 image, level 0 derivatives, level 1 image, ...]) -- interiors only (no border)"""
    b = _need_backend("buildOpticalFlowPyramid")
    levels = b.build_pyramid(img, int(winSize[0]), int(maxLevel))
    out = []
    for lv in levels:
        out.append(lv)
        if withDerivatives:
            out.append(b.scharr(lv))
    return len(levels) - 1, out


def cornerMinEigenVal(src, blockSize, ksize=3, borderType=4):
    if ksize != 3:
        raise NotImplementedError
    return _need_backend("cornerMinEigenVal").min_eig(src, int(blockSize))


def bilateralFilter(src, d, sigmaColor, sigmaSpace, borderType=4):
    return _need_backend("bilateralFilter").bilateral(src, int(d), float(sigmaColor), float(sigmaSpace))


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, rvec=None, tvec=None, useExtrinsicGuess=False,
                   iterationsCount=100, reprojectionError=8.0, confidence=0.99, inliers=None, flags=0):
    """(retval, rvec (3,1), tvec (3,1), inliers (n,1) int32) -- the shapes the reference reads at extractor.py:182-191.
    The backend's `pnp_ransac(K, X, uv, thr, conf, max_iters)` is oracle/pnp_oracle.py (RANSAC draws are not OpenCV's:
    statistical parity); no consensus -> (False, None, None, None) like OpenCV."""
    if distCoeffs is not None or useExtrinsicGuess:
        raise NotImplementedError("cv2 stub: solvePnPRansac without distortion / extrinsic guess only")
    X = np.asarray(objectPoints, np.float32).reshape(-1, 3)
    uv = np.asarray(imagePoints, np.float32).reshape(-1, 2)
    r, t, inl = _need_backend("solvePnPRansac").pnp_ransac(np.asarray(cameraMatrix, np.float64), X, uv, float(reprojectionError),
                                                           float(confidence), int(iterationsCount))
    if r is None:
        return False, None, None, None
    return True, np.asarray(r, np.float64).reshape(3, 1), np.asarray(t, np.float64).reshape(3, 1), np.asarray(inl, np.int32).reshape(-1, 1)


def waitKey(delay=0):
    return -1
