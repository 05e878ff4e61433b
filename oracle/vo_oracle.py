"""ctypes front-end of the CPU oracle (oracle/vo_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module (see the header of vo_oracle.c).  The product path never does.

The functions double as the backend of the `cv2` stub (oracle/ref_stub/cv2) so
the reference's glue code can be exercised on top of this restatement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvo_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "vo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, f32p, i32p, i16p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int16))
        L.vo_oracle_pyr_down.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.vo_oracle_pyr_down.restype = None
        L.vo_oracle_scharr.argtypes = [u8p, C.c_int, C.c_int, i16p]
        L.vo_oracle_scharr.restype = None
        L.vo_oracle_pyr_levels.argtypes = [C.c_int] * 4
        L.vo_oracle_pyr_levels.restype = C.c_int
        L.vo_oracle_klt.argtypes = [u8p, u8p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_double, C.c_float, C.c_int, f32p, u8p, f32p, i32p]
        L.vo_oracle_klt.restype = C.c_int
        L.vo_oracle_circle.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.vo_oracle_circle.restype = None
        L.vo_oracle_circle_rows.argtypes = [C.c_int, i32p]
        L.vo_oracle_circle_rows.restype = None
        L.vo_oracle_min_eig.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        L.vo_oracle_min_eig.restype = None
        L.vo_oracle_corner_response.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_double]
        L.vo_oracle_corner_response.restype = None
        L.vo_oracle_good_features.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                              C.c_int, C.c_int, f32p, f32p, i32p, C.c_int, C.c_double]
        L.vo_oracle_good_features.restype = C.c_int
        L.vo_oracle_triangulate.argtypes = [f32p, f32p, f32p, f32p, C.c_int, f32p]
        L.vo_oracle_bilateral.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, u8p]
        L.vo_oracle_bilateral.restype = C.c_int
        L.vo_oracle_triangulate.restype = None
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def pyr_down(img):
    img = _img(img)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().vo_oracle_pyr_down(_p(img, C.c_uint8), w, h, _p(out, C.c_uint8))
    return out


def scharr(img):
    img = _img(img)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().vo_oracle_scharr(_p(img, C.c_uint8), w, h, _p(out, C.c_int16))
    return out


def pyr_levels(w, h, win=31, max_level=3):
    return lib().vo_oracle_pyr_levels(w, h, win, max_level)


def build_pyramid(img, win=31, max_level=3):
    """-> list of uint8 levels [0..top]"""
    img = _img(img)
    top = pyr_levels(img.shape[1], img.shape[0], win, max_level)
    out = [img]
    for _ in range(top):
        out.append(pyr_down(out[-1]))
    return out


def klt(im0, im1, p0, winSize=(31, 31), maxLevel=3, criteria=(3, 30, 0.03), minEigThreshold=1e-4,
        acc_mode=1, return_iters=False):
    """Restatement of cv2.calcOpticalFlowPyrLK(im0, im1, p0, None, winSize, maxLevel, criteria).
    criteria = (type, maxCount, eps) with type = COUNT|EPS (both always active here)."""
    im0, im1 = _img(im0), _img(im1)
    assert im0.shape == im1.shape and winSize[0] == winSize[1]
    h, w = im0.shape
    p0 = np.ascontiguousarray(np.asarray(p0, np.float32).reshape(-1, 2))
    n = p0.shape[0]
    p1 = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    iters = np.zeros((n, maxLevel + 1), np.int32)
    lib().vo_oracle_klt(_p(im0, C.c_uint8), _p(im1, C.c_uint8), w, h, _p(p0, C.c_float), n, int(winSize[0]),
                        int(maxLevel), int(criteria[1]), float(criteria[2]), float(minEigThreshold),
                        int(acc_mode), _p(p1, C.c_float), _p(st, C.c_uint8), _p(err, C.c_float),
                        _p(iters, C.c_int32))
    if return_iters:
        return p1, st, err, iters
    return p1, st, err


def circle_mask(mask, center, radius, color=0):
    assert mask.dtype == np.uint8 and mask.flags.c_contiguous
    h, w = mask.shape
    lib().vo_oracle_circle(_p(mask, C.c_uint8), w, h, int(center[0]), int(center[1]), int(radius), int(color))
    return mask


def circle_rows(radius):
    hw = np.zeros(radius + 1, np.int32)
    lib().vo_oracle_circle_rows(radius, _p(hw, C.c_int32))
    return hw


def min_eig(img, block=31, exact_int=True):
    img = _img(img)
    h, w = img.shape
    out = np.empty((h, w), np.float32)
    lib().vo_oracle_min_eig(_p(img, C.c_uint8), w, h, block, _p(out, C.c_float), int(exact_int))
    return out


def harris(img, block=31, k=0.04, exact_int=True):
    """cv2.cornerHarris(img, block, 3, k): the response goodFeaturesToTrack(useHarrisDetector=True) ranks"""
    img = _img(img)
    h, w = img.shape
    out = np.empty((h, w), np.float32)
    lib().vo_oracle_corner_response(_p(img, C.c_uint8), w, h, block, _p(out, C.c_float), int(exact_int), 1, float(k))
    return out


def good_features(img, mask, maxCorners=1000, qualityLevel=0.03, minDistance=7, blockSize=31,
                  exact_int=True, return_aux=False, useHarrisDetector=False, k=0.04):
    img = _img(img)
    h, w = img.shape
    mp = None
    if mask is not None:
        mask = _img(mask)
        assert mask.shape == img.shape
        mp = _p(mask, C.c_uint8)
    cap = maxCorners if maxCorners > 0 else w * h
    out = np.zeros((cap, 2), np.float32)
    eig = np.empty((h, w), np.float32)
    nc = np.zeros(1, np.int32)
    n = lib().vo_oracle_good_features(_p(img, C.c_uint8), mp, w, h, int(maxCorners), float(qualityLevel),
                                      float(minDistance), int(blockSize), int(exact_int), _p(out, C.c_float),
                                      _p(eig, C.c_float), _p(nc, C.c_int32), 1 if useHarrisDetector else 0, float(k))
    if return_aux:
        return out[:n].copy(), eig, int(nc[0])
    return out[:n].copy()


def triangulate(P0, P1, uv0, uv1):
    """cv2.triangulatePoints restatement: P float32 3x4, uv (n,2) float32 -> X4 float32 (4, n)."""
    P0 = np.ascontiguousarray(P0, np.float32)
    P1 = np.ascontiguousarray(P1, np.float32)
    uv0 = np.ascontiguousarray(np.asarray(uv0, np.float32).reshape(-1, 2))
    uv1 = np.ascontiguousarray(np.asarray(uv1, np.float32).reshape(-1, 2))
    n = uv0.shape[0]
    X4 = np.zeros((4, n), np.float32)
    lib().vo_oracle_triangulate(_p(P0, C.c_float), _p(P1, C.c_float), _p(uv0, C.c_float), _p(uv1, C.c_float), n,
                                _p(X4, C.c_float))
    return X4


def bilateral(img, d=5, sigmaColor=1.5, sigmaSpace=1.5):
    """cv2.bilateralFilter(img, d, sigmaColor, sigmaSpace) for uint8 gray images (the loader's pre-filter,
    /root/reference/src/loader/loader.py:16-20,86)."""
    img = _img(img)
    h, w = img.shape
    out = np.empty_like(img)
    n = lib().vo_oracle_bilateral(_p(img, C.c_uint8), w, h, int(d), float(sigmaColor), float(sigmaSpace), _p(out, C.c_uint8))
    assert n > 0
    return out
