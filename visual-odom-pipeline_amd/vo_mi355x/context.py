"""Array-native front end: numpy in, numpy out, one VoContext per (host thread, GPU).

Thin wrapper over the C ABI (include/vo_mi355x.h).  The drop-in classes in
extractor.py / bundle_adjuster.py are built on top of this.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BaParams, BaStats, KltParams, StParams, VoError, as_c, ptr


class VoContext:
    def __init__(self, width, height, max_pts=4096, device=0, max_level=3, win=31):
        self._L = _lib.load()
        self._h = C.c_void_p()
        rc = self._L.vo_ctx_create(device, width, height, max_pts, max_level, win, C.byref(self._h))
        if rc != 0:
            msg = self._L.vo_last_error(None)
            raise VoError(rc, msg.decode() if msg else "vo_ctx_create failed")
        self.width, self.height, self.max_pts = width, height, max_pts
        self.max_level, self.win, self.device = max_level, win, device
        self._st_max_corners = 1000
        self._klt_levels = max_level + 1

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.vo_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            msg = self._L.vo_last_error(self._h)
            raise VoError(rc, msg.decode() if msg else "")

    def sync(self):
        self._ck(self._L.vo_sync(self._h))

    # -- frames ---------------------------------------------------------------------------------
    def push_frame(self, img):
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.shape != (self.height, self.width):
            raise ValueError("push_frame: expected uint8 image of shape (%d, %d)" % (self.height, self.width))
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        self._ck(self._L.vo_frame_push(self._h, ptr(img, C.c_uint8), img.strides[0]))

    def upload_sequence(self, frames):
        frames = as_c(frames, np.uint8)
        assert frames.ndim == 3 and frames.shape[1:] == (self.height, self.width)
        self._ck(self._L.vo_seq_upload(self._h, ptr(frames, C.c_uint8), frames.shape[0]))

    def push_frame_resident(self, idx):
        self._ck(self._L.vo_frame_push_resident(self._h, int(idx)))

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        self._ck(self._L.vo_pyramid_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def pyramid_read(self, which, level):
        """-> (img uint8 [h,w], deriv int16 [h,w,2]) of the prev (which=0) / cur (which=1) frame."""
        w, h = self.level_size(level)
        img = np.empty((h, w), np.uint8)
        der = np.empty((h, w, 2), np.int16)
        self._ck(self._L.vo_pyramid_read(self._h, which, level, ptr(img, C.c_uint8), ptr(der, C.c_int16)))
        return img, der

    # -- KLT ------------------------------------------------------------------------------------
    def klt_params(self, win=None, max_level=None, max_count=30, epsilon=0.03, min_eig_threshold=1e-4):
        p = KltParams()
        self._L.vo_klt_default_params(C.byref(p))
        p.win = self.win if win is None else win
        p.max_level = self.max_level if max_level is None else max_level
        p.max_count, p.epsilon, p.min_eig_threshold = max_count, epsilon, min_eig_threshold
        return p

    def klt_track(self, p0, params=None, return_iters=False):
        """prev -> cur tracking.  p0 (n,2) float32 -> p1 (n,2) f32, status (n,) u8, err (n,) f32"""
        p0 = as_c(np.asarray(p0, np.float32).reshape(-1, 2), np.float32)
        n = p0.shape[0]
        prm = params if params is not None else self.klt_params()
        p1 = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        it = np.full((n, prm.max_level + 1), -1, np.int32)
        self._ck(self._L.vo_klt_track(self._h, ptr(p0, C.c_float), n, C.byref(prm), ptr(p1, C.c_float),
                                      ptr(st, C.c_uint8), ptr(err, C.c_float), ptr(it, C.c_int32)))
        if return_iters:
            return p1, st, err, it
        return p1, st, err

    def points_upload(self, p):
        p = as_c(np.asarray(p, np.float32).reshape(-1, 2), np.float32)
        self._ck(self._L.vo_points_upload(self._h, ptr(p, C.c_float), p.shape[0]))

    def points_download(self, n, return_iters=False):
        p = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        it = np.full((n, self._klt_levels), -1, np.int32) if return_iters else None
        self._ck(self._L.vo_points_download(self._h, ptr(p, C.c_float), ptr(st, C.c_uint8), ptr(err, C.c_float),
                                            ptr(it, C.c_int32), n))
        return (p, st, err, it) if return_iters else (p, st, err)

    def klt_track_resident(self, n, params=None):
        prm = params if params is not None else self.klt_params()
        self._klt_levels = prm.max_level + 1
        self._ck(self._L.vo_klt_track_resident(self._h, n, C.byref(prm)))

    # -- Shi-Tomasi -----------------------------------------------------------------------------
    def st_params(self, max_corners=1000, quality_level=0.03, min_distance=7, block_size=31):
        p = StParams()
        self._L.vo_st_default_params(C.byref(p))
        p.max_corners, p.quality_level, p.min_distance, p.block_size = max_corners, quality_level, min_distance, block_size
        return p

    def shi_tomasi(self, cur_pts=None, mask_radius=7, mask=None, params=None):
        """Re-detection on the CURRENT frame -> corners (m,2) float32 (integer-valued x,y)"""
        prm = params if params is not None else self.st_params()
        n_cur = 0
        pp = None
        if cur_pts is not None and len(cur_pts):
            cur_pts = as_c(np.asarray(cur_pts, np.float32).reshape(-1, 2), np.float32)
            n_cur, pp = cur_pts.shape[0], ptr(cur_pts, C.c_float)
        mp = None
        if mask is not None:
            mask = as_c(mask, np.uint8)
            assert mask.shape == (self.height, self.width)
            mp = ptr(mask, C.c_uint8)
        out = np.zeros((max(prm.max_corners, 1), 2), np.float32)
        n_out = C.c_int32(0)
        self._ck(self._L.vo_shi_tomasi(self._h, pp, n_cur, int(mask_radius), mp, C.byref(prm), ptr(out, C.c_float),
                                       C.byref(n_out)))
        return out[:n_out.value].copy()

    def shi_tomasi_resident(self, n_cur, mask_radius=7, params=None):
        prm = params if params is not None else self.st_params()
        self._st_max_corners = prm.max_corners
        self._ck(self._L.vo_shi_tomasi_resident(self._h, n_cur, int(mask_radius), C.byref(prm)))

    def shi_tomasi_fetch(self):
        out = np.zeros((max(self._st_max_corners, 1), 2), np.float32)
        n_out = C.c_int32(0)
        self._ck(self._L.vo_shi_tomasi_fetch(self._h, ptr(out, C.c_float), C.byref(n_out)))
        return out[:n_out.value].copy()

    def shi_tomasi_read(self):
        eig = np.empty((self.height, self.width), np.float32)
        mask = np.empty((self.height, self.width), np.uint8)
        nc = C.c_int32(0)
        self._ck(self._L.vo_shi_tomasi_read(self._h, ptr(eig, C.c_float), ptr(mask, C.c_uint8), C.byref(nc)))
        return eig, mask, nc.value

    # -- DLT ------------------------------------------------------------------------------------
    def triangulate(self, P0, P1, uv0, uv1, K=None, H0=None, H1=None):
        """cv2.triangulatePoints replacement.  -> X4 (4,n) f32 [, depth1 (n,) f64, reproj (n,) f64]"""
        P0, P1 = as_c(P0, np.float32), as_c(P1, np.float32)
        uv0 = as_c(np.asarray(uv0, np.float32).reshape(-1, 2), np.float32)
        uv1 = as_c(np.asarray(uv1, np.float32).reshape(-1, 2), np.float32)
        n = uv0.shape[0]
        assert P0.shape == (3, 4) and P1.shape == (3, 4) and uv1.shape[0] == n
        X4 = np.zeros((4, n), np.float32)
        if K is None:
            self._ck(self._L.vo_triangulate_dlt(self._h, ptr(P0, C.c_float), ptr(P1, C.c_float), ptr(uv0, C.c_float),
                                                ptr(uv1, C.c_float), n, ptr(X4, C.c_float), None, None, None, None, None))
            return X4
        K, H0, H1 = as_c(K, np.float64), as_c(H0, np.float64), as_c(H1, np.float64)
        depth = np.zeros(n, np.float64)
        reproj = np.zeros(n, np.float64)
        self._ck(self._L.vo_triangulate_dlt(self._h, ptr(P0, C.c_float), ptr(P1, C.c_float), ptr(uv0, C.c_float),
                                            ptr(uv1, C.c_float), n, ptr(X4, C.c_float), ptr(K, C.c_double),
                                            ptr(H0, C.c_double), ptr(H1, C.c_double), ptr(depth, C.c_double),
                                            ptr(reproj, C.c_double)))
        return X4, depth, reproj

    def dlt_upload(self, P0, P1, uv0, uv1, K=None, H0=None, H1=None):
        P0, P1 = as_c(P0, np.float32), as_c(P1, np.float32)
        uv0 = as_c(np.asarray(uv0, np.float32).reshape(-1, 2), np.float32)
        uv1 = as_c(np.asarray(uv1, np.float32).reshape(-1, 2), np.float32)
        self._dlt_n, self._dlt_stats = uv0.shape[0], K is not None
        if K is not None:
            K, H0, H1 = as_c(K, np.float64), as_c(H0, np.float64), as_c(H1, np.float64)
        d = C.c_double
        self._ck(self._L.vo_dlt_upload(self._h, ptr(P0, C.c_float), ptr(P1, C.c_float), ptr(uv0, C.c_float),
                                       ptr(uv1, C.c_float), self._dlt_n, ptr(K, d), ptr(H0, d), ptr(H1, d)))

    def dlt_resident(self):
        self._ck(self._L.vo_dlt_resident(self._h))

    def dlt_fetch(self):
        n = self._dlt_n
        X4, depth, reproj = np.zeros((4, n), np.float32), np.zeros(n), np.zeros(n)
        self._ck(self._L.vo_dlt_fetch(self._h, ptr(X4, C.c_float), ptr(depth, C.c_double), ptr(reproj, C.c_double)))
        return (X4, depth, reproj) if self._dlt_stats else X4

    # -- fused frame step ----------------------------------------------------------------------
    def set_graph_mode(self, on=True):
        self._ck(self._L.vo_set_graph_mode(self._h, 1 if on else 0))

    def frame_step_resident(self, frame_idx, n_pts, do_dlt=True, do_ba=True, do_st=True, mask_radius=7,
                            klt=None, st=None, ba=None):
        """enqueue one whole frame (pyramid, KLT, DLT, BA, re-detection, result copies); async"""
        klt = klt if klt is not None else self.klt_params()
        st = st if st is not None else self.st_params()
        ba = ba if ba is not None else self.ba_params()
        self._klt_levels = klt.max_level + 1
        self._st_max_corners = st.max_corners
        self._step_cfg = (n_pts, do_dlt, do_ba, do_st)
        self._ck(self._L.vo_frame_step_resident(self._h, int(frame_idx), int(n_pts), int(do_dlt), int(do_ba), int(do_st),
                                                int(mask_radius), C.byref(klt), C.byref(st), C.byref(ba)))

    def frame_fetch(self):
        """wait for the enqueued frame and return its results as a dict of numpy arrays"""
        n_pts, do_dlt, do_ba, do_st = self._step_cfg
        out = {}
        p, stt, err = np.zeros((n_pts, 2), np.float32), np.zeros(n_pts, np.uint8), np.zeros(n_pts, np.float32)
        X4 = depth = reproj = poses = points = corners = None
        bs = BaStats()
        nc = C.c_int32(0)
        if do_dlt:
            n = self._dlt_n
            X4, depth, reproj = np.zeros((4, n), np.float32), np.zeros(n), np.zeros(n)
        if do_ba:
            W, N = self._ba_shape
            poses, points = np.zeros((W, 6)), np.zeros((N, 3))
        if do_st:
            corners = np.zeros((max(self._st_max_corners, 1), 2), np.float32)
        d = C.c_double
        self._ck(self._L.vo_frame_fetch(self._h, n_pts, ptr(p, C.c_float), ptr(stt, C.c_uint8), ptr(err, C.c_float),
                                        ptr(X4, C.c_float), ptr(depth, d), ptr(reproj, d), ptr(poses, d), ptr(points, d),
                                        C.byref(bs), ptr(corners, C.c_float), C.byref(nc) if do_st else None))
        out.update(points2d=p, status=stt, err=err)
        if do_dlt:
            out.update(X4=X4, depth1=depth, reproj=reproj)
        if do_ba:
            out.update(poses=poses, landmarks=points, ba_stats=self._stats(bs))
        if do_st:
            out["corners"] = corners[:nc.value].copy()
        return out

    # -- in-stream timing -----------------------------------------------------------------------
    PROF_FRAME, PROF_KLT, PROF_ST, PROF_DLT, PROF_BA = range(5)

    def profile_enable(self, regions=(0, 1, 2, 3, 4)):
        """regions: iterable of PROF_* ids to time with hipEvent pairs on the ctx stream; () switches timing off"""
        mask = 0
        for r in regions:
            mask |= 1 << r
        self._ck(self._L.vo_profile_enable(self._h, mask))

    def profile_read(self, region):
        t, n = C.c_double(0), C.c_int32(0)
        self._ck(self._L.vo_profile_read(self._h, region, C.byref(t), C.byref(n)))
        return t.value, n.value

    def debug_cycles(self, which):
        out = np.zeros(8, np.int64)
        self._ck(self._L.vo_debug_cycles(self._h, which, ptr(out, C.c_int64)))
        return out

    # -- BA -------------------------------------------------------------------------------------
    def ba_params(self, max_iters=50, ftol=1e-3, xtol=1e-3, gtol=1e-8, lambda0=1e-4, huber_delta=1.0):
        p = BaParams()
        self._L.vo_ba_default_params(C.byref(p))
        p.max_iters, p.ftol, p.xtol, p.gtol, p.lambda0, p.huber_delta = max_iters, ftol, xtol, gtol, lambda0, huber_delta
        return p

    @staticmethod
    def _stats(s):
        return dict(cost0=s.cost0, cost=s.cost, lam=s.lam, iters=s.iters, accepted=s.accepted, status=s.status,
                    n_obs=s.n_obs)

    def ba_adjust(self, K, poses, points, obs, params=None):
        """poses (W,6) [rvec,tvec; slot 0 newest], points (N,3), obs (W,N,2) NaN = unobserved.
        -> poses (W,6), points (N,3), stats dict"""
        K, poses, points, obs = as_c(K, np.float64), as_c(poses, np.float64), as_c(points, np.float64), as_c(obs, np.float64)
        W, N = obs.shape[:2]
        assert poses.shape == (W, 6) and points.shape == (N, 3) and obs.shape == (W, N, 2) and K.shape == (3, 3)
        prm = params if params is not None else self.ba_params()
        po, pt, st = np.zeros_like(poses), np.zeros_like(points), BaStats()
        self._ck(self._L.vo_ba_adjust(self._h, ptr(K, C.c_double), ptr(poses, C.c_double), ptr(points, C.c_double),
                                      ptr(obs, C.c_double), W, N, C.byref(prm), ptr(po, C.c_double),
                                      ptr(pt, C.c_double), C.byref(st)))
        return po, pt, self._stats(st)

    def ba_upload(self, K, poses, points, obs):
        K, poses, points, obs = as_c(K, np.float64), as_c(poses, np.float64), as_c(points, np.float64), as_c(obs, np.float64)
        W, N = obs.shape[:2]
        assert poses.shape == (W, 6) and points.shape == (N, 3) and obs.shape == (W, N, 2) and K.shape == (3, 3)
        self._ba_shape = (W, N)
        self._ck(self._L.vo_ba_upload(self._h, ptr(K, C.c_double), ptr(poses, C.c_double), ptr(points, C.c_double),
                                      ptr(obs, C.c_double), W, N))

    def ba_solve_resident(self, params=None):
        prm = params if params is not None else self.ba_params()
        self._ck(self._L.vo_ba_solve_resident(self._h, C.byref(prm)))

    def ba_fetch(self):
        W, N = self._ba_shape
        po, pt, st = np.zeros((W, 6)), np.zeros((N, 3)), BaStats()
        self._ck(self._L.vo_ba_fetch(self._h, ptr(po, C.c_double), ptr(pt, C.c_double), C.byref(st)))
        return po, pt, self._stats(st)

    def ba_probe(self, lam=1e-4, huber_delta=1.0):
        """Parity probe at the uploaded x0: residuals, normal equations, reduced system, one LM step."""
        W, N = self._ba_shape
        res = np.zeros(W * N)
        n_obs = C.c_int32(0)
        cost = C.c_double(0)
        Hpp, gp = np.zeros((W, 6, 6)), np.zeros((W, 6))
        Hll, gl = np.zeros((N, 3, 3)), np.zeros((N, 3))
        S, rhs = np.zeros((6 * W, 6 * W)), np.zeros(6 * W)
        dp, dl = np.zeros((W, 6)), np.zeros((N, 3))
        d = C.c_double
        self._ck(self._L.vo_ba_probe(self._h, lam, huber_delta, ptr(res, d), C.byref(n_obs), C.byref(cost),
                                     ptr(Hpp, d), ptr(gp, d), ptr(Hll, d), ptr(gl, d), ptr(S, d), ptr(rhs, d),
                                     ptr(dp, d), ptr(dl, d)))
        return dict(residual=res[:n_obs.value].copy(), cost=cost.value, Hpp=Hpp, gp=gp, Hll=Hll, gl=gl, S=S, rhs=rhs,
                    dposes=dp, dpoints=dl)
