"""TEST INFRASTRUCTURE / CPU BASELINE: a VoContext look-alike backed by the CPU oracle, so the drop-in Extractor / BundleAdjuster glue and the
table model of the closed loop can run without a GPU (`-m "not gpu"` tests; bench.py's cpu_baseline leg for `--workload pipeline`).  The product never imports it."""
from types import SimpleNamespace

import numpy as np

import ba_oracle as bo
import vo_oracle as o


class OracleContext:
    def __init__(self, width, height, max_pts=4096, device=0, max_level=3, win=31):
        self.width, self.height, self.max_level, self.win = width, height, max_level, win
        self._prev = self._cur = None

    def close(self):
        pass

    def push_frame(self, img):
        self._prev, self._cur = self._cur, np.ascontiguousarray(img, np.uint8).copy()

    def klt_params(self, win=None, max_level=None, max_count=30, epsilon=0.03, min_eig_threshold=1e-4):
        return SimpleNamespace(win=self.win if win is None else win, max_level=self.max_level if max_level is None else max_level,
                               max_count=max_count, epsilon=epsilon, min_eig_threshold=min_eig_threshold)

    def klt_track(self, p0, params=None, return_iters=False):
        prm = params or self.klt_params()
        p0 = np.asarray(p0, np.float32).reshape(-1, 2)
        if len(p0) == 0:
            return np.zeros((0, 2), np.float32), np.zeros(0, np.uint8), np.zeros(0, np.float32)
        return o.klt(self._prev, self._cur, p0, (prm.win, prm.win), prm.max_level, (3, prm.max_count, prm.epsilon),
                     prm.min_eig_threshold)

    def st_params(self, max_corners=1000, quality_level=0.03, min_distance=7, block_size=31, use_harris=False, harris_k=0.04):
        return SimpleNamespace(max_corners=max_corners, quality_level=quality_level, min_distance=min_distance,
                               block_size=block_size, use_harris=1 if use_harris else 0, harris_k=harris_k)

    def shi_tomasi(self, cur_pts=None, mask_radius=7, mask=None, params=None):
        prm = params or self.st_params()
        m = None
        if cur_pts is not None and len(cur_pts):
            m = np.full(self._cur.shape, 255, np.uint8) if mask is None else np.array(mask, np.uint8)
            for x, y in np.int32(np.asarray(cur_pts, np.float32).reshape(-1, 2)):
                o.circle_mask(m, (x, y), mask_radius, 0)
        elif mask is not None:
            m = np.asarray(mask, np.uint8)
        return o.good_features(self._cur, m, prm.max_corners, prm.quality_level, prm.min_distance, prm.block_size,
                               useHarrisDetector=bool(getattr(prm, "use_harris", 0)), k=getattr(prm, "harris_k", 0.04))

    def triangulate(self, P0, P1, uv0, uv1, K=None, H0=None, H1=None):
        X4 = o.triangulate(P0, P1, uv0, uv1)
        if K is None:
            return X4
        uv0 = np.asarray(uv0, np.float32).reshape(-1, 2).astype(np.float64)
        uv1 = np.asarray(uv1, np.float32).reshape(-1, 2).astype(np.float64)
        X = ((X4[:3] / X4[3]).T).astype(np.float64)        # float32 divide, as numpy does in the reference
        Xh = np.hstack([X, np.ones((len(X), 1))])
        depth1 = Xh @ H1[2]
        e = []
        for H, uv in ((H0, uv0), (H1, uv1)):
            p = Xh @ (K @ H[:3]).T
            e.append(np.linalg.norm(uv - p[:, :2] / p[:, 2:3], axis=1))
        return X4, depth1, (e[0] + e[1]) / 2

    def ba_params(self, max_iters=50, ftol=1e-3, xtol=1e-3, gtol=1e-8, lambda0=1e-4, huber_delta=1.0, lambda_min=1e-3):
        return SimpleNamespace(max_iters=max_iters, ftol=ftol, xtol=xtol, gtol=gtol, lambda0=lambda0, huber_delta=huber_delta,
                               lambda_min=lambda_min)

    def ba_adjust(self, K, poses, points, obs, params=None):
        prm = params or self.ba_params()
        r = bo.solve(K, poses, points, obs, max_iters=prm.max_iters, lam0=prm.lambda0, ftol=prm.ftol, xtol=prm.xtol,
                     gtol=prm.gtol, delta=prm.huber_delta, lam_min=prm.lambda_min)
        return r["poses"], r["points"], dict(cost0=r["cost0"], cost=r["cost"], lam=r["lam"], iters=r["iters"],
                                             accepted=r["accepted"], status=r["status"], n_obs=int(bo.valid_mask(obs).sum()))

    def bilateral(self, img, d=5, sigma_color=1.5, sigma_space=1.5):
        return o.bilateral(np.ascontiguousarray(img, np.uint8), d, sigma_color, sigma_space)

    def sift_detect_compute(self, img, mask=None, nfeatures=1000, max_out=None):
        import sift_oracle as so
        return so.detect_and_compute(img, nfeatures=nfeatures, mask=mask)

    def match_knn2(self, desc1, desc2):
        import match_oracle as mo
        return mo.knn2(desc1, desc2)

    def essential_ransac(self, K, pts1, pts2, threshold=1.0, prob=0.9999, max_iters=1000, seed=0, distance_thresh=50.0):
        import essential_oracle as eo
        E, R, t, inl, info = eo.essential_ransac(K, pts1, pts2, thr=threshold, prob=prob, max_iters=max_iters, seed=seed,
                                                 dist=distance_thresh, return_info=True)
        if E is None:
            return (np.full((3, 3), np.nan), np.full((3, 3), np.nan), np.full(3, np.nan), inl,
                    dict(status=-6, n_inliers=0, n_good=0, hypotheses=info["hyps"], best=-1))
        return E, R, t, inl, dict(status=0, n_inliers=len(inl), n_good=info["n_good"], hypotheses=info["hyps"], best=info["best"])

    def pnp_ransac(self, K, pts3d, pts2d, reproj_err=2.0, confidence=0.9999, max_iters=1000000, seed=0):
        import pnp_oracle as po
        r, t, inl, info = po.pnp_ransac(K, pts3d, pts2d, thr=reproj_err, conf=confidence, max_iters=max_iters, seed=seed,
                                        return_info=True)
        if r is None:
            return np.full(3, np.nan), np.full(3, np.nan), inl, dict(status=-6, n_inliers=0, hypotheses=info["hyps"], best=-1, cost=np.nan)
        return r, t, inl, dict(status=0, n_inliers=len(inl), hypotheses=info["hyps"], best=info["best"], cost=info["cost"])

    # -- read-backs used by the live-OpenCV comparison harness (tests/live_cv2.py) -----------------
    def pyramid_read(self, which, level, seq=0):
        img = (self._prev, self._cur)[which]
        lv = o.build_pyramid(img, self.win, self.max_level)[level]
        return lv, o.scharr(lv)
