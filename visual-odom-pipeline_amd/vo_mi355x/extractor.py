"""Drop-in for the reference's `Extractor` on the hot path (src/extractor/extractor.py).

Same method names, argument order, defaults, return arities and in-place mutation semantics as the reference
(extend_tracks :38-59, extend_landmarks :61-88, extract :90-132, triangulate_tracks :193-242,
triangulate_nonlinear :244-253, triangulate :255-277), but every OpenCV call is replaced by the HIP library
(VoContext).  Of the bootstrap (reference :114-172) the descriptor matching (match / match_lists) and both RANSAC poses
(camera_pose) and the SIFT detector / descriptor (extract(detector='custom'), cv2.SIFT_create(nfeatures=1000)) run on the
GPU too.
"""
from copy import deepcopy

import numpy as np

from .state import Keypoint, Landmark


class DMatch:
    """the fields of cv2.DMatch the reference reads (queryIdx, trainIdx; extractor.py:147-150, pipeline.py:56-63)"""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = int(queryIdx), int(trainIdx), int(imgIdx), float(distance)

    def __repr__(self):
        return "DMatch(queryIdx=%d, trainIdx=%d, distance=%.6g)" % (self.queryIdx, self.trainIdx, self.distance)


class Extractor:
    def __init__(self, cfg=None, min_kp_dist=10, ctx=None, device=0, max_pts=8192, lazy=None, lazy_backend=None):
        """lazy (default: on unless VO_LAZY=0): once a frame has come through the reference's call order (pipeline.py:98-156) the state moves
        into device tables and the lists this class hands out are views of them (vo_mi355x/lazy.py); lazy_backend: test hook
        (ctx, K, params, width, height) -> backend, default the GPU one.  max_pts: keypoints per call AND the capacity of those tables (<= 8192)."""
        from . import lazy as _lz
        self._cfg = cfg
        self._lazy_on = _lz.enabled() if lazy is None else bool(lazy)
        self._lazy_backend = lazy_backend
        self._lazy = None               # the live session, if any
        self._trace = []                # the plain-path calls of the current frame, in order (is this the reference's Pipeline.step?)
        self._seen = {}                 # parameters those calls came with (a session is created with them)
        _lz._EXTRACTORS.add(self)
        # parameters hard-coded by the reference (extractor.py:16-24)
        self._lk_params = dict(winSize=(31, 31), maxLevel=3, criteria=(3, 30, 0.03))
        self._shitomasi_params = dict(maxCorners=1000, qualityLevel=0.03, minDistance=min_kp_dist, blockSize=31)
        self._ctx = ctx
        self._device, self._max_pts = device, max_pts
        self._im_prev = None            # set by the caller, exactly like the reference (pipeline.py:36,103)
        self._dev_prev = None           # host copies of what the device frame store currently holds
        self._dev_cur = None

    # -- device frame store ---------------------------------------------------------------------
    def _context(self, img):
        if self._ctx is None:
            from .context import VoContext
            h, w = img.shape
            self._ctx = VoContext(w, h, max_pts=self._max_pts, device=self._device,
                                  max_level=self._lk_params["maxLevel"], win=self._lk_params["winSize"][0])
        return self._ctx

    def _push(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self._context(img).push_frame(img)
        self._dev_prev, self._dev_cur = self._dev_cur, img.copy()

    @staticmethod
    def _same(a, b):
        return a is not None and b is not None and a.shape == b.shape and np.array_equal(a, b)

    def _ensure_pair(self, im_prev, im_curr):
        """make (prev, cur) on the device equal (im_prev, im_curr) with as few uploads as possible"""
        if self._same(self._dev_prev, im_prev) and self._same(self._dev_cur, im_curr):
            return
        if not self._same(self._dev_cur, im_prev):
            self._push(im_prev)
        self._push(im_curr)

    def _ensure_cur(self, img):
        if not self._same(self._dev_cur, img):
            self._push(img)

    def _klt(self, p0):
        c = self._ctx
        prm = c.klt_params(win=self._lk_params["winSize"][0], max_level=self._lk_params["maxLevel"],
                           max_count=self._lk_params["criteria"][1], epsilon=self._lk_params["criteria"][2])
        return c.klt_track(p0, prm)

    def _track(self, im_curr, p0, max_bidir_error):
        """p1 and the reference's 'bidirectional' flag.  The reference's second pass tracks FORWARD again from
        p1 (extractor.py:45,66); with an infinite threshold its result cannot change `good` (NaN aside), so it
        is skipped then."""
        self._ensure_pair(self._im_prev, im_curr)
        p1, _st, _err = self._klt(p0)
        if np.isinf(max_bidir_error):
            good = ~np.isnan(p1).any(axis=1)
        else:
            p0r, _st, _err = self._klt(p1)
            good = np.abs(p0 - p0r).max(-1) < max_bidir_error
        return p1, good

    # -- the lazy boundary (lazy.py) ---------------------------------------------------------------
    def _session(self):
        s = self._lazy
        if s is not None and not s.alive:
            s = self._lazy = None
        return s

    def _plain(self, name, img=None, **seen):
        """a call is about to take the plain path: a live session ends here (its proxies become plain objects first), and the call is noted"""
        s = self._session()
        if s is not None:
            s.desync("%s took the plain path" % name)
            self._lazy = None
        if name == "extend_tracks":
            self._trace = []
        self._trace.append((name, None if img is None else id(img)))
        self._seen.update(seen)

    def _frame_is_reference_order(self):
        """extend_tracks, extend_landmarks, camera_pose('3D-2D'), triangulate_tracks -- each once, in this order, on one image"""
        names = [n for n, _ in self._trace]
        imgs = {i for n, i in self._trace[:2]}
        return names == ["extend_tracks", "extend_landmarks", "camera_pose", "triangulate_tracks"] and len(imgs) == 1

    def _fast_pushed(self, s):
        """the session pushed a frame into the context's store: keep this class's record of the store's contents"""
        if self._dev_cur is not s.cur_img:
            self._dev_prev, self._dev_cur = self._dev_cur, s.cur_img

    # -- KLT ------------------------------------------------------------------------------------
    def _survivors(self, im_curr, p1, good):
        """keep mask of the reference's rule (0 <= x <= W and 0 <= y <= H, ends included, and the 'bidirectional' flag)
        plus two independent (n, 2, 1) float32 blocks whose rows become k.uv and the new history entry -- one allocation
        each instead of two small arrays per keypoint; every keypoint still owns its own values (separate rows, and the
        history entry is not the same memory as uv, exactly like the reference's two np.array(...) calls)."""
        h, w = im_curr.shape[0], im_curr.shape[1]
        keep = good & (p1[:, 0] >= 0) & (p1[:, 0] <= w) & (p1[:, 1] >= 0) & (p1[:, 1] <= h)
        uv = p1.reshape(-1, 2, 1).copy()
        return keep, uv, uv.copy()

    def extend_tracks(self, im_curr, kp, max_bidir_error=30):
        s = self._session()
        if s is not None:
            r = s.extend_tracks(self._im_prev, im_curr, kp, max_bidir_error)
            if r is not NotImplemented:
                self._fast_pushed(s)
                return r
        self._plain("extend_tracks", im_curr, max_bidir_error=max_bidir_error)
        new_tracks = []
        if len(kp):
            p0 = self._uv_block(kp)
            p1, good = self._track(im_curr, p0, max_bidir_error)
            keep, uv, hist = self._survivors(im_curr, p1, good)
            for i in np.nonzero(keep)[0]:
                k = kp[i]
                k.uv = uv[i]
                k.t_total += 1
                k.uv_history.append(hist[i])
                new_tracks.append(k)
        return new_tracks

    def extend_landmarks(self, im_curr, landmarks, landmarks_kp, max_bidir_error=30):
        s = self._session()
        if s is not None:
            r = s.extend_landmarks(self._im_prev, im_curr, landmarks, landmarks_kp, max_bidir_error)
            if r is not NotImplemented:
                self._fast_pushed(s)
                return r
        self._plain("extend_landmarks", im_curr)
        landmarks_new, kp_new, landmarks_dead, kp_dead = [], [], [], []
        if not len(landmarks_kp):
            return landmarks_new, kp_new, landmarks_dead, kp_dead
        p0 = self._uv_block(landmarks_kp)
        p1, good = self._track(im_curr, p0, max_bidir_error)
        keep, uv, hist = self._survivors(im_curr, p1, good)
        # the reference converts p1 with .tolist() here (python floats): uv / history entries are float64 in this method
        uv, hist = uv.astype(np.float64), hist.astype(np.float64)
        for i in range(len(landmarks)):
            l, k = landmarks[i], landmarks_kp[i]
            if not keep[i]:
                landmarks_dead.append(l)
                kp_dead.append(k)
                continue
            k.uv = uv[i]
            k.t_total += 1
            k.uv_history.append(hist[i])
            l.t_latest += 1
            kp_new.append(deepcopy(k))
            landmarks_new.append(l)
        return landmarks_new, kp_new, landmarks_dead, kp_dead

    # -- re-detection ---------------------------------------------------------------------------
    def extract(self, img, t, current_kp=[], detector='custom', mask_radius=5, describe=False):
        if detector == 'custom':
            self._plain("extract_sift")
            return self._extract_sift(img, t, current_kp, describe)
        if detector != 'shi-tomasi':
            raise ValueError("detector must be 'shi-tomasi' or 'custom'")
        if describe:
            raise NotImplementedError("describe=True needs detector='custom' (SIFT): a Shi-Tomasi corner has no SIFT scale")
        s = self._session()
        if s is not None:
            r = s.extract(img, t, current_kp, mask_radius)
            if r is not NotImplemented:
                return r
        self._plain("extract", mask_radius=mask_radius)
        self._context(img)
        self._ensure_cur(img)
        c = self._ctx
        sp = self._shitomasi_params
        # the dict is what the reference splats into cv2.goodFeaturesToTrack (extractor.py:111); its optional keys are honoured too
        prm = c.st_params(max_corners=sp["maxCorners"], quality_level=sp["qualityLevel"],
                          min_distance=sp["minDistance"], block_size=sp["blockSize"],
                          use_harris=bool(sp.get("useHarrisDetector", False)), harris_k=sp.get("k", 0.04))
        # np.int32(kp.uv) truncation happens on the device; float32 carries every pixel coordinate exactly
        cur = (np.array([k.uv for k in current_kp], dtype=np.float64).reshape(-1, 2).astype(np.float32) if len(current_kp)
               else np.zeros((0, 2), np.float32))
        kp = c.shi_tomasi(cur if len(cur) else None, mask_radius=mask_radius, params=prm)
        if kp.shape[0] == 0:
            return []
        # like the reference (extractor.py:127-131) uv_first, uv and the first history entry of a new keypoint are three
        # VIEWS of the same row of the detector output (kp[i, :].reshape((2, 1)) is a view); des a view of the zero column
        kp3 = kp.reshape(-1, 2, 1)
        desc = np.zeros((kp.shape[0], 1, 1))
        return [Keypoint(t_first=t, t_total=1, uv_first=kp3[i], uv=kp3[i], des=desc[i], uv_history=[kp3[i]])
                for i in range(len(kp))]

    # -- triangulation --------------------------------------------------------------------------
    @staticmethod
    def _uv_block(kps):
        """(n, 2) float32 pixel coordinates of a keypoint list: one float64 gather, one rounding to float32 (the reference
        rounds its stacked (n, 1, 2) array the same way before cv2.triangulatePoints, extractor.py:264-265)"""
        return np.asarray([k.uv for k in kps], dtype=np.float64).reshape(-1, 2).astype(np.float32)

    def _two_view(self, K, H0, H1, keyp0, keyp1, want_stats):
        """k_dlt for one view pair -> (points [n, 3] float32, camera-1 depth, mean reprojection error).  The projection matrices
        are rounded to float32 before the solve, as the reference hands them to OpenCV (extractor.py:268-269)."""
        if self._ctx is None:
            raise RuntimeError("Extractor: no device context yet (track or extract a frame first, or pass ctx=)")
        cams = [np.float32(np.asarray(K) @ np.asarray(H)[:3]) for H in (H0, H1)]
        pix = [self._uv_block(keyp0), self._uv_block(keyp1)]
        if want_stats:
            X4, depth1, reproj = self._ctx.triangulate(cams[0], cams[1], pix[0], pix[1], K, H0, H1)
        else:
            X4, depth1, reproj = self._ctx.triangulate(cams[0], cams[1], pix[0], pix[1]), None, None
        X4 = X4.reshape(4, -1)
        return (X4[:3] / X4[3]).T, depth1, reproj            # float32 division, point by point

    @staticmethod
    def _as_landmarks(points, rows, kps, t):
        """Landmark(t, p (3, 1) float64 holding the float32 values, des of the view-1 keypoint)"""
        wide = np.asarray(points, np.float64)
        return [Landmark(t, wide[i].reshape(3, 1).copy(), kps[i].des) for i in rows]

    def triangulate(self, K, H0, H1, keyp0, keyp1, t):
        if not len(keyp0):
            return []
        pts, _, _ = self._two_view(K, H0, H1, keyp0, keyp1, False)
        return self._as_landmarks(pts, range(len(pts)), keyp1, t)

    def triangulate_nonlinear(self, K, H0, H1, keyp0, keyp1, t, max_err_reproj=1.0):
        """DLT + the filters of TriangulatorNL.refine (triangulate.py:82-146): camera-1 cheirality, then mean
        reprojection error < max_err_reproj.  The reference's scipy "refinement" never moves a point (its
        sparsity pattern is empty for the keypoint columns) and its final filter repeats the
        second one, so the result is exactly these two filters; like the reference, the surviving keypoints'
        `uv` come back as float64 (2,1) arrays."""
        if not len(keyp0):
            return [], [], []
        pts, depth1, reproj = self._two_view(K, H0, H1, keyp0, keyp1, True)
        rows = np.nonzero((np.asarray(depth1) > 0) & (np.asarray(reproj) < max_err_reproj))[0]
        for view in (keyp0, keyp1):
            for i in rows:
                view[i].uv = np.array(view[i].uv, dtype=np.float64).reshape(2, 1)
        return self._as_landmarks(pts, rows, keyp1, t), [keyp0[i] for i in rows], [keyp1[i] for i in rows]

    @staticmethod
    def _group_gate(H_birth, H_now, p_first, min_angle_deg):
        """The reference's per-group "bearing angle" test (extractor.py:231-240), which is not a bearing angle: the triangle
        side `a` is the FROBENIUS NORM of the 4x4 relative transform (>= 2), the other two sides are the lengths of the
        group's FIRST landmark rotated into either camera as a direction (w = 0, so both equal |p|).  Law of cosines,
        degrees; NaN (|cos| > 1) rejects.  In effect: accept the whole group iff |p_first| is large enough against a."""
        ray = np.zeros(4)
        ray[:3] = np.asarray(p_first, np.float64).ravel()
        a = np.linalg.norm(np.asarray(H_now) @ np.linalg.inv(H_birth))
        b, c = np.linalg.norm(np.asarray(H_birth) @ ray), np.linalg.norm(np.asarray(H_now) @ ray)
        with np.errstate(invalid='ignore', divide='ignore'):
            theta = np.degrees(np.arccos((b * b + c * c - a * a) / (2 * b * c)))
        return bool(theta > min_angle_deg)                   # False for NaN

    def triangulate_tracks(self, K, candidates_kp, trajectory, t_curr, refine=True, min_track_length=5,
                           min_bearing_angle=10, max_err_reproj=4.0):
        """Candidates that reached `min_track_length` leave the candidate list (triangulated or not); they are triangulated
        per birth frame between their first and their newest observation and a group is kept iff its gate passes
        (reference extractor.py:193-242).  -> (new landmarks, their keypoints, remaining candidates)"""
        s = self._session()
        if s is not None and refine:
            r = s.triangulate_tracks(K, candidates_kp, trajectory, t_curr, min_track_length, min_bearing_angle, max_err_reproj)
            if r is not NotImplemented:
                return r
        self._plain("triangulate_tracks", min_track_length=min_track_length, min_bearing_angle=min_bearing_angle, tri_max_err=max_err_reproj)
        ripe = [k for k in candidates_kp if k.t_total >= min_track_length]
        waiting = [k for k in candidates_kp if k.t_total < min_track_length]
        out_l, out_k = [], []
        if not ripe:
            return out_l, out_k, waiting
        by_birth = {}
        for k in ripe:
            by_birth.setdefault(k.t_first, []).append(k)
        H_now = trajectory[len(trajectory) - 1]
        # the reference walks a SET of birth frames built from the ripe list; building it the same way gives the same
        # iteration order, which is the order of the returned lists
        for born in set(k.t_first for k in ripe):
            newest = by_birth[born]
            # view-0 stand-ins: only `uv` (:= the first observation) is read and rewritten downstream
            oldest = [Keypoint(k.t_first, k.t_total, k.uv_first, np.array(k.uv_first), k.des, []) for k in newest]
            H_birth = trajectory[born]
            lms, _, kept = self.triangulate_nonlinear(K, H_birth, H_now, oldest, newest, t_curr, max_err_reproj=max_err_reproj)
            if lms and self._group_gate(H_birth, H_now, lms[0].p, min_bearing_angle):
                out_l.extend(lms)
                out_k.extend(kept)
        return out_l, out_k, waiting

    def _extract_sift(self, img, t, current_kp, describe):
        """detector='custom' (reference extractor.py:114-131): cv2.SIFT_create(nfeatures=1000).detect(img, mask) and,
        with describe=True, .compute(img, kps) -> Keypoints whose `des` is the (128, 1) float32 descriptor column
        (a (1, 1) zero column without describe).  The bootstrap calls it with no current keypoints (pipeline.py:48-49)."""
        if len(current_kp):
            raise NotImplementedError("SIFT detection with a keypoint mask is not used by the reference pipeline")
        self._context(img)
        kps, desc = self._ctx.sift_detect_compute(img, nfeatures=1000)
        kp = np.ascontiguousarray(kps[:, :2], np.float32)          # cv2.KeyPoint_convert: (n, 2) float32
        if not describe:
            desc = np.zeros((kp.shape[0], 1))
        return [Keypoint(t_first=t, t_total=1, uv_first=kp[i, :].reshape((2, 1)), uv=kp[i, :].reshape((2, 1)),
                         des=desc[i, :].reshape((-1, 1)), uv_history=[kp[i, :].reshape((2, 1))]) for i in range(len(kp))]

    # -- bootstrap (reference extractor.py:134-191) -------------------------------------------------
    _feature_method = 'sift'
    _sift_ratio = 0.80                                   # reference extractor.py:29

    def match(self, desc_1, desc_2):
        """cv2.BFMatcher().knnMatch(desc_1, desc_2, k=2) + Lowe's ratio test (reference extractor.py:134-145): the
        list of DMatch whose nearest neighbour is closer than _sift_ratio x the second nearest."""
        if self._ctx is None:
            raise RuntimeError("match needs the device context: track a frame first (or pass ctx=)")
        desc_1 = np.ascontiguousarray(desc_1, np.float32); desc_2 = np.ascontiguousarray(desc_2, np.float32)
        if len(desc_2) < 2:
            raise ValueError("not enough values to unpack (expected 2, got %d)" % len(desc_2))   # the reference's `for m, n in matches`
        idx, dist = self._ctx.match_knn2(desc_1, desc_2)
        good = []
        for q in range(len(desc_1)):
            m_d, n_d = float(dist[q, 0]), float(dist[q, 1])
            if m_d < self._sift_ratio * n_d:
                good.append(DMatch(q, idx[q, 0], m_d))
        return good

    def match_lists(self, list_1, list_2):
        """ratio-test matches between two lists of keypoints / landmarks by their `des` columns (reference
        extractor.py:147-154): m.queryIdx indexes list_1, m.trainIdx list_2"""
        rows = [np.stack([np.ravel(o.des) for o in lst]) for lst in (list_1, list_2)]
        return self.match(rows[0], rows[1])

    def match_list(self, kp_1, desc_1, keypoints):
        # the reference's match_list (extractor.py:156-160) calls self.match with four arguments and always raises
        raise TypeError("match() takes 3 positional arguments but 5 were given")

    def camera_pose(self, K, list_1, list_2, corr='2D-2D', max_err_reproj=4.0):
        """Pose from correspondences -> (inlier indices as a list, H 4x4).
        corr='3D-2D' (reference extractor.py:174-191): cv2.solvePnPRansac(..., reprojectionError=max_err_reproj,
        iterationsCount=1e6, confidence=0.9999) + Rodrigues; H maps world -> camera.
        corr='2D-2D' (reference extractor.py:162-172, the bootstrap): cv2.findEssentialMat(prob=0.9999, RANSAC,
        threshold=1.0) + cv2.recoverPose on its inliers; H maps view-1 -> view-2 coordinates, |t| = 1.
        RANSAC draws differ from OpenCV's (statistical parity, see include/vo_mi355x.h)."""
        if corr == '3D-2D':
            s = self._session()
            if s is not None:
                r = s.camera_pose(K, list_1, list_2, max_err_reproj)
                if r is not NotImplemented:
                    return r
        self._plain("camera_pose" if corr == '3D-2D' else "camera_pose_2d2d", pose_max_err=max_err_reproj)
        if self._ctx is None:
            raise RuntimeError("camera_pose needs the device context: track a frame first (or pass ctx=)")
        if corr == '2D-2D':
            kp_1_pts = np.array([kp.uv.T for kp in list_1]).astype(np.float32).reshape(-1, 2)
            kp_2_pts = np.array([kp.uv.T for kp in list_2]).astype(np.float32).reshape(-1, 2)
            _E, R, t, inliers, st = self._ctx.essential_ransac(np.asarray(K, np.float64), kp_1_pts, kp_2_pts, threshold=1.0,
                                                               prob=0.9999)
            if st["status"] != 0:
                raise RuntimeError("findEssentialMat found no model")     # cv2 returns E = None -> the reference crashes in recoverPose
            H = np.eye(4)
            H[:3, :3] = R
            H[:3, 3] = t.reshape((3,))
            return inliers.reshape((-1,)).tolist(), H
        if corr != '3D-2D':
            raise ValueError("corr must be '2D-2D' or '3D-2D'")
        from .so3 import rodrigues_vec_to_mat
        pts3d = np.array([kp.p.T for kp in list_1]).astype(np.float32).reshape(-1, 3)
        pts2d = np.array([kp.uv.T for kp in list_2]).astype(np.float32).reshape(-1, 2)
        rvec, t, inliers, st = self._ctx.pnp_ransac(np.asarray(K, np.float64), pts3d, pts2d, reproj_err=max_err_reproj,
                                                    confidence=0.9999, max_iters=1000000)
        if st["status"] != 0:
            raise RuntimeError("solvePnPRansac found no pose")          # cv2 returns retval False and inliers None -> the reference crashes too
        H = np.eye(4)
        H[:3, :3] = rodrigues_vec_to_mat(rvec)
        H[:3, 3] = t.reshape((3,))
        return inliers.reshape((-1,)).tolist(), H
