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
    def __init__(self, cfg=None, min_kp_dist=10, ctx=None, device=0, max_pts=8192):
        self._cfg = cfg
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
        new_tracks = []
        if len(kp):
            p0 = np.float32([k.uv.T for k in kp]).reshape(-1, 2)
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
        landmarks_new, kp_new, landmarks_dead, kp_dead = [], [], [], []
        if not len(landmarks_kp):
            return landmarks_new, kp_new, landmarks_dead, kp_dead
        p0 = np.float32([k.uv for k in landmarks_kp]).reshape(-1, 2)
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
            return self._extract_sift(img, t, current_kp, describe)
        if detector != 'shi-tomasi':
            raise ValueError("detector must be 'shi-tomasi' or 'custom'")
        if describe:
            raise NotImplementedError("describe=True needs detector='custom' (SIFT): a Shi-Tomasi corner has no SIFT scale")
        self._context(img)
        self._ensure_cur(img)
        c = self._ctx
        sp = self._shitomasi_params
        prm = c.st_params(max_corners=sp["maxCorners"], quality_level=sp["qualityLevel"],
                          min_distance=sp["minDistance"], block_size=sp["blockSize"])
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
    def _dlt(self, K, H0, H1, keyp0, keyp1, stats):
        uv0 = np.array([kp.uv.T for kp in keyp0]).astype(np.float32).reshape((-1, 2))
        uv1 = np.array([kp.uv.T for kp in keyp1]).astype(np.float32).reshape((-1, 2))
        P_0 = (K @ H0[:3, :]).astype(np.float32)
        P_1 = (K @ H1[:3, :]).astype(np.float32)
        if self._ctx is None:
            raise RuntimeError("Extractor: no device context yet (track or extract a frame first, or pass ctx=)")
        if stats:
            X4, depth1, reproj = self._ctx.triangulate(P_0, P_1, uv0, uv1, K, H0, H1)
        else:
            X4, depth1, reproj = self._ctx.triangulate(P_0, P_1, uv0, uv1), None, None
        points_4D = X4.reshape((4, -1)).T
        points_3D = (points_4D / points_4D[:, 3].reshape((-1, 1)))[:, :3]
        return points_3D, depth1, reproj

    def triangulate(self, K, H0, H1, keyp0, keyp1, t):
        if not len(keyp0):
            return []
        points_3D, _, _ = self._dlt(K, H0, H1, keyp0, keyp1, False)
        return [Landmark(t, np.array(p).reshape((3, 1)), keyp1[i].des) for i, p in enumerate(points_3D.tolist())]

    def triangulate_nonlinear(self, K, H0, H1, keyp0, keyp1, t, max_err_reproj=1.0):
        """DLT + the filters of TriangulatorNL.refine (triangulate.py:82-146): camera-1 cheirality, then mean
        reprojection error < max_err_reproj.  The reference's scipy "refinement" never moves a point (its
        sparsity pattern is empty for the keypoint columns) and its final filter repeats the
        second one, so the result is exactly these two filters; like the reference, the surviving keypoints'
        `uv` come back as float64 (2,1) arrays."""
        if not len(keyp0):
            return [], [], []
        points_3D, depth1, reproj = self._dlt(K, H0, H1, keyp0, keyp1, True)
        keep = [i for i in range(len(keyp0)) if depth1[i] > 0 and reproj[i] < max_err_reproj]
        landmarks, k0, k1 = [], [], []
        for i in keep:
            landmarks.append(Landmark(t, np.array(points_3D[i].tolist()).reshape((3, 1)), keyp1[i].des))
            keyp0[i].uv = np.asarray(keyp0[i].uv, np.float64).reshape((2, 1)).copy()
            keyp1[i].uv = np.asarray(keyp1[i].uv, np.float64).reshape((2, 1)).copy()
            k0.append(keyp0[i]); k1.append(keyp1[i])
        return landmarks, k0, k1

    def triangulate_tracks(self, K, candidates_kp, trajectory, t_curr, refine=True, min_track_length=5,
                           min_bearing_angle=10, max_err_reproj=4.0):
        landmarks_new, landmarks_kp_new = [], []
        landmarks_kp_tmp = [kp for kp in candidates_kp if kp.t_total >= min_track_length]
        candidates_kp_new = [kp for kp in candidates_kp if kp.t_total < min_track_length]
        if len(landmarks_kp_tmp) > 0:
            H1 = trajectory[len(trajectory) - 1]
            t_first_groups = set([k.t_first for k in landmarks_kp_tmp])
            for t_first in t_first_groups:
                kp_1 = [kp for kp in landmarks_kp_tmp if kp.t_first == t_first]
                H0 = trajectory[t_first]
                kp_0 = deepcopy(kp_1)
                for kp in kp_0:
                    kp.uv = kp.uv_first
                l, kp_0, kp_1 = self.triangulate_nonlinear(K, H0, H1, kp_0, kp_1, t_curr, max_err_reproj=max_err_reproj)
                if len(l):
                    # the reference's "bearing angle" gate, quirks included (extractor.py:231-240)
                    Hrel = H1 @ np.linalg.inv(H0)
                    P_homo = np.concatenate([l[0].p, np.zeros((1, 1))], axis=0).reshape((4, 1))
                    a = np.linalg.norm(Hrel)
                    b = np.linalg.norm(H0 @ P_homo)
                    c = np.linalg.norm(H1 @ P_homo)
                    with np.errstate(invalid='ignore', divide='ignore'):
                        bearing_angle = np.rad2deg(np.arccos((b * b + c * c - a * a) / (2 * b * c)))
                    if (not np.isnan(bearing_angle)) and (bearing_angle > min_bearing_angle):
                        landmarks_new += l
                        landmarks_kp_new += kp_1
        return landmarks_new, landmarks_kp_new, candidates_kp_new

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
        """Match two lists of keypoints/landmarks based on their descriptors (reference extractor.py:147-154).
        Returns a list of matches. Each match has m.queryIdx for list_1, and m.trainIdx for list_2."""
        desc_dim = len(list_1[0].des)
        desc_1 = np.array([pt.des.reshape(1, desc_dim) for pt in list_1]).reshape((len(list_1), -1))
        desc_2 = np.array([pt.des.reshape(1, desc_dim) for pt in list_2]).reshape((len(list_2), -1))
        return self.match(desc_1, desc_2)

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
