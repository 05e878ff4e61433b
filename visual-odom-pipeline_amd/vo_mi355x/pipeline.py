"""Drop-in for the reference's `Pipeline` (src/pipeline/pipeline.py:12-181): the same configuration constants, bootstrap
(_get_init_state :42-90) and per-frame step (:92-167) over the GPU-backed Extractor / BundleAdjuster.  Host logic only --
list bookkeeping in the reference's order; every numerical call goes to the device.  The visualiser (src/visu) is not
rebuilt: pass any object with update(im, state, landmarks_dead) / render() as `visu`, or leave it out (headless).
"""
from copy import deepcopy
import logging

import numpy as np

from .bundle_adjuster import BundleAdjuster
from .extractor import Extractor
from .state import State, Trajectory


class Pipeline:
    def __init__(self, loader, headless=True, visu=None, ctx=None, device=0):
        self._loader = loader
        self._K = loader.getCamera()

        # Pipeline Configuration Params (pipeline.py:17-25)
        self._ba = True
        self._ba_window_size = 4
        self._ba_frequency = 1
        self._min_kp_dist = 7
        self._max_bidir_error = np.inf
        self._max_reprojection_error = 2.0
        self._min_landmark_angle = 0.5
        self._kp_method = 'shi-tomasi'

        self._extractor = Extractor(min_kp_dist=self._min_kp_dist, ctx=ctx, device=device)
        self._bundle_adjuster = BundleAdjuster(verbosity=0, window_size=self._ba_window_size, method='trf', xtol=1e-3,
                                               ftol=1e-3, ctx=ctx, device=device)
        self._visu = None if headless and visu is None else visu
        self._t_step = 1

        self._landmarks_dead, self._landmarks_kp_dead = [], []
        self._state, self._t_loader, self._tra_gt = self._get_init_state()
        if ctx is None:                                    # one device context for tracking and bundle adjustment
            self._bundle_adjuster._ctx = self._extractor._ctx

        self._extractor._im_prev = self._loader.getImage(self._t_loader)
        self._show(self._extractor._im_prev)

    def _show(self, im):
        if self._visu is not None:
            self._visu.update(im, self._state, self._landmarks_dead)
            self._visu.render()

    def _get_init_state(self):
        t0, t1 = self._loader.getInit()
        im0, H0_gt = self._loader.getFrame(t0)
        im1, H1_gt = self._loader.getFrame(t1)

        # Feature detection and matching
        kp0 = self._extractor.extract(im0, 0, detector='custom', describe=True)
        kp1 = self._extractor.extract(im1, 1, detector='custom', describe=True)
        matches = self._extractor.match_lists(kp0, kp1)

        # Split keypoints into matched and un-matched
        kp0_m, kp1_m = [], []
        i1_nm = set(range(len(kp1)))
        for match in matches:
            kp0_m.append(deepcopy(kp0[match.queryIdx]))
            kp1_m.append(deepcopy(kp1[match.trainIdx]))
            i1_nm.discard(match.trainIdx)
        kp1_nm = [kp1[i] for i in sorted(i1_nm)]

        # Relative pose (bootstrapped baseline length = 1)
        H0 = np.eye(4)
        inliers, H1 = self._extractor.camera_pose(self._K, kp0_m, kp1_m, corr='2D-2D')
        kp0_m = [kp0_m[i] for i in inliers]
        kp1_m = [kp1_m[i] for i in inliers]

        # Triangulate inliers to create landmarks
        landmarks, kp0_m, kp1_m = self._extractor.triangulate_nonlinear(self._K, H0, H1, kp0_m, kp1_m, self._t_step,
                                                                        max_err_reproj=self._max_reprojection_error)
        trajectory, trajectory_gt = Trajectory({}), Trajectory({})
        trajectory.append(0, H0)
        trajectory.append(1, H1)
        trajectory_gt.append(0, H0_gt)
        trajectory_gt.append(1, H1_gt)
        return State(landmarks, kp1_m, kp1_nm, trajectory), t1, trajectory_gt

    def step(self):
        self._t_step += 1
        self._t_loader += 1
        im, H_gt = self._loader.getFrame(self._t_loader)
        ex, st = self._extractor, self._state

        # Extend track lengths (remove points that failed to track into the current frame)
        st._candidates_kp = ex.extend_tracks(im, st._candidates_kp, max_bidir_error=self._max_bidir_error)
        st._landmarks, st._landmarks_kp, landmarks_dead, landmarks_kp_dead = ex.extend_landmarks(
            im, st._landmarks, st._landmarks_kp, max_bidir_error=self._max_bidir_error)
        self._landmarks_dead += deepcopy(landmarks_dead)
        self._landmarks_kp_dead += deepcopy(landmarks_kp_dead)
        ex._im_prev = im.copy()

        # Localize with tracked keypoints
        inliers, H1 = ex.camera_pose(self._K, st._landmarks, st._landmarks_kp, corr='3D-2D',
                                     max_err_reproj=self._max_reprojection_error)

        # Remove bad landmarks (outliers) and their keypoints
        inl = set(inliers)
        landmarks, landmarks_kp = [], []
        for i in range(len(st._landmarks)):
            if i in inl:
                landmarks.append(st._landmarks[i])
                landmarks_kp.append(st._landmarks_kp[i])
            else:
                self._landmarks_dead.append(deepcopy(st._landmarks[i]))
                self._landmarks_kp_dead.append(deepcopy(st._landmarks_kp[i]))
        st._landmarks = landmarks
        st._landmarks_kp = landmarks_kp

        st._trajectory.append(self._t_step, H1)

        # Triangulate passable candidates
        landmarks_new, landmarks_kp_new, st._candidates_kp = ex.triangulate_tracks(
            self._K, st._candidates_kp, st._trajectory, t_curr=self._t_step, min_track_length=3,
            min_bearing_angle=self._min_landmark_angle, max_err_reproj=self._max_reprojection_error, refine=True)
        st._landmarks_kp += landmarks_kp_new
        st._landmarks += landmarks_new

        # Bundle Adjustment
        if self._ba and (self._t_step % self._ba_frequency == 0):
            self._state, self._landmarks_dead, self._landmarks_kp_dead = self._bundle_adjuster.adjust(
                self._state, self._landmarks_dead, self._landmarks_kp_dead, self._K, self._t_step)
            st = self._state

        # Detect new features and initialize new tracks
        st._candidates_kp += ex.extract(im, self._t_step, st._landmarks_kp + st._candidates_kp, detector=self._kp_method,
                                        mask_radius=self._min_kp_dist, describe=False)
        self._show(im)

    def full_run(self):
        # (the reference iterates one step too far and ends in the loader's AssertionError; this stops at the last frame)
        logging.info('Started Full run at timestep ' + str(self._t_loader))
        for _ in range(self._t_loader, len(self._loader) - 1):
            logging.info('Pipeline run ' + str(self._t_loader) + '/' + str(len(self._loader)))
            self.step()
