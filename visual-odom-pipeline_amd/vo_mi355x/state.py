"""Boundary data types of the hot path -- same names and fields as the reference's
src/state/{keypoint,landmark,trajectory,state}.py (keypoint.py:4-21, landmark.py:4-13,
trajectory.py:3-31, state.py:4-10), so objects of either package can be passed to
the drop-in Extractor / BundleAdjuster (they only rely on attribute names)."""
from dataclasses import dataclass

import numpy as np


@dataclass
class Keypoint:
    """uv_first / uv: (2,1) arrays; uv_history: list of (2,1) arrays, one per frame since detection"""
    t_first: int
    t_total: int
    uv_first: np.ndarray
    uv: np.ndarray
    des: np.ndarray
    uv_history: list


@dataclass
class Landmark:
    """t_latest: frame of the last successful track; p: (3,1) world point"""
    t_latest: int
    p: np.ndarray
    des: np.ndarray


class Trajectory:
    """dict t_step -> 4x4 world->camera pose H (x_cam = H x_world)"""

    def __init__(self, poses=None):
        self._poses = {} if poses is None else poses

    def __len__(self):
        return len(self._poses)

    def __getitem__(self, key):
        return self._poses[key]

    def append(self, t, pose):
        self._poses[t] = pose

    def remove(self, i):
        if i not in self._poses.keys():
            raise ValueError('Out of bounds')
        del self._poses[i]


class State:
    def __init__(self, landmarks, landmarks_kp, candidates_kp, trajectory):
        self._landmarks = landmarks
        self._landmarks_kp = landmarks_kp
        self._candidates_kp = candidates_kp
        self._trajectory = trajectory
