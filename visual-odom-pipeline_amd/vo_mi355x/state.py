"""Boundary data types of the hot path -- same names and fields as the reference's
src/state/{keypoint,landmark,trajectory,state}.py (keypoint.py:4-21, landmark.py:4-13,
trajectory.py:3-31, state.py:4-10), so objects of either package can be passed to
the drop-in Extractor / BundleAdjuster (they only rely on attribute names)."""
from copy import deepcopy
from dataclasses import dataclass

import numpy as np


def _dc(a, memo):
    """copy.deepcopy of one field with the memo protocol (objects that were the same stay the same: a new keypoint's
    uv_first, uv and first history entry are views of one array), arrays through ndarray.copy -- the generic
    deepcopy machinery costs ~10x more per object and the reference's loop deep-copies every landmark every frame
    (extractor.py:86, pipeline.py:101-102)"""
    r = memo.get(id(a))
    if r is None:
        r = a.copy() if type(a) is np.ndarray else deepcopy(a, memo)
        memo[id(a)] = r
    return r


@dataclass
class Keypoint:
    """uv_first / uv: (2,1) arrays; uv_history: list of (2,1) arrays, one per frame since detection"""
    t_first: int
    t_total: int
    uv_first: np.ndarray
    uv: np.ndarray
    des: np.ndarray
    uv_history: list

    def __deepcopy__(self, memo):
        new = Keypoint.__new__(Keypoint)
        memo[id(self)] = new
        new.t_first, new.t_total = self.t_first, self.t_total
        new.uv_first, new.uv, new.des = _dc(self.uv_first, memo), _dc(self.uv, memo), _dc(self.des, memo)
        h = self.uv_history
        if type(h) is list:
            hist = memo.get(id(h))
            if hist is None:
                hist = []
                memo[id(h)] = hist
                for e in h:
                    hist.append(_dc(e, memo))
            new.uv_history = hist
        else:
            new.uv_history = deepcopy(h, memo)
        return new


@dataclass
class Landmark:
    """t_latest: frame of the last successful track; p: (3,1) world point"""
    t_latest: int
    p: np.ndarray
    des: np.ndarray

    def __deepcopy__(self, memo):
        new = Landmark.__new__(Landmark)
        memo[id(self)] = new
        new.t_latest = self.t_latest
        new.p, new.des = _dc(self.p, memo), _dc(self.des, memo)
        return new


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
