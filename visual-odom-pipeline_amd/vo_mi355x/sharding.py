"""Host side of the landmark-sharded bundle adjustment (BASELINE config 5).

The reference adjusts one window in one process (`BundleAdjuster.adjust`, /root/reference/src/bundle_adjuster/
bundle_adjuster.py:127-215).  Here the N landmarks of the window are dealt round-robin to S shards (S = ranks x
shards-per-GPU): shard s owns landmarks j with j % S == s and ALL their observations; the W poses and K are
replicated.  Shards are padded to the same length with unobserved landmarks at the origin (they contribute exactly 0
to every sum of the solver, including the norms of the termination tests).

Pure numpy index bookkeeping -- the arithmetic of the sharded solve is in csrc/vo_ba.hip (k_ba_xsum / k_ba_xstat +
RCCL all-reduce, csrc/vo_comm.hip).
"""
import numpy as np


def shard_size(n_points, n_shards):
    return (n_points + n_shards - 1) // n_shards


def shard_problem(K, poses, points, obs, n_shards, first=0, count=None):
    """-> (K_s, poses_s, points_s, obs_s) stacked for shards first .. first + count - 1 of n_shards
    (shapes (count,3,3), (count,W,6), (count,Nloc,3), (count,W,Nloc,2)).  A rank of an R-rank job with V shards per GPU
    calls it with n_shards = R * V, first = rank * V, count = V."""
    points = np.asarray(points, np.float64)
    obs = np.asarray(obs, np.float64)
    poses = np.asarray(poses, np.float64)
    K = np.asarray(K, np.float64)
    W, N = obs.shape[0], obs.shape[1]
    count = n_shards - first if count is None else count
    nloc = shard_size(N, n_shards)
    pts_s = np.zeros((count, nloc, 3))
    obs_s = np.full((count, W, nloc, 2), np.nan)
    for q in range(count):
        idx = np.arange(first + q, N, n_shards)
        pts_s[q, :len(idx)] = points[idx]
        obs_s[q, :, :len(idx)] = obs[:, idx]
    return (np.broadcast_to(K, (count, 3, 3)).copy(), np.broadcast_to(poses, (count,) + poses.shape).copy(), pts_s, obs_s)


def unshard_points(points_all, n_points):
    """Inverse of the dealing: points_all (..., Nloc, 3) with the shard axes flattened in shard order -> (N, 3)."""
    pa = np.asarray(points_all)
    nloc = pa.shape[-2]
    pa = pa.reshape(-1, nloc, 3)
    S = pa.shape[0]
    out = np.zeros((n_points, 3))
    for s in range(S):
        idx = np.arange(s, n_points, S)
        out[idx] = pa[s, :len(idx)]
    return out
