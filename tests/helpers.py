"""Shared helpers for the tests (golden-vector decoding)."""
import numpy as np


def golden_tracks(g, prefix):
    lens = g[prefix + "hist_len"]
    off = np.concatenate([[0], np.cumsum(lens)])
    return [dict(p=g[prefix + "p"][i], t_latest=int(g[prefix + "t_latest"][i]), hist=g[prefix + "hist"][off[i]:off[i + 1]],
                 tag=float(g[prefix + "tag"][i])) for i in range(len(lens))]


def golden_ba_problem(g, rodrigues_mat_to_vec):
    """Rebuild the dense BA problem (poses [W,6], points [N,3], obs [W,N,2]) from a G1 golden file,
    following the reference's selection rules (bundle_adjuster.py:132-176)."""
    W, t_now, K = int(g["W"]), int(g["t_now"]), g["K"]
    act, dead = golden_tracks(g, "act_"), golden_tracks(g, "dead_")
    ref = list(act)
    for d in dead:
        t_earliest = d["t_latest"] - (len(d["hist"]) - 1)
        if (t_now - t_earliest) < W:
            ref.append(d)
    N = len(ref)
    obs = np.full((W, N, 2), np.nan)
    for j, r in enumerate(ref):
        L = len(r["hist"])
        for i in range(W):
            hi = (t_now - i) - r["t_latest"] + L - 1
            if 0 <= hi <= L - 1:
                obs[i, j] = r["hist"][hi]
    T = len(g["traj"])
    poses = np.zeros((W, 6))
    for i in range(W):
        if T - 1 - i < 0:
            break
        H = g["traj"][T - 1 - i]
        poses[i, :3] = rodrigues_mat_to_vec(H[:3, :3]).reshape(3)
        poses[i, 3:] = H[:3, 3]
    points = np.array([r["p"] for r in ref])
    tags = np.array([r["tag"] for r in ref])
    return K, poses, points, obs, tags
