"""GPU: 2D-2D bootstrap pose (pose part) -- five-point RANSAC + recoverPose against the numpy
oracle that defines the algorithm (same samples, same winner, same pose) and against ground truth with gross outliers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def two_view_scene(n, frac_out, seed, noise=0.3, K=None, rvec=(0.01, 0.03, -0.005), t=(0.1, -0.02, -0.9)):
    import pnp_oracle as po
    from vo_mi355x import synthetic as syn
    K = syn.KITTI_K if K is None else K
    rng = np.random.default_rng(seed)
    R = po.rodrigues(np.asarray(rvec, float)); t = np.asarray(t, float)
    X = np.stack([rng.uniform(-15, 15, n), rng.uniform(-3, 3, n), rng.uniform(8, 45, n)], 1)
    p1 = X @ K.T; p1 = p1[:, :2] / p1[:, 2:3]
    Xc = X @ R.T + t; p2 = Xc @ K.T; p2 = p2[:, :2] / p2[:, 2:3]
    p1 = p1 + rng.normal(0, noise, p1.shape); p2 = p2 + rng.normal(0, noise, p2.shape)
    out = rng.choice(n, int(frac_out * n), replace=False)
    p2[out] += rng.uniform(-60, 60, (len(out), 2)) + 10
    return K, p1.astype(np.float32), p2.astype(np.float32), R, t / np.linalg.norm(t), np.setdiff1d(np.arange(n), out)


def _same_E(a, b):
    return min(np.abs(a - b).max(), np.abs(a + b).max())


@pytest.mark.parametrize("n,frac,seed", [(600, 0.3, 3), (200, 0.5, 4), (40, 0.1, 5), (2000, 0.4, 6)])
def test_essential_matches_oracle_and_ground_truth(n, frac, seed):
    import essential_oracle as eo
    from vo_mi355x import VoContext
    K, p1, p2, R_gt, t_gt, true_inl = two_view_scene(n, frac, seed)
    with VoContext(64, 64, max_pts=64) as c:
        E, R, t, inl, st = c.essential_ransac(K, p1, p2, seed=7)
    E_o, R_o, t_o, inl_o, info = eo.essential_ransac(K, p1, p2, seed=7, return_info=True)
    assert st["status"] == 0
    # same search: number of samples, winning sample, model, consensus set (points within rounding of the threshold may flip)
    assert st["hypotheses"] == info["hyps"] and st["best"] == info["best"]
    assert _same_E(E, E_o) <= 1e-7
    assert len(np.setxor1d(inl, inl_o)) <= 2 and abs(st["n_inliers"] - info["count"]) <= 2
    assert np.abs(R - R_o).max() <= 1e-6 and np.abs(t - t_o).max() <= 1e-6
    assert abs(st["n_good"] - info["n_good"]) <= 2
    # ground truth: no gross outlier survives, the pose is noise-limited, rotation proper, |t| = 1
    assert len(np.setdiff1d(inl, true_inl)) <= 0.03 * n + 2
    assert len(np.intersect1d(inl, true_inl)) >= 0.8 * len(true_inl)
    assert abs(np.linalg.det(R) - 1) <= 1e-9 and abs(np.linalg.norm(t) - 1) <= 1e-9
    assert np.abs(R - R_gt).max() <= 1e-2 and np.abs(t - t_gt).max() <= (0.15 if n >= 200 else 0.4)


def test_essential_batch_nan_rows_sideways_motion_and_failure():
    from vo_mi355x import VoContext
    K, a1, a2, Ra, ta, _ = two_view_scene(400, 0.3, 11)
    _, b1, b2, Rb, tb, _ = two_view_scene(400, 0.1, 12, rvec=(0.0, -0.05, 0.01), t=(1.0, 0.05, 0.1))
    b1 = b1.copy(); b2 = b2.copy()
    b1[350:] = np.nan; b2[350:] = np.nan                      # ragged: the second sequence has only 350 correspondences
    with VoContext(64, 64, max_pts=64, batch=2) as c:
        E, R, t, inl, st = c.essential_ransac(np.stack([K, K]), np.stack([a1, b1]), np.stack([a2, b2]), seed=1)
    for b, (Rg, tg) in enumerate(((Ra, ta), (Rb, tb))):
        assert st[b]["status"] == 0 and np.abs(R[b] - Rg).max() <= 1e-2 and np.abs(t[b] - tg).max() <= 0.15
        Eb = E[b]
        assert abs(np.linalg.det(Eb)) <= 1e-9 and np.abs(2 * Eb @ Eb.T @ Eb - np.trace(Eb @ Eb.T) * Eb).max() <= 1e-8
    assert inl[1].max() < 350
    # pure noise: no model with >= 5 inliers at a tiny threshold -> status, NaN pose, empty mask, no crash
    rng = np.random.default_rng(0)
    with VoContext(64, 64, max_pts=64) as c:
        E, R, t, inl, st = c.essential_ransac(K, rng.uniform(0, 600, (30, 2)).astype(np.float32),
                                              rng.uniform(0, 600, (30, 2)).astype(np.float32), threshold=1e-9, max_iters=256)
    assert (st["status"] != 0 and len(inl) == 0 and np.isnan(R).all()) or st["n_inliers"] >= 5


def test_camera_pose_2d2d_dropin():
    from vo_mi355x import Extractor, Keypoint, VoContext
    K, p1, p2, R_gt, t_gt, true_inl = two_view_scene(500, 0.25, 21)
    mk = lambda p: [Keypoint(0, 1, p[i].reshape(2, 1), p[i].reshape(2, 1), np.zeros((1, 1)), [p[i].reshape(2, 1)]) for i in range(len(p))]
    with VoContext(64, 64, max_pts=64) as c:
        ext = Extractor(min_kp_dist=7, ctx=c)
        inliers, H = ext.camera_pose(K, mk(p1), mk(p2), corr='2D-2D')
        with pytest.raises(ValueError):
            ext.camera_pose(K, mk(p1), mk(p2), corr='3D-3D')
    assert isinstance(inliers, list) and H.shape == (4, 4) and np.allclose(H[3], [0, 0, 0, 1])
    assert np.abs(H[:3, :3] - R_gt).max() <= 1e-2 and np.abs(H[:3, 3] - t_gt).max() <= 0.15
    assert len(np.setdiff1d(inliers, true_inl)) <= 15
