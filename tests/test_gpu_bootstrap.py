"""GPU: the bootstrap of Pipeline._get_init_state (reference pipeline.py:42-90) from given keypoints + descriptors
(SIFT itself is not rebuilt): match_lists -> camera_pose('2D-2D') -> triangulate_nonlinear, every numerical step on
the device through the drop-in Extractor; compared with the same glue over the CPU oracle context and with ground truth."""
import copy

import numpy as np
import pytest

from test_adapters import _gpu_ctx, _oracle_ctx, _sift_like
from test_gpu_essential import two_view_scene


def _bootstrap(make_ctx, seed=2, n=400, n_extra=80):
    from vo_mi355x import Extractor, Keypoint
    K, p1, p2, R_gt, t_gt, true_inl = two_view_scene(n, 0.0, seed, noise=0.2)
    rng = np.random.default_rng(seed)
    des = _sift_like(rng, n)
    des1 = des + rng.integers(-2, 3, des.shape).astype(np.float32)
    # view 1 additionally holds unmatched detections; 10 % of the matched descriptors are swapped -> wrong matches
    wrong = rng.choice(n, n // 10, replace=False)
    des1[wrong] = des1[np.roll(wrong, 1)]
    extra_uv = rng.uniform(20, 1200, (n_extra, 2)).astype(np.float32) * np.float32([1.0, 0.3])
    extra_des = _sift_like(rng, n_extra)
    order1 = rng.permutation(n + n_extra)
    uv1 = np.concatenate([p2, extra_uv])[order1]; d1 = np.concatenate([des1, extra_des])[order1]
    mk = lambda uv, d, t: [Keypoint(t, 1, uv[i].reshape(2, 1), uv[i].reshape(2, 1), d[i].reshape(-1, 1), [uv[i].reshape(2, 1)])
                           for i in range(len(uv))]
    kp0, kp1 = mk(p1, des, 0), mk(uv1, d1, 1)
    ext = Extractor(min_kp_dist=7, ctx=make_ctx(64, 64))
    # --- the reference's flow, pipeline.py:52-74 ---
    matches = ext.match_lists(kp0, kp1)
    kp0_m, kp1_m = [], []
    i1_nm = list(range(len(kp1)))
    for match in matches:
        kp0_m.append(copy.deepcopy(kp0[match.queryIdx]))
        kp1_m.append(copy.deepcopy(kp1[match.trainIdx]))
        if match.trainIdx in i1_nm:
            i1_nm.remove(match.trainIdx)
    kp1_nm = [kp1[i] for i in i1_nm]
    H0 = np.eye(4)
    inliers, H1 = ext.camera_pose(K, kp0_m, kp1_m, corr='2D-2D')
    kp0_m = [kp0_m[i] for i in inliers]
    kp1_m = [kp1_m[i] for i in inliers]
    landmarks, kp0_m, kp1_m = ext.triangulate_nonlinear(K, H0, H1, kp0_m, kp1_m, 1, max_err_reproj=2.0)
    # --- checks against the scene ---
    assert len(matches) >= 0.8 * n and len(kp1_nm) >= n_extra
    assert np.abs(H1[:3, :3] - R_gt).max() <= 5e-3 and np.abs(H1[:3, 3] - t_gt).max() <= 0.1
    assert len(landmarks) >= 0.7 * n and len(landmarks) == len(kp0_m) == len(kp1_m)
    P = np.array([l.p.reshape(3) for l in landmarks])
    assert (P[:, 2] > 0).all()
    return matches, inliers, H1, P


@pytest.mark.gpu
def test_bootstrap_flow_gpu_equals_cpu_twin_and_ground_truth():
    mg, ig, Hg, Pg = _bootstrap(_gpu_ctx)
    mc, ic, Hc, Pc = _bootstrap(_oracle_ctx)
    assert [(m.queryIdx, m.trainIdx) for m in mg] == [(m.queryIdx, m.trainIdx) for m in mc]
    assert len(np.setxor1d(ig, ic)) <= 2 and np.abs(Hg - Hc).max() <= 1e-6
    if ig == ic:
        assert Pg.shape == Pc.shape and np.abs(Pg - Pc).max() <= 1e-3 * np.abs(Pc).max()


def test_bootstrap_flow_cpu():
    _bootstrap(_oracle_ctx)


def _bootstrap_from_images(make_ctx, w=416, h=240):
    """the whole of Pipeline._get_init_state (reference pipeline.py:42-90) from two IMAGES: SIFT on both frames,
    ratio-test matching, five-point pose, triangulation -- the scene is a textured plane under a known image motion"""
    from vo_mi355x import Extractor, synthetic as syn
    frames, motions = syn.make_sequence(7, w=w, h=h, seed=77, margin=64)
    im0, im1, A = frames[0], frames[6], motions[6]
    K = np.array([[400.0, 0, (w - 1) / 2], [0, 400.0, (h - 1) / 2], [0, 0, 1]])
    ext = Extractor(min_kp_dist=7, ctx=make_ctx(w, h))
    kp0 = ext.extract(im0, 0, detector='custom', describe=True)
    kp1 = ext.extract(im1, 1, detector='custom', describe=True)
    matches = ext.match_lists(kp0, kp1)
    kp0_m = [copy.deepcopy(kp0[m.queryIdx]) for m in matches]
    kp1_m = [copy.deepcopy(kp1[m.trainIdx]) for m in matches]
    p0 = np.array([k.uv.reshape(2) for k in kp0_m], np.float64); p1 = np.array([k.uv.reshape(2) for k in kp1_m], np.float64)
    err = np.linalg.norm(p0 @ A[:, :2].T + A[:, 2] - p1, axis=1)
    assert len(matches) >= 150 and (err < 1.5).mean() >= 0.95                 # SIFT + ratio test find the true image motion
    inliers, H1 = ext.camera_pose(K, kp0_m, kp1_m, corr='2D-2D')
    assert len(inliers) >= 0.8 * len(matches) and (err[inliers] < 1.5).mean() >= 0.98
    kp0_i = [kp0_m[i] for i in inliers]; kp1_i = [kp1_m[i] for i in inliers]
    landmarks, k0, k1 = ext.triangulate_nonlinear(K, np.eye(4), H1, kp0_i, kp1_i, 1, max_err_reproj=2.0)
    assert len(landmarks) >= 0.5 * len(inliers) and all(l.p[2, 0] > 0 for l in landmarks)
    return ([(k.uv.reshape(2).tolist(), k.des.reshape(-1).tolist()) for k in kp0], [(m.queryIdx, m.trainIdx) for m in matches], inliers, H1)


@pytest.mark.gpu
def test_bootstrap_from_images_gpu_equals_cpu_twin():
    kg, mg, ig, Hg = _bootstrap_from_images(_gpu_ctx)
    kc, mc, ic, Hc = _bootstrap_from_images(_oracle_ctx)
    assert kg == kc and mg == mc                                              # SIFT keypoints, descriptors and matches identical
    assert len(np.setxor1d(ig, ic)) <= 2 and np.abs(Hg - Hc).max() <= 1e-6


def test_bootstrap_from_images_cpu():
    _bootstrap_from_images(_oracle_ctx, w=256, h=160)
