"""Drop-in Extractor / BundleAdjuster glue vs golden vectors produced by the REFERENCE's own classes
(tests/golden/gen_golden.py: reference code unmodified, OpenCV arithmetic supplied by the oracle through the cv2 stub).

CPU tests run the adapters on the oracle-backed test double; the gpu-marked twins run them on the real HIP library."""
import copy

import numpy as np
import pytest

from helpers import golden_tracks


def _kps(g, prefix):
    from vo_mi355x import Keypoint
    lens = g[prefix + "_hist_len"]
    off = np.concatenate([[0], np.cumsum(lens)])
    out = []
    for i in range(len(lens)):
        hist = [g[prefix + "_hist"][k].astype(np.float32).reshape(2, 1) for k in range(off[i], off[i + 1])]
        out.append(Keypoint(int(g[prefix + "_t_first"][i]), int(g[prefix + "_t_total"][i]),
                            g[prefix + "_uv_first"][i].astype(np.float32).reshape(2, 1), g[prefix + "_uv"][i].astype(np.float32).reshape(2, 1),
                            np.array([[g[prefix + "_tag"][i]]]), hist))
    return out


def _check_kps(kps, g, prefix, tags=True):
    assert len(kps) == len(g[prefix + "_t_first"]), (prefix, len(kps), len(g[prefix + "_t_first"]))
    if not len(kps):
        return
    assert np.array_equal(np.array([np.asarray(k.uv, np.float64).reshape(2) for k in kps]), g[prefix + "_uv"]), prefix
    assert np.array_equal(np.array([np.asarray(k.uv_first, np.float64).reshape(2) for k in kps]), g[prefix + "_uv_first"])
    assert np.array_equal([k.t_first for k in kps], g[prefix + "_t_first"]) and np.array_equal([k.t_total for k in kps], g[prefix + "_t_total"])
    assert np.array_equal([len(k.uv_history) for k in kps], g[prefix + "_hist_len"])
    assert np.array_equal(np.concatenate([np.array(k.uv_history, np.float64).reshape(-1, 2) for k in kps]), g[prefix + "_hist"])
    if tags:
        assert np.array_equal([float(np.asarray(k.des).reshape(-1)[0]) for k in kps], g[prefix + "_tag"])
    assert all(k.uv.shape == (2, 1) and k.uv_history[-1].shape == (2, 1) for k in kps)


def _check_lms(lms, g, prefix, tol=0.0):
    assert len(lms) == len(g[prefix + "_t_latest"])
    if not len(lms):
        return
    P = np.array([np.asarray(l.p, np.float64).reshape(3) for l in lms])
    if tol == 0.0:
        assert np.array_equal(P, g[prefix + "_p"])
    else:
        assert (np.linalg.norm(P - g[prefix + "_p"], axis=1) <= tol * np.linalg.norm(g[prefix + "_p"], axis=1)).all()
    assert np.array_equal([l.t_latest for l in lms], g[prefix + "_t_latest"])
    assert np.array_equal([float(np.asarray(l.des).reshape(-1)[0]) for l in lms], g[prefix + "_tag"])
    assert all(l.p.shape == (3, 1) for l in lms)


def _run_extractor_glue(make_ctx, golden_dir, dlt_tol):
    from vo_mi355x import Extractor, Landmark, Trajectory
    g = np.load(golden_dir + "/glue_s0.npz")
    frames = g["frames"]
    h, w = frames.shape[1:]
    ext = Extractor(min_kp_dist=7, ctx=make_ctx(w, h))
    ext._im_prev = frames[0]
    new0 = ext.extract(frames[0], 1, current_kp=[], detector='shi-tomasi', mask_radius=7, describe=False)
    _check_kps(new0, g, "ex0", tags=False)
    assert all(k.des.shape == (1, 1) and k.t_total == 1 and len(k.uv_history) == 1 for k in new0)
    cands = _kps(g, "in0")
    c1 = ext.extend_tracks(frames[1], copy.deepcopy(cands), np.inf)
    _check_kps(c1, g, "tr1")
    c1b = ext.extend_tracks(frames[1], copy.deepcopy(cands), 4.1)
    _check_kps(c1b, g, "tr1b")
    assert 0 < len(c1b) < len(c1)                       # the forward-forward "bidirectional" quirk is reproduced
    ext._im_prev = frames[1]
    c2 = ext.extend_tracks(frames[2], copy.deepcopy(c1), np.inf)
    _check_kps(c2, g, "tr2")
    rng = np.random.default_rng(0)
    for _ in range(6):
        rng.uniform(5, h - 5)                            # keep the generator in step with gen_golden.py
    lms = [Landmark(2, rng.normal(0, 1, (3, 1)), k.des.copy()) for k in c1]
    ln, kn, ld, kd = ext.extend_landmarks(frames[2], copy.deepcopy(lms), copy.deepcopy(c1), np.inf)
    _check_lms(ln, g, "el_l"); _check_kps(kn, g, "el_k"); _check_lms(ld, g, "el_ld"); _check_kps(kd, g, "el_kd")
    assert all(l.t_latest == 3 for l in ln) and len(kd) > 0
    new2 = ext.extract(frames[2], 3, current_kp=c2, detector='shi-tomasi', mask_radius=7, describe=False)
    _check_kps(new2, g, "ex2", tags=False)
    assert ext.extend_tracks(frames[2], [], np.inf) == [] and ext.extract(np.full((h, w), 9, np.uint8), 4, [], 'shi-tomasi', 7) == []
    # triangulate_tracks: grouping by t_first, length gate, filters, bearing gate
    traj = Trajectory({})
    for t, H in enumerate(g["tt_traj"]):
        traj.append(t, H)
    cand = _kps(g, "tt_in")
    ln, lk, rest = ext.triangulate_tracks(g["tt_K"], cand, traj, len(g["tt_traj"]) - 1, min_track_length=3,
                                          min_bearing_angle=0.5, max_err_reproj=2.0)
    _check_lms(ln, g, "tt_l", tol=dlt_tol); _check_kps(lk, g, "tt_k"); _check_kps(rest, g, "tt_rest")
    assert all(k.uv.dtype == np.float64 for k in lk)    # like the reference, refine() hands back float64 keypoints
    with pytest.raises(NotImplementedError):
        ext.extract(frames[0], 1, [], detector='shi-tomasi', describe=True)


def _run_ba_glue(make_ctx, golden_dir, name):
    """adjust(): landmark selection, resurrection of recently-dead landmarks, aliasing of the state lists, write-back"""
    from vo_mi355x import BundleAdjuster, Keypoint, Landmark, State, Trajectory
    import ba_oracle as bo
    g = np.load("%s/%s.npz" % (golden_dir, name))
    W, t_now, K = int(g["W"]), int(g["t_now"]), g["K"]

    def mk(prefix):
        ls, ks = [], []
        for r in golden_tracks(g, prefix):
            hist = [h.reshape(2, 1) for h in r["hist"]]
            ks.append(Keypoint(r["t_latest"] - len(hist) + 1, len(hist), hist[0], hist[-1], np.array([[r["tag"]]]), hist))
            ls.append(Landmark(r["t_latest"], r["p"].reshape(3, 1), np.array([[r["tag"]]])))
        return ls, ks
    al, ak = mk("act_"); dl, dk = mk("dead_")
    traj = Trajectory({})
    for t, H in enumerate(g["traj"]):
        traj.append(t, H.copy())
    state = State(al, ak, [], traj)
    ba = BundleAdjuster(verbosity=0, window_size=W, method='trf', xtol=1e-3, ftol=1e-3, ctx=make_ctx(64, 64))
    poses, points, obs, n_active, _, _ = ba.build_problem(copy.deepcopy(state), copy.deepcopy(dl), copy.deepcopy(dk), t_now)
    assert np.array_equal(bo.pack_x0(poses, points), g["x0"])                       # x0 == the reference's x0
    assert np.abs(bo.residual_norm(K, poses, points, obs) - g["r0"]).max() <= 1e-9
    n_act0 = len(state._landmarks)
    s2, dl2, dk2 = ba.adjust(state, dl, dk, K, t_now)
    assert s2 is state and len(state._landmarks) == int(g["ref_n_state_landmarks"])  # dead landmarks resurrected into the state (App. C-7)
    assert np.array_equal([float(l.des.reshape(-1)[0]) for l in dl2], g["ref_dead_tags"])
    assert len(dk2) == len(dl2) and len(state._landmarks_kp) == len(state._landmarks)
    # result quality: our solver reaches at most the reference's final cost on the same problem
    po = np.zeros((W, 6))
    from vo_mi355x.so3 import rodrigues_mat_to_vec
    for i in range(W):
        H = state._trajectory[t_now - i]
        po[i, :3], po[i, 3:] = rodrigues_mat_to_vec(H[:3, :3]), H[:3, 3]
    pts = np.array([l.p.reshape(3) for l in state._landmarks])
    assert bo.cost(K, po, pts, obs) <= float(g["ref_cost"]) * (1 + 1e-3)
    assert ba.last_stats["cost"] <= float(g["ref_cost"]) * (1 + 1e-3) and n_act0 == n_active


def _oracle_ctx(w, h):
    from oracle_context import OracleContext
    return OracleContext(w, h)


def _gpu_ctx(w, h):
    from vo_mi355x import VoContext
    return VoContext(w, h, max_pts=8192)      # the Extractor's own default capacity (landmarks + candidates of a 640 x 480 run exceed 2048)


def test_extractor_glue_matches_reference_cpu(golden_dir):
    _run_extractor_glue(_oracle_ctx, golden_dir, dlt_tol=0.0)


@pytest.mark.parametrize("name", ["ba_s0_n64_w4", "ba_s2_n256_w10"])
def test_bundle_adjuster_glue_matches_reference_cpu(golden_dir, name):
    _run_ba_glue(_oracle_ctx, golden_dir, name)


@pytest.mark.gpu
def test_extractor_glue_matches_reference_gpu(golden_dir):
    _run_extractor_glue(_gpu_ctx, golden_dir, dlt_tol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ba_s0_n64_w4", "ba_s2_n256_w10"])
def test_bundle_adjuster_glue_matches_reference_gpu(golden_dir, name):
    _run_ba_glue(_gpu_ctx, golden_dir, name)


def test_rodrigues_host_roundtrip(golden_dir):
    from vo_mi355x.so3 import rodrigues_mat_to_vec, rodrigues_vec_to_mat
    g = np.load(golden_dir + "/rodrigues.npz")
    for r, R, back in zip(g["r"], g["R"], g["back"]):
        assert np.allclose(rodrigues_vec_to_mat(r), R, atol=1e-14)
        assert np.allclose(rodrigues_mat_to_vec(R), back, atol=1e-12)


def _run_camera_pose(make_ctx):
    """Extractor.camera_pose(corr='3D-2D') (reference extractor.py:174-191): list in, (inlier list, 4x4 H) out"""
    from vo_mi355x import Extractor, Keypoint, Landmark, synthetic as syn
    K = syn.KITTI_K
    s = syn.make_ba_scene(n_pts=200, n_slots=2, seed=9, obs_noise=0.3)
    rng = np.random.default_rng(1)
    X = s["points_gt"].astype(np.float32); uv = s["obs"][0].astype(np.float32)
    out = rng.choice(200, 50, replace=False)
    uv[out] += rng.uniform(-70, 70, (50, 2)).astype(np.float32) + np.float32(12)
    lms = [Landmark(0, X[i].astype(np.float64).reshape(3, 1), np.zeros((1, 1))) for i in range(200)]
    kps = [Keypoint(0, 1, uv[i].reshape(2, 1), uv[i].reshape(2, 1), np.zeros((1, 1)), [uv[i].reshape(2, 1)]) for i in range(200)]
    ext = Extractor(min_kp_dist=7, ctx=make_ctx(64, 64))
    inliers, H = ext.camera_pose(K, lms, kps, corr='3D-2D', max_err_reproj=2.0)
    assert isinstance(inliers, list) and all(isinstance(i, int) for i in inliers)
    assert len(np.intersect1d(inliers, out)) <= 2 and len(inliers) >= 140
    pose = s["poses_gt"][0]
    assert np.abs(H[:3, :3] - syn.rodrigues(pose[:3])).max() <= 3e-3 and np.abs(H[:3, 3] - pose[3:]).max() <= 3e-2
    return inliers, H


def test_camera_pose_cpu():
    _run_camera_pose(_oracle_ctx)


@pytest.mark.gpu
def test_camera_pose_gpu_equals_cpu_twin():
    a, Ha = _run_camera_pose(_gpu_ctx)
    b, Hb = _run_camera_pose(_oracle_ctx)
    assert len(np.setxor1d(a, b)) <= 2 and np.abs(Ha - Hb).max() <= 1e-5


def _run_camera_pose_2d2d(make_ctx):
    """Extractor.camera_pose(corr='2D-2D') (reference extractor.py:162-172): findEssentialMat + recoverPose"""
    from vo_mi355x import Extractor, Keypoint, synthetic as syn
    import pnp_oracle as po
    K = syn.KITTI_K
    rng = np.random.default_rng(5)
    n = 300
    R = po.rodrigues(np.array([0.02, -0.03, 0.01])); t = np.array([0.3, 0.02, -0.8])
    X = np.stack([rng.uniform(-15, 15, n), rng.uniform(-3, 3, n), rng.uniform(8, 40, n)], 1)
    p1 = X @ K.T; p1 = p1[:, :2] / p1[:, 2:3] + rng.normal(0, 0.3, (n, 2))
    Xc = X @ R.T + t; p2 = Xc @ K.T; p2 = p2[:, :2] / p2[:, 2:3] + rng.normal(0, 0.3, (n, 2))
    out = rng.choice(n, 75, replace=False)
    p2[out] += rng.uniform(-70, 70, (75, 2)) + 12
    mk = lambda p: [Keypoint(0, 1, p[i].astype(np.float32).reshape(2, 1), p[i].astype(np.float32).reshape(2, 1), np.zeros((1, 1)), [])
                    for i in range(n)]
    ext = Extractor(min_kp_dist=7, ctx=make_ctx(64, 64))
    inliers, H = ext.camera_pose(K, mk(p1), mk(p2), corr='2D-2D')
    assert isinstance(inliers, list) and all(isinstance(i, int) for i in inliers)
    assert len(np.intersect1d(inliers, out)) <= 8 and len(inliers) >= 170
    assert np.abs(H[:3, :3] - R).max() <= 1e-2 and np.abs(H[:3, 3] - t / np.linalg.norm(t)).max() <= 0.15
    assert np.allclose(H[3], [0, 0, 0, 1])
    return inliers, H


def test_camera_pose_2d2d_cpu():
    _run_camera_pose_2d2d(_oracle_ctx)


@pytest.mark.gpu
def test_camera_pose_2d2d_gpu_equals_cpu_twin():
    a, Ha = _run_camera_pose_2d2d(_gpu_ctx)
    b, Hb = _run_camera_pose_2d2d(_oracle_ctx)
    assert len(np.setxor1d(a, b)) <= 2 and np.abs(Ha - Hb).max() <= 1e-6


def _sift_like(rng, n, dim=128):
    d = rng.gamma(0.6, 1.0, (n, dim))
    d = np.minimum(d / np.linalg.norm(d, axis=1, keepdims=True), 0.2)
    return np.floor(512 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)      # OpenCV SIFT: integer-valued floats


def _run_match_lists(make_ctx):
    """Extractor.match_lists (reference extractor.py:147-154): descriptor lists in, DMatch list (ratio test) out"""
    from vo_mi355x import Extractor, Keypoint
    rng = np.random.default_rng(3)
    d1 = _sift_like(rng, 150)
    perm = rng.permutation(150)[:110]
    d2 = np.concatenate([d1[perm] + rng.integers(-3, 4, (110, 128)).astype(np.float32), _sift_like(rng, 60)])
    mk = lambda d: [Keypoint(0, 1, np.zeros((2, 1)), np.zeros((2, 1)), d[i].reshape(-1, 1), []) for i in range(len(d))]
    ext = Extractor(min_kp_dist=7, ctx=make_ctx(64, 64))
    ms = ext.match_lists(mk(d1), mk(d2))
    got = {m.queryIdx: m.trainIdx for m in ms}
    assert len(got) == len(ms) and sum(1 for j, q in enumerate(perm) if got.get(int(q)) == j) >= 105     # the planted pairs
    assert all(isinstance(m.queryIdx, int) and isinstance(m.trainIdx, int) for m in ms)
    assert len(ms) <= 115                                                                              # the rest fails the ratio test
    with pytest.raises(ValueError):
        ext.match(d1, d2[:1])                               # the reference fails to unpack (m, n) with a single train descriptor
    return [(m.queryIdx, m.trainIdx, m.distance) for m in ms]


def test_match_lists_cpu():
    _run_match_lists(_oracle_ctx)


@pytest.mark.gpu
def test_match_lists_gpu_equals_cpu_twin():
    assert _run_match_lists(_gpu_ctx) == _run_match_lists(_oracle_ctx)


def test_state_deepcopy_fast_path_keeps_deepcopy_semantics():
    """Keypoint / Landmark implement __deepcopy__ (the reference's loop deep-copies every landmark every frame): result must
    equal the generic one -- independent arrays, views become standalone copies, objects that were identical stay identical"""
    from vo_mi355x import Keypoint, Landmark
    base = np.arange(6, dtype=np.float32).reshape(3, 2)
    v = base[1, :].reshape(2, 1)
    k = Keypoint(3, 2, v, v, np.zeros((128, 1), np.float32), [v, np.ones((2, 1))])
    k2 = copy.deepcopy(k)
    assert k2 is not k and k2.uv is k2.uv_first and k2.uv_history[0] is k2.uv and k2.uv is not k.uv and np.array_equal(k2.uv, k.uv)
    assert k2.uv.base is None and k2.uv.shape == (2, 1) and k2.uv.dtype == np.float32
    assert k2.uv_history is not k.uv_history and len(k2.uv_history) == 2 and (k2.t_first, k2.t_total) == (3, 2)
    k2.uv_history.append(1); k2.uv[0, 0] = 99
    assert len(k.uv_history) == 2 and k.uv[0, 0] == 2 and base[1, 0] == 2
    both = copy.deepcopy([k, k])
    assert both[0] is both[1]
    lm = Landmark(5, np.ones((3, 1)), k.des)
    pair = copy.deepcopy((k, lm))
    assert pair[1].des is pair[0].des and pair[1].p is not lm.p and np.array_equal(pair[1].p, lm.p) and pair[1].t_latest == 5
