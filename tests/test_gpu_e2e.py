"""End to end: the reference's whole loop -- Pipeline._get_init_state (pipeline.py:42-90) and Pipeline.step
(pipeline.py:92-167) -- restated over the drop-in Extractor / BundleAdjuster / State classes, on a rendered sequence with
a known camera trajectory (two textured fronto-parallel planes at different depths: each one's image motion under a rolling,
approaching, side-stepping camera is a similarity warp that synthetic.render_frame renders exactly).  Every numerical call (SIFT, matching, five-point pose, triangulation, KLT,
P3P-RANSAC pose, bundle adjustment, Shi-Tomasi re-detection) runs on the device; the estimated trajectory is compared
with ground truth up to the bootstrap's unit-baseline scale, and with the same loop over the CPU oracle context."""
import copy

import numpy as np
import pytest

from test_adapters import _gpu_ctx, _oracle_ctx

W, H, F = 416, 240, 400.0
Z_BG, Z_FG = 10.0, 6.5                                     # two fronto-parallel textured planes: parallax, no planar ambiguity
FG_RECT = (120.0, 60.0, 300.0, 185.0)                      # extent of the near plane in frame-0 pixels (x0, y0, x1, y1)
K = np.array([[F, 0, (W - 1) / 2], [0, F, (H - 1) / 2], [0, 0, 1]])


def _pose(t):
    """camera pose of frame t (frame-0 camera -> frame-t camera): roll, side-step, approach"""
    ang = 0.004 * t
    Hm = np.eye(4)
    Hm[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
    Hm[:3, 3] = np.array([0.10, 0.03, -0.06]) * t
    return Hm


def _plane_motion(Hm, Z):
    """image motion (2x3 affine, frame 0 -> frame t) of the fronto-parallel plane at depth Z under the pose Hm:
    p' - c = Z / (Z + Tz) R2 (p - c) + f T_xy / (Z + Tz)"""
    c = K[:2, 2]
    s = Z / (Z + Hm[2, 3])
    R2 = Hm[:2, :2]
    A = np.zeros((2, 3)); A[:, :2] = s * R2; A[:, 2] = c - s * R2 @ c + F * Hm[:2, 3] / (Z + Hm[2, 3])
    return A


def _frames(n):
    from vo_mi355x import synthetic as syn
    margin = 96
    tex_bg = syn.make_texture(H + 2 * margin, W + 2 * margin, 2024)
    tex_fg = syn.make_texture(H + 2 * margin, W + 2 * margin, 4048)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    out = []
    for t in range(n):
        Hm = _pose(t)
        A_bg, A_fg = _plane_motion(Hm, Z_BG), _plane_motion(Hm, Z_FG)
        img = syn.render_frame(tex_bg, A_bg, W, H, margin)
        fg = syn.render_frame(tex_fg, A_fg, W, H, margin)
        Ainv = np.linalg.inv(np.vstack([A_fg, [0, 0, 1]]))[:2]
        x0 = Ainv[0, 0] * xs + Ainv[0, 1] * ys + Ainv[0, 2]; y0 = Ainv[1, 0] * xs + Ainv[1, 1] * ys + Ainv[1, 2]
        inside = (x0 >= FG_RECT[0]) & (x0 < FG_RECT[2]) & (y0 >= FG_RECT[1]) & (y0 < FG_RECT[3])
        out.append((np.where(inside, fg, img).astype(np.uint8), Hm))
    return out


def _run(make_ctx, n_steps, t_init=(0, 4), frames=None, K=K):
    """`frames`: list of (image, ground-truth pose) -- default: the rendered two-plane scene of this module"""
    from vo_mi355x import BundleAdjuster, Extractor, State, Trajectory
    if frames is None:
        frames = _frames(t_init[1] + n_steps + 1)
    ctx = make_ctx(frames[0][0].shape[1], frames[0][0].shape[0])
    # ---- Pipeline.__init__ (pipeline.py:17-31) ----
    ba_window, min_kp_dist, max_bidir, max_reproj, min_angle = 4, 7, np.inf, 2.0, 0.5
    extractor = Extractor(min_kp_dist=min_kp_dist, ctx=ctx)
    adjuster = BundleAdjuster(verbosity=0, window_size=ba_window, method='trf', xtol=1e-3, ftol=1e-3, ctx=ctx)
    t_step = 1
    dead, dead_kp = [], []
    # ---- _get_init_state (pipeline.py:42-90) ----
    t0, t1 = t_init
    im0, im1 = frames[t0][0], frames[t1][0]
    kp0 = extractor.extract(im0, 0, detector='custom', describe=True)
    kp1 = extractor.extract(im1, 1, detector='custom', describe=True)
    matches = extractor.match_lists(kp0, kp1)
    kp0_m, kp1_m = [], []
    i1_nm = list(range(len(kp1)))
    for m in matches:
        kp0_m.append(copy.deepcopy(kp0[m.queryIdx]))
        kp1_m.append(copy.deepcopy(kp1[m.trainIdx]))
        if m.trainIdx in i1_nm:
            i1_nm.remove(m.trainIdx)
    kp1_nm = [kp1[i] for i in i1_nm]
    H0 = np.eye(4)
    inliers, H1 = extractor.camera_pose(K, kp0_m, kp1_m, corr='2D-2D')
    kp0_m = [kp0_m[i] for i in inliers]; kp1_m = [kp1_m[i] for i in inliers]
    landmarks, kp0_m, kp1_m = extractor.triangulate_nonlinear(K, H0, H1, kp0_m, kp1_m, t_step, max_err_reproj=max_reproj)
    traj = Trajectory({})
    traj.append(0, H0); traj.append(1, H1)
    state = State(landmarks, kp1_m, kp1_nm, traj)
    t_loader = t1
    extractor._im_prev = frames[t_loader][0]
    n_boot = len(landmarks)
    # ---- step (pipeline.py:92-164) ----
    sizes = []
    for _ in range(n_steps):
        t_step += 1; t_loader += 1
        im = frames[t_loader][0]
        state._candidates_kp = extractor.extend_tracks(im, state._candidates_kp, max_bidir_error=max_bidir)
        state._landmarks, state._landmarks_kp, ld, lkd = extractor.extend_landmarks(im, state._landmarks, state._landmarks_kp,
                                                                                     max_bidir_error=max_bidir)
        dead += copy.deepcopy(ld); dead_kp += copy.deepcopy(lkd)
        extractor._im_prev = im.copy()
        inl, Hk = extractor.camera_pose(K, state._landmarks, state._landmarks_kp, corr='3D-2D', max_err_reproj=max_reproj)
        inl_set = set(inl)
        lms, lkp = [], []
        for i in range(len(state._landmarks)):
            if i in inl_set:
                lms.append(state._landmarks[i]); lkp.append(state._landmarks_kp[i])
            else:
                dead.append(copy.deepcopy(state._landmarks[i])); dead_kp.append(copy.deepcopy(state._landmarks_kp[i]))
        state._landmarks, state._landmarks_kp = lms, lkp
        state._trajectory.append(t_step, Hk)
        l_new, lk_new, state._candidates_kp = extractor.triangulate_tracks(K, state._candidates_kp, state._trajectory, t_curr=t_step,
                                                                          min_track_length=3, min_bearing_angle=min_angle,
                                                                          max_err_reproj=max_reproj, refine=True)
        state._landmarks_kp += lk_new; state._landmarks += l_new
        state, dead, dead_kp = adjuster.adjust(state, dead, dead_kp, K, t_step)
        state._candidates_kp += extractor.extract(im, t_step, state._landmarks_kp + state._candidates_kp, detector='shi-tomasi',
                                                  mask_radius=min_kp_dist, describe=False)
        sizes.append((len(state._landmarks), len(state._candidates_kp), len(l_new)))
    # ---- against the ground truth ----
    gt1 = frames[t1][1]
    unit = np.linalg.norm(gt1[:3, 3])                       # the bootstrap fixes |t(t1)| = 1
    errs = []
    for k in range(1, t_step + 1):
        Hk = state._trajectory[k]
        gt = frames[t1 + k - 1][1]
        cosang = (np.trace(Hk[:3, :3] @ gt[:3, :3].T) - 1) / 2
        errs.append((np.degrees(np.arccos(np.clip(cosang, -1, 1))), np.linalg.norm(Hk[:3, 3] - gt[:3, 3] / unit)))
    return dict(n_boot=n_boot, n_matches=len(matches), sizes=sizes, errs=np.array(errs),
                traj=np.array([state._trajectory[k] for k in range(t_step + 1)]))


def _check(r, n_steps):
    assert r["n_matches"] >= 200 and r["n_boot"] >= 150
    assert len(r["sizes"]) == n_steps and all(s[0] >= 100 for s in r["sizes"])          # the map never collapses
    assert sum(s[2] for s in r["sizes"]) > 0                                            # new landmarks are triangulated from tracks
    rot, tra = r["errs"][:, 0], r["errs"][:, 1]
    assert rot.max() <= 0.5, rot                                                        # degrees
    assert tra.max() <= 0.25 * (1 + 0.25 * n_steps), tra                                # in bootstrap baselines (scale drift of a planar scene)


@pytest.mark.gpu
def test_pipeline_loop_on_the_device():
    r = _run(_gpu_ctx, 8)
    _check(r, 8)


@pytest.mark.gpu
def test_pipeline_loop_gpu_equals_cpu_twin():
    g = _run(_gpu_ctx, 2)
    c = _run(_oracle_ctx, 2)
    assert g["n_matches"] == c["n_matches"] and g["n_boot"] == c["n_boot"]
    assert g["sizes"] == c["sizes"]
    assert np.abs(g["traj"] - c["traj"]).max() <= 1e-5


def test_pipeline_loop_cpu():
    r = _run(_oracle_ctx, 2)
    _check(r, 2)
