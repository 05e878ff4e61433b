"""GPU: BASELINE config B shapes (1920x1080 frames, 5000 points, 20-frame BA window) -- exercises the 32-lanes-per-landmark
BA path, the 1024-lane workgroups and the large-image Shi-Tomasi path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ba_window20_matches_oracle():
    import ba_oracle as bo
    from vo_mi355x import VoContext, synthetic as syn
    s = syn.make_ba_scene(n_pts=700, n_slots=20, seed=5, visibility=0.85, width=1920, height=1080)
    with VoContext(64, 64, max_pts=64) as c:
        c.ba_upload(s["K"], s["poses0"], s["points0"], s["obs"])
        pr = c.ba_probe(lam=1e-3)
        ne = bo.normal_equations(s["K"], s["poses0"], s["points0"], s["obs"])
        S, rhs, _, _, _ = bo.schur_system(ne, 1e-3)
        rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel(pr["Hpp"], ne["Hpp"]) <= 1e-10 and rel(pr["Hll"], ne["Hll"]) <= 1e-10 and rel(pr["gp"], ne["gp"]) <= 1e-10
        assert rel(pr["S"], S) <= 1e-8 and rel(pr["rhs"], rhs) <= 1e-8
        dp, dl, _ = bo.lm_step(ne, 1e-3)
        assert rel(pr["dposes"], dp) <= 1e-6 and rel(pr["dpoints"], dl) <= 1e-6
        po, pt, st = c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c.ba_params(max_iters=30))
        ref = bo.solve(s["K"], s["poses0"], s["points0"], s["obs"], max_iters=30)
        assert st["iters"] == ref["iters"] and st["status"] == ref["status"]
        assert abs(st["cost"] - ref["cost"]) <= 1e-7 * ref["cost"]


def test_ba_5000_landmarks_window20_large_workgroups():
    """N = 5000, W = 20 selects the 1024-lane build / update kernels (157 partial sets)"""
    import ba_oracle as bo
    from vo_mi355x import VoContext, synthetic as syn
    s = syn.make_ba_scene(n_pts=5000, n_slots=20, seed=1, width=1920, height=1080)
    K = np.array([[1100.0, 0, 960.0], [0, 1100.0, 540.0], [0, 0, 1]])
    s = syn.make_ba_scene(n_pts=5000, n_slots=20, K=K, seed=1)
    with VoContext(64, 64, max_pts=64) as c:
        po, pt, st = c.ba_adjust(K, s["poses0"], s["points0"], s["obs"], c.ba_params(max_iters=12))
    assert st["n_obs"] == int((~np.isnan(s["obs"][..., 0])).sum()) > 95000 and st["cost"] < 0.05 * st["cost0"]
    assert abs(bo.cost(K, po, pt, s["obs"]) - st["cost"]) <= 1e-9 * st["cost"]


def test_frontend_1080p_5000_points():
    import vo_oracle as o
    from vo_mi355x import VoContext, synthetic as syn
    w, h, n = 1920, 1080, 5000
    frames, motions = syn.make_sequence(2, w=w, h=h, seed=99, margin=96)
    p0 = syn.grid_points(n, w, h, seed=3)
    with VoContext(w, h, max_pts=8192) as c:
        c.push_frame(frames[0]); c.push_frame(frames[1])
        p1, st, err, it = c.klt_track(p0, return_iters=True)
        q1, qs, qe, qi = o.klt(frames[0], frames[1], p0, return_iters=True)
        assert np.array_equal(p1, q1) and np.array_equal(st, qs) and np.array_equal(err, qe) and np.array_equal(it, qi)
        d = np.linalg.norm(p1 - syn.warp_points(motions[1], p0), axis=1)
        assert np.median(d) <= 0.05
        corners = c.shi_tomasi(p1, 7)
        mask = np.full((h, w), 255, np.uint8)
        for x, y in np.int32(p1):
            o.circle_mask(mask, (x, y), 7, 0)
        ref, _, nc = o.good_features(frames[1], mask, return_aux=True)
        assert nc <= 16384, "candidate capacity of k_st_select"
        assert np.array_equal(corners, ref)


def test_shi_tomasi_more_candidates_than_the_lds_sort_holds():
    """> 16384 NMS candidates (1080p, no exclusion discs, tiny quality level): the selection runs on the 16384 strongest
    (radix select) and stops when they fill max_corners; when they cannot, it goes on through the list in rank-ordered chunks
    (accepted corners carried along) -- like OpenCV, which never refuses an image."""
    import vo_oracle as o
    from vo_mi355x import VoContext
    rng = np.random.default_rng(12)
    h, w = 1080, 1920
    img = rng.integers(0, 256, (h, w)).astype(np.float32)
    img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1)) / 3.0          # slightly correlated noise: maxima everywhere
    img = np.clip(img, 0, 255).astype(np.uint8)
    with VoContext(w, h, max_pts=64) as c:
        c.push_frame(img)
        prm = c.st_params(max_corners=1000, quality_level=1e-4, min_distance=5)
        corners = c.shi_tomasi(None, 7, params=prm)
        _, _, nc = c.shi_tomasi_read()
        assert nc > 16384, nc
        ref = o.good_features(img, None, maxCorners=1000, qualityLevel=1e-4, minDistance=5, blockSize=31)
        assert len(corners) == 1000 and np.array_equal(corners, ref)
        # min distance 60 px: fewer than 1000 corners fit, so every one of the > 16384 candidates has to be looked at
        for md in (60, 25):
            corners = c.shi_tomasi(None, 7, params=c.st_params(max_corners=1000, quality_level=1e-4, min_distance=md))
            ref = o.good_features(img, None, maxCorners=1000, qualityLevel=1e-4, minDistance=md, blockSize=31)
            assert len(ref) < 1000 or md == 25
            assert np.array_equal(corners, ref), md
