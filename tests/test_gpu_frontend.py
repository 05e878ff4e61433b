"""GPU parity: pyramid / Scharr / KLT / Shi-Tomasi / DLT through the C ABI vs the CPU oracle.

Integer paths are compared BIT-EXACTLY (the KLT positions too: the GPU and the oracle sum
the normal equations exactly, so there is no order dependence)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx_big(seq3):
    from vo_mi355x import VoContext
    frames, _ = seq3
    c = VoContext(frames.shape[2], frames.shape[1], max_pts=4096)
    c.push_frame(frames[0])
    c.push_frame(frames[1])
    yield c
    c.close()


def test_pyramid_and_scharr_bit_exact(ctx_big, seq3):
    import vo_oracle as o
    frames, _ = seq3
    for which, fr in ((0, frames[0]), (1, frames[1])):
        pyr = o.build_pyramid(fr)
        assert len(pyr) == 4
        for l, ref in enumerate(pyr):
            img, der = ctx_big.pyramid_read(which, l)
            assert img.shape == ref.shape
            assert np.array_equal(img, ref), "pyramid level %d differs" % l
            assert np.array_equal(der, o.scharr(ref)), "Scharr level %d differs" % l


def test_klt_bit_exact_vs_oracle(ctx_big, seq3):
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    frames, motions = seq3
    p0 = syn.grid_points(2000, frames.shape[2], frames.shape[1])
    p1, st, err, it = ctx_big.klt_track(p0, return_iters=True)
    q1, qs, qe, qi = o.klt(frames[0], frames[1], p0, return_iters=True)
    assert np.array_equal(st, qs)
    assert np.array_equal(it, qi), "iteration counts differ"
    assert np.array_equal(p1, q1), "max |dp| = %g" % np.abs(p1 - q1).max()
    assert np.array_equal(err, qe)
    # analytic ground truth of the synthetic warp (KLT-2): median <= 0.05 px, 95 % <= 0.2 px
    d = np.linalg.norm(p1 - syn.warp_points(motions[1], p0), axis=1)
    assert np.median(d) <= 0.05 and np.percentile(d, 95) <= 0.2
    # vs OpenCV's float-accumulator order (oracle acc_mode 0): <= 1e-3 px for >= 99 % of the points
    f1, fs, _ = o.klt(frames[0], frames[1], p0, acc_mode=0)
    assert (np.abs(p1 - f1).max(axis=1) <= 1e-3).mean() >= 0.99


def test_klt_edge_cases(ctx_big, seq3):
    """points outside / on the border, flat regions are not present in the texture, n = 0 and n = 1"""
    import vo_oracle as o
    frames, _ = seq3
    h, w = frames.shape[1:]
    p0 = np.array([[0, 0], [w - 1, h - 1], [-40.0, 10.0], [w + 50.0, h + 50.0], [5.5, 370.25], [1240.9, 0.1],
                   [-15.0, -15.0], [620.123, 188.456], [w * 4.0, 10.0], [3.0, h - 0.01]], np.float32)
    p1, st, err, it = ctx_big.klt_track(p0, return_iters=True)
    q1, qs, qe, qi = o.klt(frames[0], frames[1], p0, return_iters=True)
    assert np.array_equal(st, qs) and np.array_equal(it, qi)
    assert np.array_equal(p1, q1) and np.array_equal(err, qe)
    e1, es, ee = ctx_big.klt_track(np.zeros((0, 2), np.float32))
    assert e1.shape == (0, 2)
    s1, ss, se = ctx_big.klt_track(p0[7:8])
    assert np.array_equal(s1, q1[7:8])


@pytest.mark.parametrize("win,max_level,max_count,eps", [(21, 3, 30, 0.01), (31, 2, 10, 0.03), (15, 1, 5, 0.1), (31, 0, 30, 0.03)])
def test_klt_parameter_sweep(ctx_big, seq3, win, max_level, max_count, eps):
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    frames, _ = seq3
    p0 = syn.grid_points(300, frames.shape[2], frames.shape[1], seed=11)
    prm = ctx_big.klt_params(win=win, max_level=max_level, max_count=max_count, epsilon=eps)
    p1, st, err, it = ctx_big.klt_track(p0, prm, return_iters=True)
    q1, qs, qe, qi = o.klt(frames[0], frames[1], p0, (win, win), max_level, (3, max_count, eps), return_iters=True)
    assert np.array_equal(st, qs) and np.array_equal(it, qi)
    assert np.array_equal(p1, q1) and np.array_equal(err, qe)


def test_klt_small_image_truncated_pyramid(seq_small):
    """320x240: level 3 would be 40x30 <= 31 -> only 3 levels, like buildOpticalFlowPyramid"""
    import vo_oracle as o
    from vo_mi355x import VoContext, synthetic as syn
    frames, _ = seq_small
    with VoContext(320, 240, max_pts=512) as c:
        c.push_frame(frames[0]); c.push_frame(frames[2])
        p0 = syn.grid_points(400, 320, 240, margin=8, seed=3)
        p1, st, err, it = c.klt_track(p0, return_iters=True)
        q1, qs, qe, qi = o.klt(frames[0], frames[2], p0, return_iters=True)
        assert (qi[:, 3] == -1).all()
        assert np.array_equal(it, qi) and np.array_equal(st, qs)
        assert np.array_equal(p1, q1) and np.array_equal(err, qe)


def test_klt_resident_chain_matches_host_chain(seq3):
    from vo_mi355x import VoContext, synthetic as syn
    frames, _ = seq3
    h, w = frames.shape[1:]
    p0 = syn.grid_points(500, w, h, seed=5)
    with VoContext(w, h, max_pts=1024) as c:
        c.push_frame(frames[0]); c.push_frame(frames[1])
        a1, _, _ = c.klt_track(p0)
        c.push_frame(frames[2])
        a2, as2, ae2 = c.klt_track(a1)
    with VoContext(w, h, max_pts=1024) as c:
        c.upload_sequence(frames)
        c.points_upload(p0)
        c.push_frame_resident(0); c.push_frame_resident(1)
        c.klt_track_resident(500)
        c.push_frame_resident(2)
        c.klt_track_resident(500)
        b2, bs2, be2 = c.points_download(500)
    assert np.array_equal(a2, b2) and np.array_equal(as2, bs2) and np.array_equal(ae2, be2)


def test_shi_tomasi_bit_exact(ctx_big, seq3):
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    frames, _ = seq3
    h, w = frames.shape[1:]
    cur = frames[1]
    # no mask
    c0 = ctx_big.shi_tomasi(None)
    eig, mask, nc = ctx_big.shi_tomasi_read()
    r0, reig, rnc = o.good_features(cur, None, return_aux=True)
    assert np.array_equal(eig, reig), "min-eig map differs: %g" % np.abs(eig - reig).max()
    assert (mask == 255).all() and nc == rnc
    assert np.array_equal(c0, r0)
    # with exclusion discs around 2000 tracked points (incl. border / negative coordinates)
    pts = syn.grid_points(2000, w, h, margin=0, seed=9) + np.float32(0.37)
    pts[:5] = [[-3.5, 4.2], [w - 0.5, h - 0.5], [w + 3.0, 10.0], [0.0, 0.0], [7.99, -6.99]]
    c1 = ctx_big.shi_tomasi(pts, mask_radius=7)
    _, mask1, nc1 = ctx_big.shi_tomasi_read()
    rmask = np.full((h, w), 255, np.uint8)
    for x, y in np.int32(pts):
        o.circle_mask(rmask, (x, y), 7, 0)
    assert np.array_equal(mask1, rmask)
    r1, _, rnc1 = o.good_features(cur, rmask, return_aux=True)
    assert nc1 == rnc1 and np.array_equal(c1, r1)
    # invariants: min distance respected, all outside the discs, integer coordinates
    d = np.linalg.norm(c1[:, None, :] - c1[None, :, :], axis=2) + np.eye(len(c1)) * 1e9
    assert d.min() >= 7.0 and (c1 == np.rint(c1)).all()
    assert (rmask[c1[:, 1].astype(int), c1[:, 0].astype(int)] == 255).all()


def test_harris_response_bit_exact(ctx_big, seq3):
    """`vo_st_params.use_harris` = cv2.goodFeaturesToTrack(useHarrisDetector=True, k) (the option the reference's dict extractor.py:21-24
    leaves off): response map, candidate count and the ordered corner list bit-equal to the oracle, with the
    reference's exclusion discs and without, k = 0.04 and another one; the default parameters still give the minimum-eigenvalue corners.
    Bit-equal to the ORACLE'S restatement (the scalar form of OpenCV's calcHarris); a SIMD build of cv2 evaluates the k term in float for all but
    the last few pixels of a row and differs from it by ~1 ulp, depending on its vector width -- not a cv2-bit-for-bit claim"""
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    frames, _ = seq3
    h, w = frames.shape[1:]
    cur = frames[1]
    for k in (0.04, 0.15):
        prm = ctx_big.st_params(use_harris=True, harris_k=k)
        c0 = ctx_big.shi_tomasi(None, params=prm)
        resp, mask, nc = ctx_big.shi_tomasi_read()
        r0, rresp, rnc = o.good_features(cur, None, return_aux=True, useHarrisDetector=True, k=k)
        assert np.array_equal(resp, rresp), "Harris response differs: %g" % np.abs(resp - rresp).max()
        assert nc == rnc and np.array_equal(c0, r0) and len(c0) > 50
    pts = syn.grid_points(2000, w, h, margin=0, seed=9) + np.float32(0.37)
    prm = ctx_big.st_params(use_harris=True)
    c1 = ctx_big.shi_tomasi(pts, mask_radius=7, params=prm)
    rmask = np.full((h, w), 255, np.uint8)
    for x, y in np.int32(pts):
        o.circle_mask(rmask, (x, y), 7, 0)
    assert np.array_equal(c1, o.good_features(cur, rmask, useHarrisDetector=True))
    assert np.array_equal(ctx_big.shi_tomasi(None), o.good_features(cur, None))          # the default is untouched
    assert not np.array_equal(c0, ctx_big.shi_tomasi(None))


def test_shi_tomasi_vs_opencv_float_order(ctx_big, seq3):
    """the bridge to OpenCV's own arithmetic (ST-1 / ST-2): cv2's boxFilter keeps float running sums, the HIP path and the oracle's
    default keep exact int32 sums (oracle/vo_oracle.c header).  Against the oracle in OpenCV's float order (exact_int=False):
    eigenvalue map <= 1e-5 of its maximum, corner sets equal up to near-ties (<= 2 % symmetric difference), with and without discs."""
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    frames, _ = seq3
    h, w = frames.shape[1:]
    cur = frames[1]
    got = ctx_big.shi_tomasi(None)
    eig, _, _ = ctx_big.shi_tomasi_read()
    feig = o.min_eig(cur, exact_int=False)
    assert np.abs(eig - feig).max() <= 1e-5 * feig.max()
    want = o.good_features(cur, None, exact_int=False)
    diff = set(map(tuple, got.tolist())) ^ set(map(tuple, want.tolist()))
    assert len(diff) <= 0.02 * len(want), (len(diff), len(want))
    pts = syn.grid_points(2000, w, h, margin=0, seed=9) + np.float32(0.37)
    rmask = np.full((h, w), 255, np.uint8)
    for x, y in np.int32(pts):
        o.circle_mask(rmask, (x, y), 7, 0)
    got = ctx_big.shi_tomasi(pts, mask_radius=7)
    want = o.good_features(cur, rmask, exact_int=False)
    diff = set(map(tuple, got.tolist())) ^ set(map(tuple, want.tolist()))
    assert len(diff) <= 0.02 * max(len(want), 1), (len(diff), len(want))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_shi_tomasi_dense_candidates_flush_the_band_lists(ctx_big, seq3, seed):
    """white noise at a tiny quality level: about every ninth pixel survives the 3 x 3 suppression, so every band of the fused kernel
    fills and empties its LDS candidate list many times (the decision to empty it must be taken identically by all waves of a
    workgroup).  Corners, candidate count and the eigenvalue map against the oracle, three images."""
    import vo_oracle as o
    rng = np.random.default_rng(100 + seed)
    img = rng.integers(0, 256, (376, 1241), dtype=np.uint8)
    ctx_big.push_frame(img)
    prm = ctx_big.st_params(max_corners=1000, quality_level=1e-4, min_distance=7.0, block_size=31)
    got = ctx_big.shi_tomasi(None, params=prm)
    eig, _, nc = ctx_big.shi_tomasi_read()
    want, weig, wnc = o.good_features(img, None, 1000, 1e-4, 7.0, 31, return_aux=True)
    frames, _ = seq3
    ctx_big.push_frame(frames[0]); ctx_big.push_frame(frames[1])       # the module's context goes back to its two frames
    assert np.array_equal(eig, weig) and nc == wnc and nc > 20000, (nc, wnc)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("maxc,q,md,bs,radius", [(200, 0.05, 12.0, 15, 5), (1000, 0.01, 3.0, 7, 0), (50, 0.2, 0.5, 3, 10), (4000, 0.001, 5.0, 31, 7)])
def test_shi_tomasi_parameter_sweep(seq_small, maxc, q, md, bs, radius):
    import vo_oracle as o
    from vo_mi355x import VoContext, synthetic as syn
    frames, _ = seq_small
    with VoContext(320, 240, max_pts=512) as c:
        c.push_frame(frames[1])
        pts = syn.grid_points(100, 320, 240, margin=0, seed=2)
        prm = c.st_params(max_corners=maxc, quality_level=q, min_distance=md, block_size=bs)
        got = c.shi_tomasi(pts if radius else None, mask_radius=radius, params=prm)
        rmask = None
        if radius:
            rmask = np.full((240, 320), 255, np.uint8)
            for x, y in np.int32(pts):
                o.circle_mask(rmask, (x, y), radius, 0)
        ref = o.good_features(frames[1], rmask, maxc, q, md, bs)
        assert np.array_equal(got, ref)


def test_shi_tomasi_explicit_mask_and_empty(seq_small):
    import vo_oracle as o
    from vo_mi355x import VoContext
    frames, _ = seq_small
    with VoContext(320, 240, max_pts=64) as c:
        c.push_frame(frames[0])
        m = np.zeros((240, 320), np.uint8)
        assert c.shi_tomasi(None, mask=m).shape == (0, 2)          # everything masked -> no corners
        m[50:200, 40:300] = 1
        assert np.array_equal(c.shi_tomasi(None, mask=m), o.good_features(frames[0], m))
        c.push_frame(np.full((240, 320), 77, np.uint8))             # flat image -> no corners
        assert c.shi_tomasi(None).shape == (0, 2)


def test_dlt_vs_oracle_and_ground_truth(golden_dir):
    import vo_oracle as o
    from vo_mi355x import VoContext
    g = np.load(golden_dir + "/tri_s0.npz")
    K, H0, H1 = g["K"], g["H0"], g["H1"]
    P0 = (K @ H0[:3]).astype(np.float32)
    P1 = (K @ H1[:3]).astype(np.float32)
    uv0, uv1 = g["uv0"].astype(np.float32), g["uv1"].astype(np.float32)
    with VoContext(64, 64, max_pts=512) as c:
        X4, depth1, reproj = c.triangulate(P0, P1, uv0, uv1, K, H0, H1)
    R4 = o.triangulate(P0, P1, uv0, uv1)
    X = (X4[:3] / X4[3]).T
    Xr = (R4[:3] / R4[3]).T
    # DLT-1: dehomogenised points agree to 1e-4 relative (f32 outputs, sign of the null vector is free)
    rel = np.linalg.norm(X - Xr, axis=1) / np.linalg.norm(Xr, axis=1)
    assert rel.max() <= 1e-4, rel.max()
    # TRI-1: the reference's own filter statistics (golden G2, generated by importing the reference)
    ok = np.abs(g["depth1"]) > 1e-3
    assert np.allclose(depth1[ok], g["depth1"][ok], rtol=1e-3, atol=1e-3)
    assert np.abs(reproj - g["f0_all"]).max() <= 1e-3 * max(1.0, g["f0_all"].max()) or \
        np.percentile(np.abs(reproj - g["f0_all"]), 99) <= 1e-3
    keep = np.nonzero((depth1 > 0) & (reproj < float(g["max_err"])))[0]
    border = np.abs(g["f0_all"] - float(g["max_err"])) < 1e-3
    sym = set(keep.tolist()) ^ set(g["keep"].tolist())
    assert all(border[i] or abs(g["depth1"][i]) < 1e-3 for i in sym)


def test_klt_two_keypoints_per_wave_is_bit_identical(ctx_big, seq3):
    """k_klt_track2 (experiment: compiled with -DVO_EXPERIMENTS only, vo_tuning.klt_pair): the two halves of a wave track one keypoint each and share
    the lane-uniform work.  Same integer sums and float expressions -> the same bits as the shipped kernel, for an odd point count, points at and
    beyond the borders and a truncated iteration budget (it is slower -- profiles/r03_klt_pair_counters.txt -- and is not in the shipped library)"""
    from vo_mi355x import VoError, synthetic as syn
    try:
        ctx_big.set_tuning(klt_pair=3)
    except VoError:
        pytest.skip("the library was built without -DVO_EXPERIMENTS (the default): k_klt_track2 is not in it")
    frames, _ = seq3
    p0 = syn.grid_points(1999, 1241, 376, seed=5)
    p0[::97] = [-40.0, 10.0]; p0[1::131] = [1300.0, 400.0]; p0[2::211] = [620.5, -31.5]
    p0[5] = [1239.7, 374.2]; p0[6] = [-31.0, -31.0]; p0[7] = [3.25, 370.9]
    try:
        for prm in (None, ctx_big.klt_params(max_count=2), ctx_big.klt_params(win=21, max_level=2)):
            ctx_big.set_tuning(klt_pair=0)
            a = ctx_big.klt_track(p0, prm, return_iters=True)
            for mode in (3, 4):
                ctx_big.set_tuning(klt_pair=mode)
                b = ctx_big.klt_track(p0, prm, return_iters=True)
                for x, y in zip(a, b):
                    assert np.array_equal(x, y, equal_nan=True)
    finally:
        ctx_big.set_tuning(klt_pair=0)
