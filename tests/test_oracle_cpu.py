"""CPU-only tests: the oracle against the reference-generated golden vectors and against analytic truth."""
import glob

import numpy as np
import pytest


def _load(path):
    from helpers import golden_ba_problem, ref_stub_cv2
    cv2 = ref_stub_cv2()   # Rodrigues only
    g = np.load(path)
    return (g,) + golden_ba_problem(g, lambda R: cv2.Rodrigues(R)[0])


def test_golden_files_present(golden_dir):
    assert len(glob.glob(golden_dir + "/ba_*.npz")) >= 4 and len(glob.glob(golden_dir + "/tri_*.npz")) >= 2


@pytest.mark.parametrize("name", ["ba_s0_n64_w4", "ba_s1_n64_w4", "ba_s2_n256_w10", "ba_s0_n256_w10"])
def test_ba_oracle_pinned_to_reference(golden_dir, name):
    """x0 packing, landmark selection (incl. resurrected dead landmarks), observation table, sparsity pattern and
    residual vector equal what the reference itself produced (bundle_adjuster.py:127-194)."""
    import ba_oracle as bo
    g, K, poses, points, obs, tags = _load("%s/%s.npz" % (golden_dir, name))
    assert np.array_equal(tags, g["refine_tags"])
    assert np.array_equal(bo.pack_x0(poses, points), g["x0"])
    m = bo.valid_mask(obs)
    slot = np.concatenate([np.full(int(m[i].sum()), i) for i in range(obs.shape[0])])
    lm = np.concatenate([np.nonzero(m[i])[0] for i in range(obs.shape[0])])
    assert np.array_equal(slot, g["obs_slot"]) and np.array_equal(lm, g["obs_lm"])
    r0 = bo.residual_norm(K, poses, points, obs)
    assert np.abs(r0 - g["r0"]).max() <= 1e-9
    rows, cols = bo.sparsity_coo(obs)
    assert set(zip(rows.tolist(), cols.tolist())) == set(zip(g["A_row"].tolist(), g["A_col"].tolist()))
    assert tuple(g["A_shape"]) == (len(r0), 3 * len(points) + 6 * obs.shape[0])
    # scipy's robust cost at x0
    assert abs(bo.cost(K, poses, points, obs) - 0.5 * bo.huber_rho(g["r0"] ** 2).sum()) < 1e-9 * bo.cost(K, poses, points, obs)


def test_ba_analytic_jacobian_vs_reference_fd(golden_dir):
    import ba_oracle as bo
    g, K, poses, points, obs, _ = _load(golden_dir + "/ba_s0_n64_w4.npz")
    J = bo.dense_jacobian_norm_form(K, poses, points, obs)
    Jfd = np.zeros_like(J)
    Jfd[g["Jfd_row"], g["Jfd_col"]] = g["Jfd_val"]
    big = np.abs(J) > 1.0
    assert (np.abs(J - Jfd)[big] / np.abs(J)[big]).max() <= 5e-3
    # and against our own central differences (tight)
    x0 = bo.pack_x0(poses, points)
    N, W = len(points), len(poses)
    rng = np.random.default_rng(0)
    for col in rng.choice(len(x0), 12, replace=False):
        h = 1e-6 * max(1.0, abs(x0[col]))
        xp, xm = x0.copy(), x0.copy()
        xp[col] += h; xm[col] -= h
        pp, qp = bo.unpack_x(xp, N, W); pm, qm = bo.unpack_x(xm, N, W)
        fd = (bo.residual_norm(K, pp, qp, obs) - bo.residual_norm(K, pm, qm, obs)) / (2 * h)
        assert np.abs(fd - J[:, col]).max() <= 1e-5 * max(1.0, np.abs(J[:, col]).max())


@pytest.mark.parametrize("name", ["ba_s0_n64_w4", "ba_s2_n256_w10"])
def test_ba_oracle_solver_beats_reference_cost(golden_dir, name):
    import ba_oracle as bo
    g, K, poses, points, obs, _ = _load("%s/%s.npz" % (golden_dir, name))
    res = bo.solve(K, poses, points, obs, max_iters=50, ftol=1e-3, xtol=1e-3)
    assert res["cost"] <= float(g["ref_cost"]) * (1 + 1e-3)
    res = bo.solve(K, poses, points, obs, max_iters=200, ftol=1e-12, xtol=1e-12)
    assert res["cost"] <= float(g["tight_cost"]) * (1 + 1e-4)


@pytest.mark.parametrize("name", ["s0_n64_w4", "s1_n64_w4", "s2_n256_w10", "s0_n256_w10"])
def test_ba_pose_deltas_and_points_vs_reference_solutions(golden_dir, name):
    """BA-6 / BA-7 (SURVEY 8a'): gauge-free pose deltas and points of the oracle LM against the reference's own solutions
    (bundle_adjuster.py:189-213) -- see helpers.ba_solution_parity; the GPU twin is tests/test_gpu_ba.py."""
    import ba_oracle as bo
    from helpers import ba_solution_parity
    g, K, poses, points, obs, _ = _load("%s/ba_%s.npz" % (golden_dir, name))
    gp = np.load("%s/bapolish_%s.npz" % (golden_dir, name))

    def solve(max_iters, ftol, xtol):
        r = bo.solve(K, poses, points, obs, max_iters=max_iters, ftol=ftol, xtol=xtol)
        return r["poses"], r["points"], r["cost"]
    rep = ba_solution_parity(solve, g, gp, K, poses, points, obs)
    print(name, {k: (tuple(float("%.3g" % x) for x in v) if isinstance(v, tuple) else float("%.6g" % v)) for k, v in rep.items()})


def test_ba_full_size_golden_pinned(golden_dir):
    """the BASELINE-shape golden (2000 landmarks, 10-frame window; 12 064 observations): selection, x0 and residual vector
    equal the reference's (bundle_adjuster.py:127-194)"""
    import ba_oracle as bo
    g, K, poses, points, obs, tags = _load(golden_dir + "/bafull_s0_n2000_w10.npz")
    assert np.array_equal(tags, g["refine_tags"]) and np.array_equal(bo.pack_x0(poses, points), g["x0"])
    r0 = bo.residual_norm(K, poses, points, obs)
    assert len(r0) == len(g["r0"]) and np.abs(r0 - g["r0"]).max() <= 1e-9


def test_schur_step_equals_dense_solve(golden_dir):
    import ba_oracle as bo
    g, K, poses, points, obs, _ = _load(golden_dir + "/ba_s1_n64_w4.npz")
    ne = bo.normal_equations(K, poses, points, obs)
    lam = 1e-2
    dp, dl, pred = bo.lm_step(ne, lam)
    # dense (H + lam diag H) d = -g
    e, Jp, Jl, m = bo.jacobian_blocks(K, poses, points, obs)
    W, N = m.shape
    rows = []
    wts = []
    for i in range(W):
        for j in np.nonzero(m[i])[0]:
            for k in range(2):
                r = np.zeros(6 * W + 3 * N)
                r[6 * i:6 * i + 6] = Jp[i, j, k]
                r[6 * W + 3 * j:6 * W + 3 * j + 3] = Jl[i, j, k]
                rows.append(r)
            s = e[i, j] @ e[i, j]
            wts += [float(bo.huber_weight(np.array(s)))] * 2
    J = np.array(rows); w = np.array(wts)
    ev = np.concatenate([e[i][m[i]].reshape(-1) for i in range(W)])
    H = J.T @ (w[:, None] * J)
    gvec = J.T @ (w * ev)
    d = np.linalg.solve(H + lam * np.diag(np.maximum(np.diag(H), 1e-12)), -gvec)
    assert np.allclose(d[:6 * W], dp.reshape(-1), rtol=1e-6, atol=1e-9)
    assert np.allclose(d[6 * W:], dl.reshape(-1), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("seed", [0, 1])
def test_triangulation_oracle_vs_reference_filters(golden_dir, seed):
    """DLT oracle: reprojects onto the measurements; reference filter decisions (golden G2) follow from its outputs."""
    import vo_oracle as o
    g = np.load("%s/tri_s%d.npz" % (golden_dir, seed))
    K, H0, H1 = g["K"], g["H0"], g["H1"]
    P0, P1 = (K @ H0[:3]).astype(np.float32), (K @ H1[:3]).astype(np.float32)
    X4 = o.triangulate(P0, P1, g["uv0"].astype(np.float32), g["uv1"].astype(np.float32))
    assert np.array_equal(X4.T, g["X4"])                      # the golden was generated with this very DLT
    assert np.allclose(np.linalg.norm(X4, axis=0), 1.0, atol=1e-6)
    # numpy SVD cross-check of the null vector (up to sign)
    for i in range(0, X4.shape[1], 17):
        A = np.zeros((4, 4))
        for v, (P, uv) in enumerate(((P0, g["uv0"]), (P1, g["uv1"]))):
            x, y = np.float32(uv[i])
            A[2 * v] = float(x) * P[2].astype(np.float64) - P[0]
            A[2 * v + 1] = float(y) * P[2].astype(np.float64) - P[1]
        v = np.linalg.svd(A)[2][3]
        x = X4[:, i].astype(np.float64)
        assert min(np.abs(v - x).max(), np.abs(v + x).max()) < 1e-6
    X3 = g["X3"].astype(np.float64)
    depth1 = (X3 @ H1[2, :3]) + H1[2, 3]
    assert np.allclose(depth1, g["depth1"], rtol=1e-12, atol=1e-9)
    keep = np.nonzero((g["depth1"] > 0) & (g["f0_all"] < float(g["max_err"])))[0]
    assert np.array_equal(keep, g["keep"])                    # refine() == the two filters
    assert np.array_equal(g["out_p"], X3[g["keep"]])          # ... and never moves a point


def test_rodrigues_golden(golden_dir):
    import ba_oracle as bo
    g = np.load(golden_dir + "/rodrigues.npz")
    for r, R, back in zip(g["r"], g["R"], g["back"]):
        assert np.allclose(bo.rodrigues_exp(r), R, atol=1e-14)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert np.allclose(bo.rodrigues_exp(back), R, atol=1e-6)


def test_pyrdown_scharr_properties():
    import vo_oracle as o
    rng = np.random.default_rng(0)
    flat = np.full((37, 53), 91, np.uint8)
    assert (o.pyr_down(flat) == 91).all() and o.pyr_down(flat).shape == (19, 27)
    assert (o.scharr(flat) == 0).all()
    ramp = np.tile(np.arange(60, dtype=np.uint8) * 2, (40, 1))
    d = o.scharr(ramp)
    assert (d[:, 1:-1, 0] == 2 * 2 * 16).all() and (d[..., 1] == 0).all()     # (3+10+3) * (I(x+1)-I(x-1))
    assert (d[:, 0, 0] == 0).all()                                              # reflect-101: I(-1) = I(1)
    img = rng.integers(0, 256, (45, 64), dtype=np.uint8)
    # direct 5x5 evaluation of one interior and one corner pixel
    k = np.array([1, 4, 6, 4, 1])
    pad = np.pad(img.astype(np.int64), 2, mode="reflect")
    pd = o.pyr_down(img)
    for (y, x) in ((0, 0), (5, 7), (22, 31), (22, 0)):
        blk = pad[2 * y:2 * y + 5, 2 * x:2 * x + 5]
        assert pd[y, x] == ((k[:, None] * k[None, :] * blk).sum() + 128) >> 8
    assert o.pyr_levels(1241, 376) == 3 and o.pyr_levels(320, 240) == 2 and o.pyr_levels(1920, 1080) == 3


def test_klt_oracle_tracks_known_warp(seq3):
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    frames, motions = seq3
    p0 = syn.grid_points(500, frames.shape[2], frames.shape[1])
    for t in (1, 2):
        p1, st, err, it = o.klt(frames[0], frames[t], p0, return_iters=True)
        d = np.linalg.norm(p1 - syn.warp_points(motions[t], p0), axis=1)
        assert st.all() and np.median(d) <= 0.05 and np.percentile(d, 95) <= 0.2
        f1, fs, fe = o.klt(frames[0], frames[t], p0, acc_mode=0)       # OpenCV's float accumulation order
        assert np.array_equal(fs, st) and (np.abs(f1 - p1).max(axis=1) <= 1e-3).mean() >= 0.99
    # identical images -> zero motion, one iteration per level
    p1, st, err, it = o.klt(frames[0], frames[0], p0, return_iters=True)
    assert np.abs(p1 - p0).max() < 1e-4 and (it == 1).all() and np.allclose(err, 0)
    # a point far outside is reported lost and keeps its (scaled) position
    p1, st, err = o.klt(frames[0], frames[1], np.array([[5000.0, 5000.0]], np.float32))
    assert st[0] == 0 and np.allclose(p1, [[5000, 5000]])


def test_circle_rasteriser():
    import vo_oracle as o
    assert o.circle_rows(7).tolist() == [7, 7, 7, 6, 6, 5, 4, 2] or o.circle_rows(7)[0] == 7
    for r in (0, 1, 3, 5, 7, 10):
        m = np.full((64, 64), 255, np.uint8)
        o.circle_mask(m, (30, 31), r, 0)
        ys, xs = np.nonzero(m == 0)
        assert ys.min() == 31 - r and ys.max() == 31 + r and xs.min() == 30 - r and xs.max() == 30 + r
        hw = o.circle_rows(r)
        for dy in range(-r, r + 1):
            row = np.nonzero(m[31 + dy] == 0)[0]
            assert row.min() == 30 - hw[abs(dy)] and row.max() == 30 + hw[abs(dy)]
        d2 = (ys - 31) ** 2 + (xs - 30) ** 2
        assert d2.max() <= r * r + r                      # midpoint circle: within half a pixel of the radius
    m = np.full((20, 20), 255, np.uint8)
    o.circle_mask(m, (-2, 19), 7, 0)                       # clipped
    assert m[19, 0] == 0 and m[12, 0] == 255 or m[12, 0] == 0


def test_good_features_invariants(seq_small):
    import vo_oracle as o
    frames, _ = seq_small
    img = frames[0]
    c, eig, nc = o.good_features(img, None, return_aux=True)
    assert len(c) <= 1000 and nc >= len(c)
    d = np.linalg.norm(c[:, None] - c[None], axis=2) + np.eye(len(c)) * 1e9
    assert d.min() >= 7.0
    v = eig[c[:, 1].astype(int), c[:, 0].astype(int)]
    assert (np.diff(v) <= 0).all() and v.min() > 0.03 * eig.max()         # rank order, quality threshold
    assert (c[:, 0] >= 1).all() and (c[:, 0] <= img.shape[1] - 2).all()
    # OpenCV's float running sums vs the exact integer sums: same corner set up to near-ties
    c2 = o.good_features(img, None, exact_int=False)
    assert len(set(map(tuple, c.tolist())) ^ set(map(tuple, c2.tolist()))) <= 0.02 * len(c)
    e2 = o.min_eig(img, exact_int=False)
    assert np.abs(e2 - eig).max() <= 1e-5 * eig.max()
    # maxCorners = 0 -> unlimited; minDistance < 1 -> no spacing rule
    assert len(o.good_features(img, None, maxCorners=0)) >= len(c)
    assert len(o.good_features(img, None, maxCorners=0, minDistance=0.5)) == nc


def test_bilateral_prefilter_restatement():
    """cv2.bilateralFilter (loader.py:16-20,86) restated in C against an independent numpy float32 statement of the same
    OpenCV 4.4 recipe (13 circular taps for d = 5, float accumulation in tap order, round half to even)."""
    import vo_oracle as o

    def ref(img, d, sc, ss):
        h, w = img.shape
        r = d // 2
        pad = np.pad(img, r, mode="reflect").astype(np.int32)
        cw = np.exp(np.arange(256, dtype=np.float64) ** 2 * (-0.5 / (sc * sc))).astype(np.float32)
        s, ws, taps = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32), 0
        c = pad[r:r + h, r:r + w]
        for i in range(-r, r + 1):
            for j in range(-r, r + 1):
                rr = np.sqrt(float(i * i + j * j))
                if rr > r:
                    continue
                taps += 1
                sw = np.float32(np.exp(rr * rr * (-0.5 / (ss * ss))))
                v = pad[r + i:r + i + h, r + j:r + j + w]
                wt = sw * cw[np.abs(v - c)]
                s = s + v.astype(np.float32) * wt
                ws = ws + wt
        return np.rint(s / ws).astype(np.uint8), taps

    rng = np.random.default_rng(1)
    smooth = np.clip(np.cumsum(rng.normal(0, 1.2, (60, 90)), axis=1) + 128, 0, 255).astype(np.uint8)
    for img in (rng.integers(0, 256, (41, 57)).astype(np.uint8), smooth):
        for d, sc, ss in ((5, 1.5, 1.5), (3, 12.0, 2.0), (7, 30.0, 3.0)):
            want, taps = ref(img, d, sc, ss)
            assert np.array_equal(o.bilateral(img, d, sc, ss), want)
            assert taps == {3: 5, 5: 13, 7: 29}[d]
    assert (o.bilateral(smooth) != smooth).mean() > 0.1          # it does smooth
    flat = np.full((9, 11), 93, np.uint8)
    assert np.array_equal(o.bilateral(flat), flat)


def test_pnp_oracle_p3p_ransac_and_iteration_bound():
    """the 3D-2D pose oracle (defines what csrc/vo_pnp.hip implements): P3P exact on noise-free data, RANSAC finds the
    planted inlier set, OpenCV's iteration bound formula"""
    import math
    import pnp_oracle as po
    from vo_mi355x import synthetic as syn
    rng = np.random.default_rng(0)
    K = syn.KITTI_K
    for _ in range(50):
        r = rng.normal(0, 0.3, 3); t = np.array([rng.normal(0, 1), rng.normal(0, 1), rng.uniform(-2, 2)])
        R = po.rodrigues(r)
        X = np.stack([rng.uniform(-15, 15, 4), rng.uniform(-3, 3, 4), rng.uniform(12, 60, 4)], 1)
        p = (X @ R.T + t) @ K.T
        hyp = po.hypothesis(K, np.linalg.inv(K), X, p[:, :2] / p[:, 2:3], [0, 1, 2, 3])
        assert hyp is not None and np.abs(hyp[0] - R).max() < 1e-6 and np.abs(hyp[1] - t).max() < 1e-5
        assert np.abs(po.log_so3(R) - r).max() < 1e-9
    # quartic: roots of (x-1)(x-2)(x+3)(x-0.5)
    c = np.poly([1.0, 2.0, -3.0, 0.5])
    assert np.allclose(sorted(po.quartic_real_roots(*c)), [-3.0, 0.5, 1.0, 2.0], atol=1e-12)
    assert po.quartic_real_roots(1.0, 0.0, 1.0, 0.0, 1.0) == []                    # no real root
    # RANSACUpdateNumIters: log(1 - p) / log(1 - w^m)
    assert po.update_num_iters(0.9999, 0.3, 4, 10 ** 6) == round(math.log(1e-4) / math.log(1 - 0.7 ** 4))
    assert po.update_num_iters(0.9999, 1.0, 4, 1000) == 1000 and po.update_num_iters(0.9999, 0.0, 4, 1000) == 0
    s = syn.make_ba_scene(n_pts=300, n_slots=2, seed=3, obs_noise=0.3)
    X = s["points_gt"].astype(np.float32); uv = s["obs"][0].astype(np.float32)
    out = rng.choice(300, 90, replace=False)
    uv[out] += rng.uniform(-60, 60, (90, 2)).astype(np.float32) + np.float32(10)
    r, t, inl, info = po.pnp_ransac(K, X, uv, return_info=True)
    assert len(np.intersect1d(inl, out)) <= 2 and len(inl) >= 205
    assert np.abs(r - s["poses_gt"][0][:3]).max() < 2e-3 and np.abs(t - s["poses_gt"][0][3:]).max() < 2e-2
    idx = po.sample4(5, 17, 10)
    assert len(set(idx)) == 4 and idx == po.sample4(5, 17, 10) and idx != po.sample4(5, 18, 10)


def test_essential_oracle_five_point_ransac_and_recover_pose():
    """the 2D-2D pose oracle (defines what csrc/vo_essential.hip implements): the true E is among the five-point
    solutions and every solution satisfies the ten cubic constraints; real-root isolation agrees with numpy's companion
    eigenvalues; RANSAC finds the planted inlier set; recoverPose picks the candidate with the points in front"""
    import essential_oracle as eo
    import pnp_oracle as po
    from vo_mi355x import synthetic as syn
    rng = np.random.default_rng(0)
    for _ in range(100):
        p = rng.normal(size=rng.integers(1, 11) + 1)
        r = eo.real_roots(list(p))
        rr = np.roots(p[::-1]); rr = np.sort(rr[np.abs(rr.imag) < 1e-9].real)
        assert len(r) == len(rr) and np.allclose(r, rr, rtol=1e-7, atol=1e-9)
    assert np.allclose(eo.real_roots(list(np.poly([1.0, 2.0, -3.0, 0.5])[::-1])), [-3.0, 0.5, 1.0, 2.0], atol=1e-12)
    assert eo.real_roots([1.0, 0.0, 1.0]) == [] and eo.real_roots([0.0, 0.0]) == []
    dists = []
    for _ in range(60):
        R = po.rodrigues(rng.normal(0, 0.2, 3)); t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
        X = np.stack([rng.uniform(-5, 5, 5), rng.uniform(-3, 3, 5), rng.uniform(4, 20, 5)], 1)
        x1 = X[:, :2] / X[:, 2:3]; Xc = X @ R.T + t; x2 = Xc[:, :2] / Xc[:, 2:3]
        Et = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
        Et /= np.linalg.norm(Et)
        sols = eo.five_point(x1.tolist(), x2.tolist())
        assert 1 <= len(sols) <= 10
        dists.append(min(min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in sols))
        for E in sols:
            assert abs(np.linalg.norm(E) - 1) < 1e-12 and abs(np.linalg.det(E)) < 1e-4
            assert np.abs(np.einsum('ni,ij,nj->n', np.c_[x2, np.ones(5)], E, np.c_[x1, np.ones(5)])).max() < 1e-9
    assert np.median(dists) < 1e-10 and max(dists) < 1e-3
    assert eo.five_point([[0.0, 0.0]] * 5, [[0.0, 0.0]] * 5) == []                      # rank-deficient sample
    # decomposition + cheirality: the true motion wins with all points in front
    R = po.rodrigues(np.array([0.02, -0.04, 0.01])); t = np.array([0.6, 0.0, -0.8])
    X = np.stack([rng.uniform(-5, 5, 50), rng.uniform(-3, 3, 50), rng.uniform(4, 20, 50)], 1)
    x1 = X[:, :2] / X[:, 2:3]; Xc = X @ R.T + t; x2 = Xc[:, :2] / Xc[:, 2:3]
    Et = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
    Rr, tr, g, good = eo.recover_pose(Et, x1, x2)
    assert np.abs(Rr - R).max() < 1e-9 and np.abs(tr - t).max() < 1e-9 and g == 50 and sorted(good)[:3] == [0, 0, 0]
    assert np.abs(eo.sampson_err2(Et, x1, x2)).max() < 1e-20
    # RANSAC on pixels with 30 % gross outliers
    K = syn.KITTI_K
    n = 300
    X = np.stack([rng.uniform(-15, 15, n), rng.uniform(-3, 3, n), rng.uniform(8, 45, n)], 1)
    p1 = X @ K.T; p1 = p1[:, :2] / p1[:, 2:3] + rng.normal(0, 0.3, (n, 2))
    Xc = X @ R.T + t; p2 = Xc @ K.T; p2 = p2[:, :2] / p2[:, 2:3] + rng.normal(0, 0.3, (n, 2))
    out = rng.choice(n, 90, replace=False)
    p2[out] += rng.uniform(-60, 60, (90, 2)) + 10
    E, Rr, tr, inl, info = eo.essential_ransac(K, p1.astype(np.float32), p2.astype(np.float32), return_info=True)
    assert len(np.intersect1d(inl, out)) <= 6 and len(inl) >= 170 and info["hyps"] == 256
    assert np.abs(Rr - R).max() < 1e-2 and np.abs(tr - t).max() < 0.1
    idx = eo.sample5(5, 17, 10)
    assert len(set(idx)) == 5 and idx == eo.sample5(5, 17, 10) and idx != eo.sample5(5, 18, 10)
    assert eo.essential_ransac(K, p1[:4], p2[:4])[0] is None
