"""GPU parity: bundle adjustment through the C ABI vs the reference (golden G1) and the numpy oracle."""
import glob

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(path):
    from helpers import golden_ba_problem, ref_stub_cv2
    cv2 = ref_stub_cv2()   # Rodrigues only
    g = np.load(path)
    K, poses, points, obs, tags = golden_ba_problem(g, lambda R: cv2.Rodrigues(R)[0])
    return g, K, poses, points, obs


@pytest.fixture(autouse=True, params=["wave_private", "lane_per_observation"])
def ba_kernels(request, monkeypatch, ctx):
    """every test of this module runs through BOTH kernel families: the wave-private build / update (csrc/vo_ba_wave.h: what a batched context
    runs for windows <= 10) and the lane-per-observation ones (windows of 11-20 slots; vo_tuning.ba_kernels = 1 selects them for every window).
    The field takes effect at every upload: set on the module's shared context and as the default of every context a test makes."""
    from vo_mi355x import VoContext
    fam = 2 if request.param == "wave_private" else 1
    monkeypatch.setattr(VoContext, "default_tuning", {"ba_kernels": fam})
    ctx.set_tuning(ba_kernels=fam)
    return request.param


@pytest.fixture(scope="module")
def ctx():
    from vo_mi355x import VoContext
    c = VoContext(64, 64, max_pts=64)
    yield c
    c.close()


def _goldens(golden_dir):
    return sorted(glob.glob(golden_dir + "/ba_*.npz"))


def test_ba_residual_matches_reference(ctx, golden_dir):
    """BA-1: residual vector at x0 in the reference's order == the reference's own output (1e-9 px)"""
    for path in _goldens(golden_dir):
        g, K, poses, points, obs = _load(path)
        ctx.ba_upload(K, poses, points, obs)
        pr = ctx.ba_probe(lam=1e-4)
        assert len(pr["residual"]) == len(g["r0"])
        assert np.abs(pr["residual"] - g["r0"]).max() <= 1e-9
        import ba_oracle as bo
        assert abs(pr["cost"] - 0.5 * bo.huber_rho(g["r0"] ** 2).sum()) <= 1e-9 * max(1.0, pr["cost"])


def test_ba_normal_equations_and_step_match_oracle(ctx, golden_dir):
    """BA-3/4/5: Huber-weighted J^T J blocks, gradient, reduced camera system and one LM step vs float64 numpy"""
    import ba_oracle as bo
    for path in _goldens(golden_dir):
        g, K, poses, points, obs = _load(path)
        lam = 1e-3
        ctx.ba_upload(K, poses, points, obs)
        pr = ctx.ba_probe(lam=lam)
        ne = bo.normal_equations(K, poses, points, obs)

        def rel(a, b):
            return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
        assert rel(pr["Hpp"], ne["Hpp"]) <= 1e-10
        assert rel(pr["gp"], ne["gp"]) <= 1e-10
        assert rel(pr["Hll"], ne["Hll"]) <= 1e-10
        assert rel(pr["gl"], ne["gl"]) <= 1e-10
        S, rhs, _, _, _ = bo.schur_system(ne, lam)
        assert rel(pr["S"], S) <= 1e-8, rel(pr["S"], S)
        assert rel(pr["rhs"], rhs) <= 1e-8
        dp, dl, _ = bo.lm_step(ne, lam)
        assert rel(pr["dposes"], dp) <= 1e-6, rel(pr["dposes"], dp)
        assert rel(pr["dpoints"], dl) <= 1e-6, rel(pr["dpoints"], dl)


def test_ba_jacobian_vs_reference_finite_differences(golden_dir):
    """BA-3: analytic blocks (as used on the GPU, via the oracle twin) vs the reference's FD Jacobian"""
    import ba_oracle as bo
    path = _goldens(golden_dir)[0]
    g, K, poses, points, obs = _load(path)
    J = bo.dense_jacobian_norm_form(K, poses, points, obs)
    Jfd = np.zeros_like(J)
    Jfd[g["Jfd_row"], g["Jfd_col"]] = g["Jfd_val"]
    big = np.abs(J) > 1.0
    assert (np.abs(J - Jfd)[big] / np.abs(J)[big]).max() <= 5e-3


def test_ba_solution_vs_oracle_and_reference(ctx, golden_dir):
    """BA-6/7: same iterates as the numpy LM; final cost <= the reference's (default and near-converged runs)"""
    import ba_oracle as bo
    for path in _goldens(golden_dir):
        g, K, poses, points, obs = _load(path)
        prm = ctx.ba_params(max_iters=50, ftol=1e-3, xtol=1e-3)
        po, pt, st = ctx.ba_adjust(K, poses, points, obs, prm)
        ref = bo.solve(K, poses, points, obs, max_iters=50, ftol=1e-3, xtol=1e-3)
        assert st["iters"] == ref["iters"] and st["accepted"] == ref["accepted"] and st["status"] == ref["status"]
        assert abs(st["cost"] - ref["cost"]) <= 1e-7 * ref["cost"]
        assert abs(st["cost0"] - ref["cost0"]) <= 1e-9 * ref["cost0"]
        assert np.abs(po - ref["poses"]).max() <= 1e-6 and np.abs(pt - ref["points"]).max() <= 1e-5
        assert abs(bo.cost(K, po, pt, obs) - st["cost"]) <= 1e-9 * st["cost"]
        # BA-7: at the reference's tolerances the build must not be worse than the reference's scipy run
        assert st["cost"] <= float(g["ref_cost"]) * (1 + 1e-3)
        # tight run: at least as low as the reference's near-converged cost
        prm = ctx.ba_params(max_iters=200, ftol=1e-12, xtol=1e-12)
        po2, pt2, st2 = ctx.ba_adjust(K, poses, points, obs, prm)
        assert st2["cost"] <= float(g["tight_cost"]) * (1 + 1e-4)


@pytest.mark.parametrize("name", ["ba_s0_n64_w4", "ba_s1_n64_w4", "ba_s2_n256_w10", "ba_s0_n256_w10", "bafull_s0_n2000_w10"])
def test_ba_pose_deltas_and_points_vs_reference_solutions(ctx, golden_dir, name):
    """BA-6 / BA-7 (SURVEY 8a'): `vo_ba_adjust` poses and points, gauge removed (pose deltas relative to the newest frame,
    window baseline = 1), against the reference's own solutions of `BundleAdjuster.adjust` (bundle_adjuster.py:189-213):
    * the run to stagnation lands on the point the reference's solver itself accepts as a minimum (warm-started there it
      hands the point back): rotations <= 1e-4 rad, translations <= 1e-3 baseline, points <= 1e-3 of their depth;
    * at the reference's tolerances (xtol = ftol = 1e-3) cost <= the reference's, pose deltas and points at least as close
      to that minimum as the reference's own answer.  Incl. the BASELINE shape: 2000 landmarks, 10-frame window."""
    from helpers import ba_solution_parity
    g, K, poses, points, obs = _load("%s/%s.npz" % (golden_dir, name))
    gp = np.load("%s/%s.npz" % (golden_dir, name.replace("bafull_", "bapolish_").replace("ba_", "bapolish_")))

    def solve(max_iters, ftol, xtol):
        po, pt, st = ctx.ba_adjust(K, poses, points, obs, ctx.ba_params(max_iters=max_iters, ftol=ftol, xtol=xtol))
        return po, pt, st["cost"]
    rep = ba_solution_parity(solve, g, gp, K, poses, points, obs)
    print(name, {k: (tuple(float("%.3g" % x) for x in v) if isinstance(v, tuple) else float("%.6g" % v)) for k, v in rep.items()})


def test_ba_workspace_survives_a_smaller_problem_with_more_partials():
    """One context, W = 4: 2600 landmarks run as 41 workgroups of 64 landmarks, 2500 as 157 of 16 -- the second adjust needs
    MORE per-workgroup partial sets than the first allocated (the drop-in BundleAdjuster.adjust changes N every frame)."""
    import ba_oracle as bo
    from vo_mi355x import VoContext, synthetic as syn
    with VoContext(64, 64, max_pts=64) as c:
        prm = c.ba_params(max_iters=6)
        for n in (2600, 2500, 2600, 300):
            s = syn.make_ba_scene(n_pts=n, n_slots=4, seed=n)
            po, pt, st = c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], prm)
            ref = bo.solve(s["K"], s["poses0"], s["points0"], s["obs"], max_iters=6)
            assert st["iters"] == ref["iters"] and abs(st["cost"] - ref["cost"]) <= 1e-7 * ref["cost"], n
            assert np.abs(po - ref["poses"]).max() <= 1e-6


def test_ba_resident_repeatable(ctx):
    from vo_mi355x import synthetic as syn
    s = syn.make_ba_scene(n_pts=500, n_slots=10, seed=3, visibility=0.8)
    prm = ctx.ba_params(max_iters=8)
    ctx.ba_upload(s["K"], s["poses0"], s["points0"], s["obs"])
    outs = []
    for _ in range(2):
        ctx.ba_solve_resident(prm)
        outs.append(ctx.ba_fetch())
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])   # bitwise reproducible
    po, pt, st = ctx.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], prm)
    assert np.array_equal(po, outs[0][0]) and st["cost"] == outs[0][2]["cost"]
    assert st["cost"] < 0.05 * st["cost0"]


def test_ba_full_size_properties(ctx):
    """BASELINE shape (N = 2000, W = 10): gauge-invariant checks (cost decrease, reprojection RMS, gauge invariance)"""
    import ba_oracle as bo
    from vo_mi355x import synthetic as syn
    s = syn.make_ba_scene(n_pts=2000, n_slots=10, seed=0)
    po, pt, st = ctx.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], ctx.ba_params(max_iters=30))
    assert st["n_obs"] == 20000 and st["cost"] < 0.02 * st["cost0"]
    r = bo.residual_norm(s["K"], po, pt, s["obs"])
    assert np.sqrt((r ** 2).mean()) < 0.5          # observation noise is 0.3 px per axis
    ref = bo.solve(s["K"], s["poses0"], s["points0"], s["obs"], max_iters=30)
    assert abs(st["cost"] - ref["cost"]) <= 1e-6 * ref["cost"]


def test_ba_bank_select_equals_individual_uploads():
    """a bank of resident problems (vo_ba_upload_bank / vo_ba_select_problem): solving problem k of the bank = uploading and solving it alone,
    bit for bit, in any order, on a batch of 2; a fetch while a frame step is in flight does not disturb that step's results"""
    from vo_mi355x import VoContext, synthetic as syn
    B, nb = 2, 3
    sc = [[syn.make_ba_scene(n_pts=300, n_slots=6, seed=10 * b + k, pt_noise=0.3 + 0.3 * k, visibility=1.0 - 0.1 * k) for k in range(nb)] for b in range(B)]
    K = np.stack([sc[b][0]["K"] for b in range(B)])
    with VoContext(64, 64, max_pts=64, batch=B) as c:
        prm = c.ba_params(max_iters=12)
        alone = []
        for k in range(nb):
            c.ba_upload(K, np.stack([sc[b][k]["poses0"] for b in range(B)]), np.stack([sc[b][k]["points0"] for b in range(B)]),
                        np.stack([sc[b][k]["obs"] for b in range(B)]))
            c.ba_solve_resident(prm)
            alone.append(c.ba_fetch())
        c.ba_upload_bank(K, np.stack([[sc[b][k]["poses0"] for b in range(B)] for k in range(nb)]),
                         np.stack([[sc[b][k]["points0"] for b in range(B)] for k in range(nb)]),
                         np.stack([[sc[b][k]["obs"] for b in range(B)] for k in range(nb)]))
        for k in (2, 0, 1, 2):
            c.ba_select(k)
            c.ba_solve_resident(prm)
            po, pt, st = c.ba_fetch()
            assert np.array_equal(po, alone[k][0]) and np.array_equal(pt, alone[k][1])
            assert [x["iters"] for x in st] == [x["iters"] for x in alone[k][2]] and [x["cost"] for x in st] == [x["cost"] for x in alone[k][2]]
        its = [x["iters"] for x in alone[0][2]] + [x["iters"] for x in alone[2][2]]
        assert len(set(its)) > 1                      # the problems really differ in difficulty
        with pytest.raises(Exception):
            c.ba_select(nb)


def test_ba_chunked_partial_sets_match_one_chunk_per_workgroup(monkeypatch):
    """The headline batch (32 x 125 landmark chunks) makes a workgroup of k_ba_build walk up to 4 chunks and ADD their partial sums into
    one set; a single problem keeps one chunk per workgroup.  vo_tuning.ba_chunks forces either form: same LM iteration / acceptance sequence,
    cost and solution to 1e-12 (the summation order differs, so not bit for bit), also for the last, partly filled workgroup
    (125 chunks = 31 x 4 + 1), and equal to the oracle."""
    import ba_oracle as bo
    from vo_mi355x import VoContext, synthetic as syn
    s = syn.make_ba_scene(n_pts=2000, n_slots=10, seed=3, visibility=0.9)
    ref = bo.solve(s["K"], s["poses0"], s["points0"], s["obs"], max_iters=12)
    out = {}
    with VoContext(64, 64, max_pts=64) as c:
        for cpw in (1, 2, 4, 3):
            c.set_tuning(ba_chunks=cpw)
            out[cpw] = c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c.ba_params(max_iters=12))
        c.set_tuning(ba_chunks=0)
        out[0] = c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c.ba_params(max_iters=12))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])          # the rule: one chunk per workgroup for one problem
    for cpw in (2, 3, 4):
        po, pt, st = out[cpw]
        assert st["iters"] == out[1][2]["iters"] == ref["iters"] and st["accepted"] == out[1][2]["accepted"]
        assert abs(st["cost"] - out[1][2]["cost"]) <= 1e-12 * st["cost"] and abs(st["cost"] - ref["cost"]) <= 1e-7 * ref["cost"]
        assert np.abs(po - out[1][0]).max() <= 1e-10 and np.abs(pt - out[1][1]).max() <= 1e-9


@pytest.mark.parametrize("W,n_pts", [(4, 2000), (8, 1000), (3, 333), (5, 700)])
def test_ba_eight_lanes_per_landmark_matches_oracle_and_the_sixteen_lane_form(monkeypatch, W, n_pts):
    """Windows of <= 8 slots run k_ba_build<256, 8> / k_ba_update<256, 8> (a landmark owns 8 lanes, 32 landmarks per 256-lane workgroup);
    vo_tuning.ba_lanes = 16 forces the 16-lane form.  Same LM iteration / acceptance sequence, cost and solution to 1e-10 between the two (the summation
    order differs), equal to the oracle; also with a workgroup walking several landmark chunks (vo_tuning.ba_chunks) and N not a multiple of 32."""
    import ba_oracle as bo
    from vo_mi355x import VoContext, synthetic as syn
    s = syn.make_ba_scene(n_pts=n_pts, n_slots=W, seed=10 + W, visibility=0.85)
    ref = bo.solve(s["K"], s["poses0"], s["points0"], s["obs"], max_iters=12)
    out = {}
    for key, tune in (("l8", {}), ("l16", {"ba_lanes": 16}), ("l8c3", {"ba_chunks": 3})):
        with VoContext(64, 64, max_pts=64) as c:          # (the lane count is fixed when the workspace is built: a fresh context per form)
            c.set_tuning(**tune)
            out[key] = c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c.ba_params(max_iters=12))
    for key in ("l8", "l16", "l8c3"):
        po, pt, st = out[key]
        assert st["iters"] == ref["iters"] and st["accepted"] == ref["accepted"], (key, st, ref["iters"], ref["accepted"])
        assert abs(st["cost"] - ref["cost"]) <= 1e-7 * ref["cost"], (key, st["cost"], ref["cost"])
        assert abs(st["cost"] - out["l16"][2]["cost"]) <= 1e-10 * st["cost"]
        assert np.abs(po - out["l16"][0]).max() <= 1e-9 and np.abs(pt - out["l16"][1]).max() <= 1e-8
