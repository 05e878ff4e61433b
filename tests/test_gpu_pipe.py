"""Device-resident Pipeline.step (C ABI vo_pipe_*, csrc/vo_pipeline.hip) against the reference's loop over Python objects
(pipeline.py:92-167 restated over the drop-in classes, tests/pipe_helpers.ObjectLoop, every numerical call on the same GPU):
frame by frame the candidate / landmark / dead lists object by object (birth frame, track length, float32 pixel positions,
histories, t_latest exact; landmark positions and the trajectory to 1e-7), including the reference's resurrection of recently
dead landmarks into the window and the object sharing that follows from it; then the reference's own glue goldens (G1 adjust,
G3 extractor glue: tests/golden/gen_golden.py) replayed stage by stage through the device tables."""
import copy

import numpy as np
import pytest

import pipe_helpers as ph
from helpers import golden_tracks

pytestmark = pytest.mark.gpu


def _ctx(w, h, max_pts=2048, batch=1):
    from vo_mi355x import VoContext
    return VoContext(w, h, max_pts=max_pts, batch=batch)


def _sharing(e):
    dl, dk = {}, {}
    for i, (l, k) in enumerate(zip(e["rows"]["dead_l"], e["rows"]["dead_k"])):
        dl.setdefault(int(l), i); dk.setdefault(int(k), i)
    return [(dl.get(int(l), -1) >= 0, dk.get(int(k), -1) >= 0) for l, k in zip(e["rows"]["lm_l"], e["rows"]["lm_k"])]


@pytest.mark.parametrize("ba_window,n_steps,seed,period,amp", [(4, 12, 2024, 24.0, (0.9, 0.25, -0.5)), (10, 10, 2024, 24.0, (0.9, 0.25, -0.5)),
                                                              (4, 30, 7, 16.0, (1.6, 0.4, -0.8)), (6, 16, 99, 20.0, (1.2, -0.3, 0.6))])
def test_resident_pipeline_equals_object_loop(ba_window, n_steps, seed, period, amp):
    """the last two: other scenes, faster motion (more tracks leave the image, more deaths and resurrections), a longer run (the history
    ring wraps at 32 entries per keypoint only for tracks older than the run: 30 frames stay inside it)"""
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1 = 256, 160, 3
    sc = ph.scene(t1 + n_steps + 1, w=w, h=h, f=260.0, seed=seed, pose_fn=lambda t: ph.sway_pose(t, amp=amp, period=period))
    ctx_a, ctx_b = _ctx(w, h), _ctx(w, h)
    state, t_loader = ph.gt_bootstrap(ctx_a, sc, 0, t1)
    loop = ph.ObjectLoop(ctx_a, sc["K"], copy.deepcopy(state), sc["frames"][t_loader], ba_window=ba_window, ba_max_iters=16)
    rp = ResidentPipeline(ctx_b, sc["K"], ba_window=ba_window, ba_max_iters=16, pnp_blind_batches=8)
    rp.seed(state, [], [], t_step=1)
    ctx_b.push_frame(sc["frames"][t_loader])
    ph.compare_lists(loop, rp.entries(), what="seed")
    seen = dict(res=0, new=0, shared=0, dup=0)
    final = {}
    for s in range(n_steps):
        im = sc["frames"][t_loader + 1 + s]
        loop.step(im)
        ctx_b.push_frame(im)
        rp.step()
        rec = rp.fetch()
        what = "step %d" % (s + 2)
        assert rec["status"] == 0 and rec["overflow"] == 0 and rec["t"] == loop.t_step, (what, rec)
        e = rp.entries()
        ph.compare_lists(loop, e, what=what, p_tol=1e-7)
        st = loop.state
        assert (rec["n_landmarks"], rec["n_candidates"], rec["n_dead_total"]) == (len(st._landmarks), len(st._candidates_kp), len(loop.dead)), what
        assert (rec["n_new"], rec["n_resurrected"], rec["n_detected"], rec["pnp_inliers"]) == \
               (loop.info["n_new"], loop.info["n_resurrected"], loop.info["n_detected"], loop.info["n_inliers"]), (what, rec, loop.info)
        assert rec["pnp_bound_reached"] == 1 and rec["ba_done"] == 1
        if loop.info["ba"] is not None:
            assert rec["ba_iters"] == loop.info["ba"]["iters"] and abs(rec["ba_cost"] - loop.info["ba"]["cost"]) <= 1e-7 * loop.info["ba"]["cost"]
        for t in range(loop.t_step + 1):
            assert np.abs(e["poses"][t] - st._trajectory[t]).max() <= 1e-7, (what, t)
        assert np.abs(rec["H"] - st._trajectory[loop.t_step]).max() <= 1e-7
        assert rec["t_final"] == max(-1, loop.t_step - (ba_window - 1))
        if rec["t_final"] >= 0:
            final[rec["t_final"]] = rec["H_final"]
        sl = ph.sharing_signature(loop)
        assert [(a >= 0, b >= 0) for a, b in sl] == _sharing(e), what
        seen["res"] += rec["n_resurrected"]; seen["new"] += rec["n_new"]
        seen["shared"] += sum(1 for a, b in sl if a >= 0 and b < 0)
        ids = [id(l) for l in st._landmarks]
        seen["dup"] += len(ids) - len(set(ids))
    assert seen["res"] > 0 and seen["new"] > 0 and seen["shared"] > 0, seen
    if ba_window > 4:
        assert seen["dup"] > 0, seen
    # the poses collected as they left the window ARE the reference's final trajectory for those steps (no later adjust touches them)
    assert final and all(np.abs(H - loop.state._trajectory[t]).max() <= 1e-7 for t, H in final.items())
    # tables -> objects: the reference's classes with the sharing restored
    st2, dead2, dead_kp2 = rp.objects()
    assert len(st2._landmarks) == len(loop.state._landmarks) and len({id(l) for l in st2._landmarks}) == len({id(l) for l in loop.state._landmarks})


def test_steps_in_flight_and_batch_equal_one_at_a_time():
    """4 steps enqueued before the first fetch, on a batch of 3 sequences (two of them the same scene): equal to stepping and fetching
    one frame at a time; batch entries are independent"""
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1, n = 256, 160, 3, 8
    scs = [ph.scene(t1 + n + 1, w=w, h=h, f=260.0, seed=sd, pose_fn=lambda t: ph.sway_pose(t, period=24.0)) for sd in (2024, 77)]
    order = [0, 1, 0]
    c1 = _ctx(w, h)
    states = [ph.gt_bootstrap(c1, sc, 0, t1)[0] for sc in scs]
    ref = []
    for sc, st in zip(scs, states):
        rp = ResidentPipeline(c1, sc["K"], ba_max_iters=12)
        rp.seed(copy.deepcopy(st), [], [], 1)
        c1.push_frame(sc["frames"][t1])
        recs = []
        for s in range(n):
            c1.push_frame(sc["frames"][t1 + 1 + s]); rp.step(); recs.append(rp.fetch())
        ref.append((recs, rp.entries()))
    cb = _ctx(w, h, batch=3)
    cb.upload_sequence(np.stack([scs[i]["frames"] for i in order]))
    rpb = ResidentPipeline(cb, np.stack([scs[i]["K"] for i in order]), ba_max_iters=12)
    rpb.seed([copy.deepcopy(states[i]) for i in order], None, None, 1)
    cb.push_frame_resident(t1)
    got = []
    for s0 in range(0, n, 4):
        for s in range(s0, s0 + 4):
            rpb.step(t1 + 1 + s)
        for s in range(s0, s0 + 4):
            got.append(rpb.fetch())
    T = rpb.read_tables()
    for b, i in enumerate(order):
        recs, ent = ref[i]
        for s in range(n):
            for k in recs[s]:
                if k in ("H", "H_final"):
                    assert np.array_equal(got[s][b][k], recs[s][k]), (b, s, k)
                else:
                    assert got[s][b][k] == recs[s][k], (b, s, k, got[s][b][k], recs[s][k])
        eb = rpb.entries(b, T)
        for name in ("cand", "lm", "dead"):
            assert len(eb[name]) == len(ent[name])
            for x, y in zip(eb[name], ent[name]):
                assert x[0] == y[0] and (x[1] is None or np.array_equal(x[1], y[1])) and x[2:4] == y[2:4] and np.array_equal(x[5], y[5]) and np.array_equal(x[7], y[7])


def test_capacity_policy_matches_the_model():
    """a table too small for the scene: detections / promotions / resurrections are cut in list order exactly as oracle/pipe_oracle.py
    defines, and the record says so"""
    import pipe_oracle as po
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1, n = 256, 160, 3, 8
    cap = 160
    sc = ph.scene(t1 + n + 1, w=w, h=h, f=260.0, seed=2024, pose_fn=lambda t: ph.sway_pose(t, period=24.0))
    ctx_a, ctx_b = _ctx(w, h), _ctx(w, h, max_pts=cap)
    state, _ = ph.gt_bootstrap(ctx_a, sc, 0, t1, n_landmarks=80)
    state._candidates_kp = state._candidates_kp[:40]
    model = po.PipeModel(ctx_a, sc["K"], w, h, cap=cap, params=po.Params(ba_window=10, ba_max_iters=12))
    model.seed(copy.deepcopy(state), [], [], 1)
    ctx_a.push_frame(sc["frames"][t1])
    rp = ResidentPipeline(ctx_b, sc["K"], ba_window=10, ba_max_iters=12, pnp_blind_batches=8)
    rp.seed(state, [], [], 1)
    ctx_b.push_frame(sc["frames"][t1])
    any_overflow = 0
    for s in range(n):
        im = sc["frames"][t1 + 1 + s]
        model.step(im)
        ctx_b.push_frame(im); rp.step(); rec = rp.fetch()
        assert rec["status"] == 0 and model.status == 0
        assert rec["overflow"] == model.info.get("overflow", 0), (s, rec["overflow"], model.info)
        any_overflow |= rec["overflow"]
        e = rp.entries()
        assert (len(e["cand"]), len(e["lm"]), len(e["dead"]), e["n_dead_total"]) == (len(model.cand), len(model.lm_L), len(model.dead_L),
                                                                                    len(model.dead_L) + model.n_dead_inert), s
        for (l, k), x in zip(zip(model.lm_L, model.lm_K), e["lm"]):
            y = model.entry(l, k)
            assert x[0] == y[0] and x[2:4] == y[2:4] and np.array_equal(x[5], y[5]) and np.linalg.norm(x[1] - y[1]) <= 1e-7 * np.linalg.norm(y[1])
    assert any_overflow & 8 and any_overflow & (2 | 4), any_overflow


# ---------------------------------------------------------------------------------------------------------
# the reference's glue goldens through the device tables, stage by stage
# ---------------------------------------------------------------------------------------------------------
def _kps(g, prefix):
    from vo_mi355x import Keypoint
    lens = g[prefix + "_hist_len"]
    off = np.concatenate([[0], np.cumsum(lens)])
    return [Keypoint(int(g[prefix + "_t_first"][i]), int(g[prefix + "_t_total"][i]), g[prefix + "_uv_first"][i].astype(np.float32).reshape(2, 1),
                     g[prefix + "_uv"][i].astype(np.float32).reshape(2, 1), np.array([[g[prefix + "_tag"][i]]]),
                     [g[prefix + "_hist"][k].astype(np.float32).reshape(2, 1) for k in range(off[i], off[i + 1])]) for i in range(len(lens))]


def _check_entries(entries, g, prefix, with_l=False, p_tol=0.0):
    assert len(entries) == len(g[prefix + "_t_first"]), (prefix, len(entries), len(g[prefix + "_t_first"]))
    if not entries:
        return
    assert np.array_equal(np.array([np.float64(e[5]) for e in entries]), g[prefix + "_uv"]), prefix
    assert np.array_equal(np.array([np.float64(e[4]) for e in entries]), g[prefix + "_uv_first"])
    assert np.array_equal([e[2] for e in entries], g[prefix + "_t_first"]) and np.array_equal([e[3] for e in entries], g[prefix + "_t_total"])
    assert np.array_equal([e[6] for e in entries], g[prefix + "_hist_len"])
    assert np.array_equal(np.concatenate([np.float64(e[7]) for e in entries]), g[prefix + "_hist"])


def test_extractor_glue_golden_through_the_tables(golden_dir):
    """G3: extend_tracks x2, extend_landmarks, extract with exclusion discs, triangulate_tracks (two birth groups, length gate, filters,
    bearing gate) -- the reference's own outputs (tests/golden/glue_s0.npz)"""
    from vo_mi355x import Landmark, State, Trajectory
    from vo_mi355x.resident import ResidentPipeline, TRACK, DETECT, TRIANGULATE
    g = np.load(golden_dir + "/glue_s0.npz")
    frames = g["frames"]
    h, w = frames.shape[1:]
    ctx = _ctx(w, h)
    K = g["tt_K"]
    rp = ResidentPipeline(ctx, K)
    cands = _kps(g, "in0")
    # extend_tracks: frames 0 -> 1, then 1 -> 2
    rp.seed(State([], [], copy.deepcopy(cands), Trajectory({0: np.eye(4)})), [], [], t_step=0)
    ctx.push_frame(frames[0]); ctx.push_frame(frames[1])
    rp.step(-1, TRACK); rp.fetch()
    _check_entries(rp.entries()["cand"], g, "tr1")
    ctx.push_frame(frames[2])
    rp.step(-1, TRACK); rp.fetch()
    _check_entries(rp.entries()["cand"], g, "tr2")
    # extend_landmarks on frame 1 -> 2: survivors (t_latest + 1) and the dead lists
    c1 = _kps(g, "tr1")
    rng = np.random.default_rng(0)
    for _ in range(6):
        rng.uniform(5, h - 5)                            # keep the generator in step with gen_golden.py
    lms = [Landmark(2, rng.normal(0, 1, (3, 1)), k.des.copy()) for k in c1]
    ctx2 = _ctx(w, h)
    rp2 = ResidentPipeline(ctx2, K)
    rp2.seed(State(lms, copy.deepcopy(c1), [], Trajectory({0: np.eye(4)})), [], [], t_step=2)
    ctx2.push_frame(frames[1]); ctx2.push_frame(frames[2])
    rp2.step(-1, TRACK); rec = rp2.fetch()
    e = rp2.entries()
    _check_entries(e["lm"], g, "el_k"); _check_entries(e["dead"], g, "el_kd")
    assert np.array_equal([x[0] for x in e["lm"]], g["el_l_t_latest"]) and np.array_equal([x[0] for x in e["dead"]], g["el_ld_t_latest"])
    assert np.array_equal(np.array([x[1] for x in e["lm"]]), g["el_l_p"]) and np.array_equal(np.array([x[1] for x in e["dead"]]), g["el_ld_p"])
    assert rec["n_dead_total"] == len(g["el_ld_t_latest"]) > 0
    # extract: exclusion discs at the tracked keypoints, corners appended as candidates born at t = 3
    c2 = _kps(g, "tr2")
    ctx3 = _ctx(w, h)
    rp3 = ResidentPipeline(ctx3, K)
    rp3.seed(State([], [], copy.deepcopy(c2), Trajectory({0: np.eye(4)})), [], [], t_step=3)
    ctx3.push_frame(frames[2])
    rp3.step(-1, DETECT); rec = rp3.fetch()
    e = rp3.entries()
    assert rec["n_detected"] == len(g["ex2_t_first"]) > 0
    _check_entries(e["cand"][len(c2):], g, "ex2")
    _check_entries(e["cand"][:len(c2)], g, "tr2")
    # triangulate_tracks
    traj = Trajectory({t: H for t, H in enumerate(g["tt_traj"])})
    ctx4 = _ctx(w, h)
    rp4 = ResidentPipeline(ctx4, K)
    rp4.seed(State([], [], _kps(g, "tt_in"), traj), [], [], t_step=len(g["tt_traj"]) - 1)
    rp4.step(-1, TRIANGULATE); rec = rp4.fetch()
    e = rp4.entries()
    assert rec["status"] == 0 and rec["n_new"] == len(g["tt_l_t_latest"]) > 0
    _check_entries(e["lm"], g, "tt_k"); _check_entries(e["cand"], g, "tt_rest")
    assert np.array_equal([x[0] for x in e["lm"]], g["tt_l_t_latest"])
    P = np.array([x[1] for x in e["lm"]])
    assert (np.linalg.norm(P - g["tt_l_p"], axis=1) <= 1e-4 * np.linalg.norm(g["tt_l_p"], axis=1)).all()


@pytest.mark.parametrize("name", ["ba_s0_n64_w4", "ba_s2_n256_w10"])
def test_adjust_golden_through_the_tables(golden_dir, name):
    """G1: BundleAdjuster.adjust on the reference's own inputs -- which dead landmarks come back into the state's lists, the order
    of the dead list afterwards, x0 and the observation table (through the residual vector the reference's objective returns at x0),
    and the solution equal to the drop-in class on the same device"""
    from vo_mi355x import BundleAdjuster, Keypoint, Landmark, State, Trajectory
    from vo_mi355x.resident import ResidentPipeline, ADJUST
    g = np.load("%s/%s.npz" % (golden_dir, name))
    W, t_now, K = int(g["W"]), int(g["t_now"]), g["K"]

    def mk(prefix):
        ls, ks = [], []
        for r in golden_tracks(g, prefix):
            # pixel positions are float32 in the pipeline (KLT output); the golden's synthetic observations are float64 noise:
            # the tables hold them rounded, so does the comparison
            hist = [np.float32(h).astype(np.float64).reshape(2, 1) for h in r["hist"]]
            ks.append(Keypoint(r["t_latest"] - len(hist) + 1, len(hist), hist[0], hist[-1], np.array([[r["tag"]]]), hist))
            ls.append(Landmark(r["t_latest"], r["p"].reshape(3, 1), np.array([[r["tag"]]])))
        return ls, ks
    al, ak = mk("act_"); dl, dk = mk("dead_")
    traj = Trajectory({t: H.copy() for t, H in enumerate(g["traj"])})
    state = State(al, ak, [], traj)
    import ba_oracle as bo
    poses_r, points_r, obs_r, _, _, _ = BundleAdjuster(window_size=W).build_problem(copy.deepcopy(state), copy.deepcopy(dl), copy.deepcopy(dk), t_now)
    r0_rounded = bo.residual_norm(K, poses_r, points_r, obs_r)
    assert np.array_equal(bo.pack_x0(poses_r, points_r), g["x0"]) and np.abs(r0_rounded - g["r0"]).max() <= 1e-4
    tags = [float(l.des.reshape(-1)[0]) for l in al + dl]           # table rows are dealt in this order by seed()
    ctx = _ctx(64, 64, max_pts=512)
    # (a) budget 0: the problem as built on the device, probed at x0
    rp = ResidentPipeline(ctx, K, ba_window=W, ba_max_iters=50, ba_budget=0)
    rp.seed(copy.deepcopy(state), copy.deepcopy(dl), copy.deepcopy(dk), t_step=t_now)
    rp.step(-1, ADJUST); rec = rp.fetch()
    e = rp.entries()
    assert rec["n_landmarks"] == int(g["ref_n_state_landmarks"]) and rec["n_dead_total"] == len(g["ref_dead_tags"])
    assert np.array_equal([tags[int(i)] for i in e["rows"]["lm_l"]], g["refine_tags"])
    dead_tags = [tags[int(i)] for i in e["rows"]["dead_l"]]
    n_res = rec["n_resurrected"]
    assert dead_tags[:n_res] == list(g["ref_dead_tags"])[:n_res]        # the resurrected ones lead the dead list, in their old order
    it = iter(list(g["ref_dead_tags"]))
    assert all(any(x == y for y in it) for x in dead_tags)              # the rest follows in order; inert entries are only counted
    assert rec["n_resurrected"] == int(g["ref_n_state_landmarks"]) - len(al) > 0
    r0 = ctx.ba_probe(lam=1e-4)["residual"]
    assert len(r0) == len(g["r0"]) and np.abs(r0 - r0_rounded).max() <= 1e-9 and np.abs(r0 - g["r0"]).max() <= 1e-4
    assert rec["ba_observations"] == len(g["r0"])
    # (b) the solve: same result as the drop-in BundleAdjuster on the same device
    ba = BundleAdjuster(verbosity=0, window_size=W, method='trf', xtol=1e-3, ftol=1e-3, ctx=_ctx(64, 64, max_pts=512), max_iters=50)
    s2, dl2, dk2 = ba.adjust(copy.deepcopy(state), copy.deepcopy(dl), copy.deepcopy(dk), K, t_now)
    rp.set_ba_budget(50)
    rp.seed(copy.deepcopy(state), copy.deepcopy(dl), copy.deepcopy(dk), t_step=t_now)
    rp.step(-1, ADJUST); rec = rp.fetch()
    e = rp.entries()
    assert rec["ba_done"] == 1 and rec["ba_iters"] == ba.last_stats["iters"] and rec["ba_cost"] <= float(g["ref_cost"]) * (1 + 1e-3)
    assert abs(rec["ba_cost"] - ba.last_stats["cost"]) <= 1e-9 * ba.last_stats["cost"]
    P = np.array([x[1] for x in e["lm"]]); Pr = np.array([l.p.reshape(3) for l in s2._landmarks])
    assert np.abs(P - Pr).max() <= 1e-9 * np.abs(Pr).max()
    for i in range(W):
        assert np.abs(e["poses"][t_now - i] - s2._trajectory[t_now - i]).max() <= 1e-9


def test_resident_pipeline_seeded_by_the_sift_bootstrap():
    """the reference's own start: Pipeline._get_init_state (pipeline.py:42-90 -- SIFT on two frames, ratio matching, five-point pose,
    triangulation) through the drop-in Extractor on the device, its State handed to the tables, then 8 frames resident on the device
    against the reference's loop over Python objects from the same State"""
    from test_gpu_e2e import K, _frames
    from vo_mi355x import Extractor, State, Trajectory
    from vo_mi355x.resident import ResidentPipeline
    n_steps, t0, t1 = 8, 0, 4
    frames = _frames(t1 + n_steps + 1)
    h, w = frames[0][0].shape
    ctx_a, ctx_b = _ctx(w, h, max_pts=4096), _ctx(w, h, max_pts=4096)
    ex = Extractor(min_kp_dist=7, ctx=ctx_a)
    kp0 = ex.extract(frames[t0][0], 0, detector='custom', describe=True)
    kp1 = ex.extract(frames[t1][0], 1, detector='custom', describe=True)
    matches = ex.match_lists(kp0, kp1)
    kp0_m, kp1_m, i1_nm = [], [], list(range(len(kp1)))
    for m in matches:
        kp0_m.append(copy.deepcopy(kp0[m.queryIdx])); kp1_m.append(copy.deepcopy(kp1[m.trainIdx]))
        if m.trainIdx in i1_nm:
            i1_nm.remove(m.trainIdx)
    inliers, H1 = ex.camera_pose(K, kp0_m, kp1_m, corr='2D-2D')
    kp0_m = [kp0_m[i] for i in inliers]; kp1_m = [kp1_m[i] for i in inliers]
    landmarks, kp0_m, kp1_m = ex.triangulate_nonlinear(K, np.eye(4), H1, kp0_m, kp1_m, 1, max_err_reproj=2.0)
    assert len(landmarks) >= 150
    state = State(landmarks, kp1_m, [kp1[i] for i in i1_nm], Trajectory({0: np.eye(4), 1: H1}))
    loop = ph.ObjectLoop(ctx_a, K, copy.deepcopy(state), frames[t1][0], ba_window=4, ba_max_iters=16)
    rp = ResidentPipeline(ctx_b, K, ba_window=4, ba_max_iters=16, pnp_blind_batches=8)
    rp.seed(state, [], [], t_step=1)
    ctx_b.push_frame(frames[t1][0])
    for s in range(n_steps):
        im = frames[t1 + 1 + s][0]
        loop.step(im)
        ctx_b.push_frame(im); rp.step(); rec = rp.fetch()
        assert rec["status"] == 0 and rec["overflow"] == 0, rec
        e = rp.entries()
        ph.compare_lists(loop, e, what="step %d" % (s + 2), p_tol=1e-7)
        assert np.abs(rec["H"] - loop.state._trajectory[loop.t_step]).max() <= 1e-7
    # and the run is a sane odometry: the pose of the last frame against the rendered ground truth, in bootstrap baselines
    gt = frames[t1 + n_steps][1]
    unit = np.linalg.norm(frames[t1][1][:3, 3])
    cosang = (np.trace(rec["H"][:3, :3] @ gt[:3, :3].T) - 1) / 2
    assert np.degrees(np.arccos(np.clip(cosang, -1, 1))) <= 0.5 and np.linalg.norm(rec["H"][:3, 3] - gt[:3, 3] / unit) <= 0.75


def test_resident_pipeline_equals_object_loop_at_the_baseline_shape():
    """1241 x 376, ~1000 bootstrap keypoints growing towards the 4096 slots, 10-frame window: frames of the device tables against the
    reference's loop over Python objects until the table is full (four list entries per thread, four pyramid levels, the image size the
    benchmark runs)"""
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1, n_steps = 1241, 376, 4, 5
    sc = ph.scene(t1 + n_steps + 1, w=w, h=h, f=718.856, seed=4321, pose_fn=lambda t: ph.sway_pose(t, period=40.0))
    ctx_a, ctx_b = _ctx(w, h, max_pts=8192), _ctx(w, h, max_pts=4096)
    state, t_loader = ph.gt_bootstrap(ctx_a, sc, 0, t1)
    assert len(state._landmarks) + len(state._candidates_kp) >= 800
    loop = ph.ObjectLoop(ctx_a, sc["K"], copy.deepcopy(state), sc["frames"][t_loader], ba_window=10, ba_max_iters=12)
    rp = ResidentPipeline(ctx_b, sc["K"], ba_window=10, ba_max_iters=12, pnp_blind_batches=8)
    rp.seed(state, [], [], t_step=1)
    ctx_b.push_frame(sc["frames"][t_loader])
    for s in range(n_steps):
        im = sc["frames"][t_loader + 1 + s]
        loop.step(im)
        ctx_b.push_frame(im); rp.step(); rec = rp.fetch()
        what = "step %d" % (s + 2)
        assert rec["status"] == 0, (what, rec)
        if rec["overflow"]:
            break                                        # the 4096 slots are full: from here on the capacity policy (tested against the model) acts
        ph.compare_lists(loop, rp.entries(), what=what, p_tol=1e-7)
        assert np.abs(rec["H"] - loop.state._trajectory[loop.t_step]).max() <= 1e-7
        assert rec["ba_observations"] > 1000 and rec["pnp_inliers"] == loop.info["n_inliers"]
    assert s >= 2


@pytest.mark.gpu
def test_side_stream_layout_equals_one_stream_with_steps_in_flight():
    """With a side stream the re-detection and the spawn of frame t and the pyramid + KLT of frame t + 1 run beside the bundle adjustment of
    frame t (csrc/vo_pipeline.hip, pipe_step).  Full image size, two sequences, three steps in flight: records and tables are bit-identical
    to the same run on ONE stream -- whatever overlaps, nothing reads what the other stream has not finished writing."""
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1, n = 1241, 376, 4, 12
    scs = [ph.scene(t1 + n + 1, w=w, h=h, f=718.856, seed=sd, pose_fn=lambda t: ph.sway_pose(t, period=40.0)) for sd in (99, 4321)]
    boot = _ctx(w, h, max_pts=4096)
    states = [ph.gt_bootstrap(boot, sc, 0, t1)[0] for sc in scs]
    runs = []
    for side in (True, False):
        c = _ctx(w, h, max_pts=2048, batch=2)
        c.set_side_stream(side)
        c.upload_sequence(np.stack([sc["frames"] for sc in scs]))
        rp = ResidentPipeline(c, np.stack([sc["K"] for sc in scs]), ba_window=4, ba_max_iters=10)
        rp.seed([copy.deepcopy(s) for s in states], None, None, 1)
        c.push_frame_resident(t1)
        recs = []
        for s0 in range(0, n, 3):
            for s in range(s0, s0 + 3):
                rp.step(t1 + 1 + s)
            for s in range(s0, s0 + 3):
                recs.append(rp.fetch())
        runs.append((recs, rp.read_tables()))
    (ra, Ta), (rb, Tb) = runs
    for s in range(n):
        for b in range(2):
            assert ra[s][b]["status"] == 0 and ra[s][b]["ba_observations"] > 1000, (s, b, ra[s][b])
            for k in ra[s][b]:
                assert np.array_equal(np.asarray(ra[s][b][k]), np.asarray(rb[s][b][k])), (s, b, k, ra[s][b][k], rb[s][b][k])
    for name in Ta:
        assert np.array_equal(Ta[name], Tb[name], equal_nan=True) if Ta[name].dtype.kind == "f" else np.array_equal(Ta[name], Tb[name]), name


@pytest.mark.gpu
def test_device_side_corner_limit_gives_the_same_corners(monkeypatch):
    """In the closed loop the re-detection can use at most (free slots of the table + 1) corners of a sequence; k_st_select then consumes the
    candidate list in short rank-ordered chunks (radix select of the strongest 512 ...) instead of sorting all ~1 600 candidates of a full-size
    frame.  Same corners, same records, same tables as with the limit switched off (vo_tuning.st_host_limit), from an empty-ish table that takes
    hundreds of corners per frame to the full table that takes a handful."""
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1, n = 1241, 376, 4, 10
    scs = [ph.scene(t1 + n + 1, w=w, h=h, f=718.856, seed=sd, pose_fn=lambda t: ph.sway_pose(t, period=40.0)) for sd in (99, 4321)]
    boot = _ctx(w, h, max_pts=4096)
    states = [ph.gt_bootstrap(boot, sc, 0, t1)[0] for sc in scs]
    runs = []
    for host_limit in (0, 1):
        c = _ctx(w, h, max_pts=1500, batch=2)              # 1 500 slots: the bootstrap fills ~1 000, the table is full after a few frames
        c.set_tuning(st_host_limit=host_limit)
        c.upload_sequence(np.stack([sc["frames"] for sc in scs]))
        rp = ResidentPipeline(c, np.stack([sc["K"] for sc in scs]), ba_window=4, ba_max_iters=10)
        rp.seed([copy.deepcopy(s) for s in states], None, None, 1)
        c.push_frame_resident(t1)
        recs = []
        for s in range(n):
            rp.step(t1 + 1 + s); recs.append(rp.fetch())
        runs.append((recs, rp.read_tables()))
    (ra, Ta), (rb, Tb) = runs
    dets = [ra[s][b]["n_detected"] for s in range(n) for b in range(2)]
    assert max(dets) >= 100 and min(dets) <= 40, dets      # both regimes were visited
    for s in range(n):
        for b in range(2):
            assert ra[s][b]["status"] == 0
            for k in ra[s][b]:
                assert np.array_equal(np.asarray(ra[s][b][k]), np.asarray(rb[s][b][k])), (s, b, k, ra[s][b][k], rb[s][b][k])
    for name in Ta:
        assert np.array_equal(Ta[name], Tb[name], equal_nan=True) if Ta[name].dtype.kind == "f" else np.array_equal(Ta[name], Tb[name]), name


def test_tables_beyond_4096_slots_at_1080p_window20():
    """BASELINE config 5's shape in the closed loop: 1920 x 1080, a 20-frame window, tables of 8 192 slots (the list kernels' 8-entries-per-thread
    form: the per-landmark-row words move from LDS to the sequence's global scratch) -- against the table model frame by frame while the state
    grows past 4 096 keypoints (1 000 new corners per frame)"""
    import pipe_oracle as po
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1, n = 1920, 1080, 3, 9
    cap = 8192
    sc = ph.scene(t1 + n + 1, w=w, h=h, f=1100.0, seed=11, pose_fn=lambda t: ph.sway_pose(t, period=40.0))
    ctx_a, ctx_b = _ctx(w, h, max_pts=cap), _ctx(w, h, max_pts=cap)
    state, _ = ph.gt_bootstrap(ctx_a, sc, 0, t1)
    model = po.PipeModel(ctx_a, sc["K"], w, h, cap=cap, params=po.Params(ba_window=20, ba_max_iters=12))
    model.seed(copy.deepcopy(state), [], [], 1)
    ctx_a.push_frame(sc["frames"][t1])
    rp = ResidentPipeline(ctx_b, sc["K"], ba_window=20, ba_max_iters=12, pnp_blind_batches=8)
    rp.seed(state, [], [], 1)
    ctx_b.push_frame(sc["frames"][t1])
    biggest, clean = 0, True
    for s in range(n):
        im = sc["frames"][t1 + 1 + s]
        model.step(im)
        ctx_b.push_frame(im); rp.step(); rec = rp.fetch()
        what = "step %d" % (s + 2)
        # (at a 20-frame window the reference's resurrection brings every young death back every frame: after a few frames even 8 192 slots are
        #  full and the capacity policy acts -- identically in the model)
        assert rec["status"] == 0 and model.status == 0 and rec["overflow"] == model.info.get("overflow", 0), (what, rec, model.info)
        clean = clean and rec["overflow"] == 0
        e = rp.entries()
        assert (len(e["cand"]), len(e["lm"]), len(e["dead"]), e["n_dead_total"]) == (len(model.cand), len(model.lm_L), len(model.dead_L),
                                                                                    len(model.dead_L) + model.n_dead_inert), what
        assert (rec["n_new"], rec["n_resurrected"], rec["n_detected"]) == (model.info["n_new"], model.info["n_resurrected"], model.info["n_detected"]), (what, rec)
        if clean:
            biggest = max(biggest, len(model.cand) + len(model.lm_L))
        for name, pairs in (("lm", zip(model.lm_L, model.lm_K)), ("dead", zip(model.dead_L, model.dead_K)), ("cand", ((None, k) for k in model.cand))):
            for i, ((l, k), x) in enumerate(zip(pairs, e[name])):
                y = model.entry(l, k)
                assert x[2:4] == y[2:4] and x[6] == y[6] and np.array_equal(x[5], y[5]) and np.array_equal(x[4], y[4]) and np.array_equal(x[7], y[7]), (what, name, i)
                if l is not None:
                    assert x[0] == y[0] and np.linalg.norm(x[1] - y[1]) <= 1e-7 * np.linalg.norm(y[1]), (what, name, i)
        for t in range(model.t + 1):
            assert np.abs(e["poses"][t] - model.poses[t]).max() <= 1e-7, (what, t)
    assert biggest > 4096, biggest          # reached with nothing cut


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["pinned", "pageable"])
def test_closed_loop_with_host_frames_equals_resident_frames(source):
    """`vo_pipe_step_host` (the new frame handed over by the host, as Pipeline.step(img) receives it: pipeline.py:98,171-172; upload on the copy
    stream, pyramid + tracking on the side stream) = `vo_pipe_step` on the uploaded sequence: records and tables bit for bit, three steps in flight,
    a batch of 3 sequences; page-locked images (one gather launch) and pageable ones (staged copies)"""
    from vo_mi355x import VoContext
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1, n = 256, 160, 3, 9
    scs = [ph.scene(t1 + n + 1, w=w, h=h, f=260.0, seed=sd, pose_fn=lambda t: ph.sway_pose(t, period=24.0)) for sd in (2024, 77)]
    order = [0, 1, 0]
    boot = _ctx(w, h)
    states = [ph.gt_bootstrap(boot, sc, 0, t1)[0] for sc in scs]
    frames = np.stack([scs[i]["frames"] for i in order])            # [3, frames, h, w]

    def run(host):
        c = _ctx(w, h, batch=3)
        rp = ResidentPipeline(c, np.stack([scs[i]["K"] for i in order]), ba_max_iters=12)
        rp.seed([copy.deepcopy(states[i]) for i in order], None, None, 1)
        if host is None:
            c.upload_sequence(frames)
            c.push_frame_resident(t1)
        else:
            c.push_frame(frames[:, t1])
        recs = []
        for s0 in range(0, n, 3):
            for s in range(s0, s0 + 3):
                if host is None:
                    rp.step(t1 + 1 + s)
                else:
                    rp.step_host(host(t1 + 1 + s))
            for s in range(s0, s0 + 3):
                recs.append(rp.fetch())
        return recs, rp.read_tables()

    ref_recs, ref_T = run(None)
    if source == "pinned":
        store = VoContext.host_alloc(scs[0]["frames"].shape[:1] + (2,) + (h, w))      # [frame][scene][h][w]: sequences 0 and 2 share an image
        for k in range(2):
            store[:, k] = scs[k]["frames"]
        give = lambda f: [store[f, i] for i in order]
    else:
        give = lambda f: [scs[i]["frames"][f] for i in order]
    got_recs, got_T = run(give)
    for s in range(n):
        for b in range(3):
            for k, v in ref_recs[s][b].items():
                if isinstance(v, np.ndarray):
                    assert np.array_equal(got_recs[s][b][k], v), (s, b, k)
                else:
                    assert got_recs[s][b][k] == v, (s, b, k, got_recs[s][b][k], v)
    for name in ref_T:
        assert np.array_equal(got_T[name], ref_T[name]), name
