"""G5 on the device -- CONDITIONAL ON ONE SUBSTITUTION (see tests/test_pipe_golden.py): the goldens are the reference's `Pipeline.step` with scipy's
`least_squares` replaced by the LM of oracle/ba_oracle.py.

`vo_mi355x.resident.ResidentPipeline` (csrc/vo_pipeline.hip, every stage a HIP kernel) against the reference's
OWN `Pipeline.step` (/root/reference/src/pipeline/pipeline.py:92-167, dumped frame by frame by tests/golden/gen_golden.py --pipe-only
over the CPU oracle; tests/test_pipe_golden.py holds the CPU restatements to the same files bit for bit).

Exact: every list's length and order, birth frame, track length, history length, every float32 pixel position and history entry,
`t_latest`, which entries share which Landmark / Keypoint object, the number of dead entries.  Within a stated tolerance: landmark positions
and poses -- the device's DLT (float64 one-sided Jacobi, float32 output) and LM (MFMA Gram product, other summation order) are not the
oracle's arithmetic bit for bit; P_TOL / POSE_TOL below are what the stage tolerances of SURVEY 8a' allow after twelve closed-loop frames.
Also: the device tables against `pipe_oracle.PipeModel` over the CPU-oracle context on another scene (the same comparison without a file)."""
import json
import os

import numpy as np
import pytest

import pipe_golden as pg
import pipe_helpers as ph

pytestmark = pytest.mark.gpu

P_TOL = 1e-4       # relative, landmark positions (units of the bootstrap baseline; a landmark sits ~40 units away)
POSE_TOL = 2e-5    # absolute, entries of [R | t]   (measured over the three goldens: gpurun_out/g5_deviation_*.json -> profiles/r04_g5_deviation.txt)


def _ctx(w, h, max_pts=2048):
    from vo_mi355x import VoContext
    return VoContext(w, h, max_pts=max_pts)


def _deviation(ref, got):
    """largest relative position / absolute pose difference of a frame (for the report)"""
    dp = max([np.linalg.norm(a[1] - np.float64(b[1]).reshape(3)) / max(np.linalg.norm(a[1]), 1e-12) for n in ("lm", "dead") for a, b in zip(ref[n], got[n])] + [0.0])
    dH = max([np.abs(np.asarray(H)[:3] - ref["traj"][t][:3]).max() for t, H in got["poses"].items()] + [0.0])
    return dp, dH


@pytest.mark.parametrize("name,n_steps", [("w4", 12), ("w10", 10), ("w20", 8), ("groups", 5)])
def test_device_tables_equal_the_references_own_pipeline_step(name, n_steps):
    from vo_mi355x.resident import ResidentPipeline
    g = pg.load(name)
    sc = pg.scene_frames(g)
    w, h, W, t0 = int(g["w"]), int(g["h"]), int(g["ba_window"]), int(g["t_step0"])
    fos = g["frame_of_step"]
    ctx = _ctx(w, h)
    state, dead, dead_kp = pg.seed_objects(g)
    rp = ResidentPipeline(ctx, sc["K"], ba_window=W, ba_max_iters=50, pnp_blind_batches=8)
    rp.seed(state, dead, dead_kp, t_step=t0)
    ctx.push_frame(sc["frames"][fos[t0]])
    pg.assert_entries(pg.frame(g, 0), pg.device_entries(rp), "device seed")
    worst = [0.0, 0.0]
    for s in range(1, n_steps + 1):
        ctx.push_frame(sc["frames"][fos[t0 + s]])
        rp.step()
        rec = rp.fetch()
        what = "device %s step %d" % (name, s)
        assert rec["status"] == 0 and rec["overflow"] == 0 and rec["t"] == t0 + s, (what, rec)
        ref, got = pg.frame(g, s), pg.device_entries(rp)
        dp, dH = _deviation(ref, got) if (len(ref["lm"]), len(ref["dead"])) == (len(got["lm"]), len(got["dead"])) else (np.nan, np.nan)
        worst = [max(worst[0], dp), max(worst[1], dH)]
        pg.assert_entries(ref, got, what, p_tol=P_TOL, pose_tol=POSE_TOL)
        info = g["info"][s - 1]           # pnp n, inliers, landmarks, candidates, dead, ba iters, ba landmarks, ba observations
        assert (rec["n_tracked"] >= info[0], rec["pnp_inliers"], rec["n_landmarks"], rec["n_candidates"], rec["n_dead_total"]) == \
               (True, info[1], info[2], info[3], info[4]), (what, rec, info)
        assert (rec["ba_landmarks"], rec["ba_observations"]) == (info[6], info[7]), (what, rec, info)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "g5_deviation_%s.json" % name), "w") as f:
            json.dump(dict(case=name, steps=n_steps, max_rel_position=worst[0], max_abs_pose=worst[1], p_tol=P_TOL, pose_tol=POSE_TOL), f)
    if name == "groups":
        # the frame with four ripe birth groups: walked in CPython's set order on the device (pipe_set_order)
        births = [e[2] for e in pg.device_entries(rp)["lm"]]
        runs = [births[0]] + [b for a, b in zip(births[:-1], births[1:]) if a != b]
        assert runs[:4] == [16, 9, 2, 8], runs


@pytest.mark.parametrize("ba_window,n_steps,seed,period,amp", [(4, 10, 7, 16.0, (1.6, 0.4, -0.8)), (6, 8, 99, 20.0, (1.2, -0.3, 0.6))])
def test_device_tables_equal_the_table_model_over_the_cpu_oracle(ba_window, n_steps, seed, period, amp):
    """closed loop on the GPU = closed loop on the oracle: PipeModel's numerical calls go to tests/oracle_context.OracleContext (C / numpy
    restatements), the device's to the HIP kernels; other scenes and faster motion than the goldens"""
    import copy
    import pipe_oracle as po
    from oracle_context import OracleContext
    from vo_mi355x.resident import ResidentPipeline
    w, h, t1 = 256, 160, 3
    sc = ph.scene(t1 + n_steps + 1, w=w, h=h, f=260.0, seed=seed, pose_fn=lambda t: ph.sway_pose(t, amp=amp, period=period))
    octx, gctx = OracleContext(w, h), _ctx(w, h)
    state, t_loader = ph.gt_bootstrap(octx, sc, 0, t1)
    model = po.PipeModel(octx, sc["K"], w, h, cap=2048, params=po.Params(ba_window=ba_window))
    model.seed(copy.deepcopy(state), [], [], 1)
    octx.push_frame(sc["frames"][t_loader])
    rp = ResidentPipeline(gctx, sc["K"], ba_window=ba_window, ba_max_iters=50, pnp_blind_batches=8)
    rp.seed(state, [], [], t_step=1)
    gctx.push_frame(sc["frames"][t_loader])
    for s in range(n_steps):
        im = sc["frames"][t_loader + 1 + s]
        model.step(im)
        gctx.push_frame(im); rp.step(); rec = rp.fetch()
        what = "step %d" % (s + 2)
        assert rec["status"] == 0 and model.status == 0 and rec["overflow"] == 0, (what, rec)
        ref = pg.model_entries(model)
        ref["traj"] = {t: H for t, H in model.poses.items()}
        got = pg.device_entries(rp)
        pg.assert_entries(ref, got, what, p_tol=P_TOL, pose_tol=POSE_TOL)
        assert (rec["n_new"], rec["n_resurrected"], rec["n_detected"]) == (model.info["n_new"], model.info["n_resurrected"], model.info["n_detected"]), (what, rec, model.info)
