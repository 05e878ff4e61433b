"""Randomised STATES through single stages of the device-resident pipeline against the table model (oracle/pipe_oracle.py, itself
pinned to the reference's loop over Python objects): lists with duplicates, landmark objects shared between several state entries and
dead entries, keypoint objects shared with the dead list, histories of every length up to beyond the 32-entry ring, empty lists, lists
that fill the tables exactly, tracks at and beyond the image border.  After each stage every list entry (row contents through the
indices, order, who shares what) must be equal; landmark positions / poses to 1e-9 where a solve ran."""
import copy

import numpy as np
import pytest

import pipe_helpers as ph

pytestmark = pytest.mark.gpu
W_IMG, H_IMG = 256, 160


def _random_state(rng, cap, t_step, n_l, n_c, n_d, share=0.3, long_hist=False):
    """State + dead lists with object sharing like the reference produces (and a little worse)"""
    from vo_mi355x import Keypoint, Landmark, State, Trajectory

    def kp(t_first=None, border=False):
        L = int(rng.integers(1, 40 if long_hist else 9))
        t_first = t_step - L + 1 - int(rng.integers(0, 3)) if t_first is None else t_first
        hist = [np.float32(rng.uniform([4, 4], [W_IMG - 4, H_IMG - 4])).reshape(2, 1) for _ in range(L)]
        if border:
            hist[-1] = np.float32([[rng.choice([-3.0, 0.0, W_IMG, W_IMG + 2.5, 17.25])], [rng.choice([-1.0, 0.0, H_IMG, H_IMG + 4.0, 33.5])]])
        return Keypoint(int(t_first), L, hist[0].copy(), hist[-1].copy(), np.zeros((1, 1)), hist)

    def lm(k):
        return Landmark(int(k.t_first + len(k.uv_history) - 1 + rng.integers(0, 2)), rng.normal(0, 3, (3, 1)) + np.array([[0], [0], [12.0]]), np.zeros((1, 1)))
    lms, kps = [], []
    for _ in range(n_l):
        k = kp(border=rng.random() < 0.15)
        if lms and rng.random() < share:                 # another entry of an existing landmark object
            lms.append(lms[int(rng.integers(0, len(lms)))]); kps.append(k)
        else:
            lms.append(lm(k)); kps.append(k)
    cands = [kp(t_first=t_step - int(rng.integers(0, 4)), border=rng.random() < 0.15) for _ in range(n_c)]
    for c in cands:
        c.t_total = len(c.uv_history)
    dead, dead_kp = [], []
    for _ in range(n_d):
        r = rng.random()
        if lms and r < share:                            # shares landmark AND keypoint object with a state entry (just resurrected)
            j = int(rng.integers(0, len(lms))); dead.append(lms[j]); dead_kp.append(kps[j])
        elif lms and r < 2 * share:                      # shares only the landmark object (resurrected earlier, survived since)
            j = int(rng.integers(0, len(lms))); dead.append(lms[j]); dead_kp.append(kp())
        else:
            k = kp(); dead.append(lm(k)); dead_kp.append(k)
    traj = Trajectory({})
    for t in range(t_step + 1):
        H = np.eye(4)
        ang = 0.01 * t
        H[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]; H[:3, 3] = [0.1 * t, 0.02 * t, -0.05 * t]
        traj.append(t, H)
    return State(lms, kps, cands, traj), dead, dead_kp


def _same(model, rp, what, p_tol=1e-9):
    e = rp.entries()
    m = dict(cand=[model.entry(None, k) for k in model.cand], lm=[model.entry(l, k) for l, k in zip(model.lm_L, model.lm_K)],
             dead=[model.entry(l, k) for l, k in zip(model.dead_L, model.dead_K)])
    assert e["status"] == model.status, (what, e["status"], model.status)
    assert e["n_dead_total"] == len(model.dead_L) + model.n_dead_inert, what
    for name in ("cand", "lm", "dead"):
        assert len(e[name]) == len(m[name]), (what, name, len(e[name]), len(m[name]))
        for i, (x, y) in enumerate(zip(e[name], m[name])):
            assert x[0] == y[0] and x[2:4] == y[2:4] and x[6] == y[6], (what, name, i, x[:4], y[:4])
            assert np.array_equal(x[4], y[4]) and np.array_equal(x[5], y[5]) and np.array_equal(x[7], y[7], equal_nan=True), (what, name, i)
            if y[1] is not None:
                assert np.abs(x[1] - y[1]).max() <= p_tol * max(1.0, np.abs(y[1]).max()), (what, name, i, x[1], y[1])
    # sharing structure: entries that refer to one row in the model refer to one row on the device (rows are numbered differently)
    def classes(ids):
        seen = {}
        return [seen.setdefault(int(v), len(seen)) for v in ids]
    r = e["rows"]
    assert classes(list(r["lm_l"]) + list(r["dead_l"])) == classes(model.lm_L + model.dead_L), what          # landmark rows
    assert classes(list(r["lm_k"]) + list(r["dead_k"])) == classes(model.lm_K + model.dead_K), what          # keypoint rows


@pytest.mark.parametrize("cap", [64, 1024, 1100, 2500, 5000, 8192])      # 1 / 2 / 4 / 8 entries per thread (above 4 096: row words in global scratch)
def test_random_states_stage_by_stage(cap):
    import pipe_oracle as po
    from vo_mi355x import VoContext
    from vo_mi355x.resident import ResidentPipeline, TRACK, TRIANGULATE, ADJUST, DETECT
    rng = np.random.default_rng(1000 + cap)
    sc = ph.scene(3, w=W_IMG, h=H_IMG, f=260.0, seed=5, pose_fn=lambda t: ph.sway_pose(t, period=24.0))
    ca, cb = VoContext(W_IMG, H_IMG, max_pts=cap), VoContext(W_IMG, H_IMG, max_pts=cap)
    for trial in range(14 if cap <= 64 else 6 if cap < 2000 else 3 if cap < 4096 else 2):
        full = trial % 3 == 2
        n_l = int(rng.integers(0, cap // 2)) if not full else cap // 2
        n_c = int(rng.integers(0, cap // 3)) if not full else cap - n_l          # the lists fill the table exactly
        n_d = int(rng.integers(0, cap // 2)) if trial % 4 else 0
        W = int(rng.choice([4, 10, 20]))
        t_step = int(rng.integers(6, 30))
        state, dead, dead_kp = _random_state(rng, cap, t_step, n_l, n_c, n_d, long_hist=trial % 2 == 1)
        for stage, name in ((TRACK, "track"), (ADJUST, "adjust"), (DETECT, "detect"), (TRIANGULATE, "triangulate")):
            model = po.PipeModel(ca, sc["K"], W_IMG, H_IMG, cap=cap, params=po.Params(ba_window=W, ba_max_iters=6))
            model.seed(copy.deepcopy(state), copy.deepcopy(dead), copy.deepcopy(dead_kp), t_step)
            # (deepcopy of the three lists one by one would break the sharing between them: copy them as ONE structure)
            st2, d2, dk2 = copy.deepcopy((state, dead, dead_kp))
            model = po.PipeModel(ca, sc["K"], W_IMG, H_IMG, cap=cap, params=po.Params(ba_window=W, ba_max_iters=6))
            model.seed(st2, d2, dk2, t_step)
            rp = ResidentPipeline(cb, sc["K"], ba_window=W, ba_max_iters=6)
            st3, d3, dk3 = copy.deepcopy((state, dead, dead_kp))
            rp.seed(st3, d3, dk3, t_step=t_step)
            what = "cap %d trial %d %s (n_l %d n_c %d n_d %d W %d)" % (cap, trial, name, n_l, n_c, n_d, W)
            _same(model, rp, what + " seed")
            if stage == TRACK:
                for c, mdl in ((ca, None), (cb, None)):
                    c.push_frame(sc["frames"][0])
                model.track(sc["frames"][1])
                cb.push_frame(sc["frames"][1]); rp.step(-1, TRACK)
            elif stage == ADJUST:
                model.adjust()
                rp.step(-1, ADJUST)
            elif stage == DETECT:
                ca.push_frame(sc["frames"][2]); cb.push_frame(sc["frames"][2])
                model.detect()
                rp.step(-1, DETECT)
            else:
                model.triangulate()
                rp.step(-1, TRIANGULATE)
            rec = rp.fetch()
            if stage == ADJUST and model.status == 0:
                _same(model, rp, what, p_tol=1e-6)       # a 6-iteration LM on random geometry: positions agree as far as the solve is conditioned
                assert rec["n_resurrected"] == model.info["n_resurrected"], what
            else:
                _same(model, rp, what)
            assert rec["overflow"] == model.info.get("overflow", 0), (what, rec["overflow"], model.info)
    ca.close(); cb.close()
