"""GPU: the device-resident track table against the reference's own Extractor bookkeeping
(golden G3: extend_tracks x2 + extract run unmodified on the CPU oracle) and against a Python-list model of the same
rules over a long sequence (deaths, re-detection, capacity, history-ring wrap, ragged counts in a batch)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _hist(g, pre):
    ln = g[pre + "_hist_len"]
    h, o = [], 0
    for n in ln:
        h.append(g[pre + "_hist"][o:o + n]); o += n
    return h


def test_tracks_replay_reference_glue_golden():
    from vo_mi355x import VoContext
    g = np.load(os.path.join(GOLD, "glue_s0.npz"))
    frames = g["frames"]
    h, w = frames[0].shape
    in_uv = g["in0_uv"].astype(np.float32)
    assert np.array_equal(in_uv.astype(np.float64), g["in0_uv"])                    # the golden inputs are float32 values
    pos_of_tag = {float(t): i for i, t in enumerate(g["in0_tag"])}
    with VoContext(w, h, max_pts=512) as c:
        c.push_frame(frames[0])
        c.tracks_seed(in_uv, t=1)                                                  # extract(..., t=1) + the injected border tracks
        c.push_frame(frames[1])
        c.tracks_track(2)                                                          # extend_tracks(frames[1], ...)
        r = c.tracks_read()
        want_tags = np.array([pos_of_tag[float(t)] for t in g["tr1_tag"]])
        assert np.array_equal(r["tag"], want_tags)                                 # same survivors, same order
        assert np.array_equal(r["uv"].astype(np.float64), g["tr1_uv"]) and np.array_equal(r["uv_first"].astype(np.float64), g["tr1_uv_first"])
        assert np.array_equal(r["t_total"], g["tr1_t_total"]) and np.array_equal(r["t_first"], g["tr1_t_first"])
        dead = sorted(set(range(len(in_uv))) - set(want_tags.tolist()))
        assert sorted(r["dead_tag"].tolist()) == dead and len(dead) == len(in_uv) - len(want_tags) > 0
        obs = c.tracks_obs(t_now=2, window=3)
        for i, hh in enumerate(_hist(g, "tr1")):                                   # uv_history = [uv(t=1), uv(t=2)]
            assert np.array_equal(obs[0, i], hh[1]) and np.array_equal(obs[1, i], hh[0]) and np.isnan(obs[2, i]).all()
        assert np.isnan(obs[:, len(want_tags):]).all()
        c.push_frame(frames[2])
        c.tracks_track(3)                                                          # extend_tracks(frames[2], ...)
        r = c.tracks_read()
        assert np.array_equal(r["tag"], np.array([pos_of_tag[float(t)] for t in g["tr2_tag"]]))
        assert np.array_equal(r["uv"].astype(np.float64), g["tr2_uv"]) and np.array_equal(r["t_total"], g["tr2_t_total"])
        obs = c.tracks_obs(t_now=3, window=4)
        for i, hh in enumerate(_hist(g, "tr2")):
            assert all(np.array_equal(obs[s, i], hh[2 - s]) for s in range(3)) and np.isnan(obs[3, i]).all()
        n_before = len(r["tag"])
        c.tracks_detect(3, mask_radius=7, params=c.st_params(min_distance=7), max_new=1000)   # extract(frames[2], 3, c2, 'shi-tomasi', 7)
        r = c.tracks_read()
        new = slice(n_before, None)
        assert np.array_equal(r["uv"][new].astype(np.float64), g["ex2_uv"]) and len(r["uv"]) == n_before + len(g["ex2_uv"])
        assert np.array_equal(r["t_first"][new], g["ex2_t_first"]) and np.array_equal(r["t_total"][new], g["ex2_t_total"])
        assert np.array_equal(r["uv_first"][new], r["uv"][new]) and np.array_equal(r["tag"][new], 268 + np.arange(len(g["ex2_uv"])))


def _list_model(frames, seeds, cap, n_frames, max_new, mc):
    """the reference's rules on Python lists, OpenCV arithmetic from the CPU oracle"""
    import vo_oracle as o
    h, w = frames[0].shape
    tr = [dict(uv=p.copy(), first=p.copy(), tf=0, tt=1, tag=i, hist={0: p.copy()}) for i, p in enumerate(seeds)]
    next_tag, log = len(tr), []
    for t in range(1, n_frames):
        dead = []
        if tr:
            p1, _, _ = o.klt(frames[t - 1], frames[t], np.array([k["uv"] for k in tr], np.float32))
            keep = []
            for k, (x, y) in zip(tr, p1):
                if 0 <= x <= w and 0 <= y <= h:
                    k["uv"] = np.array([x, y], np.float32); k["tt"] += 1; k["hist"][t] = k["uv"].copy(); keep.append(k)
                else:
                    dead.append(k["tag"])
            tr = keep
        if t % 3 == 0:
            mask = np.full((h, w), 255, np.uint8)
            for k in tr:
                o.circle_mask(mask, tuple(np.int32(k["uv"])), 5, 0)
            new = o.good_features(frames[t], mask, maxCorners=mc, qualityLevel=0.03, minDistance=5, blockSize=15)
            for q in new[:min(len(new), max_new, cap - len(tr))]:
                q = q.astype(np.float32)
                tr.append(dict(uv=q.copy(), first=q.copy(), tf=t, tt=1, tag=next_tag, hist={t: q.copy()})); next_tag += 1
        log.append(([dict(k, hist=dict(k["hist"])) for k in tr], dead))
    return log


def test_tracks_long_sequence_vs_list_model_and_ragged_batch():
    from vo_mi355x import VoContext, synthetic as syn
    w, h, cap, T, max_new, mc = 160, 120, 96, 40, 40, 30
    fr = [syn.make_sequence(T, w=w, h=h, seed=70 + b, margin=96)[0] for b in range(2)]     # 40 frames: the 32-deep ring wraps
    seeds = [syn.grid_points(60, w, h, seed=5, margin=12), syn.grid_points(35, w, h, seed=6, margin=30)]
    logs = [_list_model(fr[b], seeds[b], cap, T, max_new, mc) for b in range(2)]

    def check(r, obs, want, dead, t, W_):
        assert np.array_equal(r["tag"], [k["tag"] for k in want]), t
        if want:
            assert np.array_equal(r["uv"], np.array([k["uv"] for k in want])) and np.array_equal(r["uv_first"], np.array([k["first"] for k in want]))
            assert np.array_equal(r["t_first"], [k["tf"] for k in want]) and np.array_equal(r["t_total"], [k["tt"] for k in want])
        assert sorted(r["dead_tag"].tolist()) == sorted(dead)
        for i, k in enumerate(want):
            for s in range(W_):
                tau = t - s
                if tau in k["hist"]:
                    assert np.array_equal(obs[s, i], k["hist"][tau].astype(np.float64)), (t, i, s)
                else:
                    assert np.isnan(obs[s, i]).all()
        assert np.isnan(obs[:, len(want):]).all()

    stp = None
    # batch of 2 with different seed counts (the shorter list is padded for the upload and pruned right away: padding is
    # placed outside the image so that the first extend kills it -- exactly what a caller with ragged lists would do)
    pad = np.full((60 - 35, 2), -100.0, np.float32)
    with VoContext(w, h, max_pts=cap, batch=2) as c:
        stp = c.st_params(max_corners=mc, quality_level=0.03, min_distance=5, block_size=15)
        c.push_frame(np.stack([fr[0][0], fr[1][0]]))
        c.tracks_seed(np.stack([seeds[0], np.vstack([seeds[1], pad])]), t=0)
        for t in range(1, T):
            c.push_frame(np.stack([fr[0][t], fr[1][t]]))
            c.tracks_track(t)
            if t % 3 == 0:
                c.tracks_detect(t, mask_radius=5, params=stp, max_new=max_new)
            rs = c.tracks_read()
            obs = c.tracks_obs(t, 20)
            for b in range(2):
                want, dead = logs[b][t - 1]
                got_dead = rs[b]["dead_tag"]
                if b == 1 and t == 1:
                    got_dead = got_dead[got_dead < 35]                     # the padding died here
                    rs[b] = dict(rs[b], dead_tag=got_dead)
                    tagmap = None
                check(rs[b], obs[b], want if b == 0 else [dict(k, tag=k["tag"] + (25 if k["tag"] >= 35 else 0)) for k in want],
                      dead if b == 0 else [d + (25 if d >= 35 else 0) for d in dead], t, 20)
    assert max(len(x[0]) for x in logs[0]) > 60 and any(x[1] for x in logs[0])      # the scenario exercises growth and deaths


def test_ba_observations_straight_from_the_track_ring():
    """vo_ba_obs_from_tracks fills the resident BA problem on the device; solving it equals solving the host-gathered table"""
    from vo_mi355x import VoContext, synthetic as syn
    w, h, cap, W = 240, 180, 64, 5
    frames, _ = syn.make_sequence(W + 1, w=w, h=h, seed=9, margin=64)
    seeds = syn.grid_points(50, w, h, seed=2, margin=20)
    s = syn.make_ba_scene(n_pts=cap, n_slots=W, seed=4)
    with VoContext(w, h, max_pts=cap) as c:
        c.push_frame(frames[0])
        c.tracks_seed(seeds, t=0)
        for t in range(1, W + 1):
            c.push_frame(frames[t])
            c.tracks_track(t)
        obs_host = c.tracks_obs(W, W)
        n_live = len(c.tracks_read()["tag"])
        assert 0 < n_live <= 50 and np.isfinite(obs_host[:, :n_live]).all() and np.isnan(obs_host[:, n_live:]).all()
        # resident problem with placeholder observations, then the device gather
        c.ba_upload(s["K"], s["poses0"], s["points0"], np.full((W, cap, 2), np.nan))
        c.ba_obs_from_tracks(W)
        c.ba_solve_resident(c.ba_params(max_iters=3))
        po_a, pt_a, st_a = c.ba_fetch()
        po_b, pt_b, st_b = c.ba_adjust(s["K"], s["poses0"], s["points0"], obs_host, c.ba_params(max_iters=3))
    assert np.array_equal(po_a, po_b) and np.array_equal(pt_a, pt_b) and st_a["cost"] == st_b["cost"] and st_a["iters"] == st_b["iters"]
