"""The lazy object boundary (vo_mi355x/lazy.py): the reference's `Pipeline.step` call sequence (pipeline.py:92-167, restated by
tests/pipe_helpers.ObjectLoop with `if i in inliers` written as the reference writes it) over the drop-in Extractor / BundleAdjuster, whose
lists are views of the pipeline tables -- on the CPU the tables are the model of oracle/pipe_oracle.py (tests/lazy_model_backend.py), the
numerical calls the oracle's.  After every frame the caller's objects, materialised field by field through the proxies, must equal the
reference's OWN run (G5 goldens, tests/golden/pipe_*.npz) entry for entry, bit for bit -- sharing, the dropped-as-inert dead entries the
caller still holds, the trajectory included."""
import copy

import numpy as np
import pytest

import pipe_golden as pg
import pipe_helpers as ph
from lazy_model_backend import ModelBackend
from test_adapters import _oracle_ctx


def _loop(g, lazy=True, cap=2048):
    sc = pg.scene_frames(g)
    w, h, W, t0 = int(g["w"]), int(g["h"]), int(g["ba_window"]), int(g["t_step0"])
    fos = g["frame_of_step"]
    ctx = _oracle_ctx(w, h)
    ctx.max_pts, ctx.batch = cap, 1
    state, dead, dead_kp = pg.seed_objects(g)
    loop = ph.ObjectLoop(ctx, sc["K"], state, sc["frames"][fos[t0]], t_step=t0, ba_window=W, literal=True, lazy=lazy,
                         lazy_backend=lambda c, K, prm, ww, hh: ModelBackend(c, K, prm, ww, hh, cap=cap))
    loop.dead, loop.dead_kp = dead, dead_kp
    return loop, sc, fos, t0


@pytest.mark.parametrize("name,n_steps", [("w4", 12), ("w10", 10), ("w20", 8), ("groups", 5)])
def test_lazy_lists_equal_the_references_own_pipeline_step(name, n_steps):
    g = pg.load(name)
    loop, sc, fos, t0 = _loop(g)
    ex = loop.extractor
    for s in range(1, n_steps + 1):
        loop.step(sc["frames"][fos[t0 + s]])
        sess = ex._lazy
        assert sess is not None and sess.alive, (s, getattr(ex, "_lazy_error", None), sess and sess.reason)
        # frame 1 runs the plain path and seeds the tables at the end of adjust (extract is already a view); from frame 2 on all six calls are
        assert sess.stats["fast"] == (1 if s == 1 else 1 + 6 * (s - 1)), (s, sess.stats)
        pg.assert_entries(pg.frame(g, s), pg.loop_entries(loop), "lazy %s step %d" % (name, s))
    # the caller holds proxies, the same object for the same table row
    st = loop.state
    from vo_mi355x.lazy import InlierList, LazyKeypoint, LazyLandmark
    assert all(type(l) is LazyLandmark for l in st._landmarks) and all(type(k) is LazyKeypoint for k in st._landmarks_kp + st._candidates_kp)
    assert sess.stats["gathers"] < 12 * n_steps          # bulk gathers, not one per object


def test_lazy_desync_and_reseed():
    """an attribute written from outside, a foreign object in a list, another call order: the session ends, the plain path takes over with the
    same results, and the next adjust of a reference-order frame starts a new session"""
    g = pg.load("w4")
    loop, sc, fos, t0 = _loop(g)
    ex = loop.extractor
    seeds, started = 0, []
    for s in range(1, 11):
        if s == 4:        # the caller edits an object (writes the value it already has: results must not change)
            k = loop.state._candidates_kp[0]
            k.t_total = int(k.t_total)
            assert ex._lazy is None or not ex._lazy.alive
        if s == 7:        # a foreign (plain) object replaces a proxy in a list the caller hands in
            loop.state._landmarks_kp[3] = copy.deepcopy(loop.state._landmarks_kp[3])
        before = ex._lazy
        loop.step(sc["frames"][fos[t0 + s]])
        pg.assert_entries(pg.frame(g, s), pg.loop_entries(loop), "lazy/desync step %d" % s)
        if ex._lazy is not None and ex._lazy is not before:
            seeds += 1
            started.append(s)
    # frame 1; frame 4 (the write ends the session before the frame starts, the whole frame runs plain and re-seeds); frame 8 (in frame 7 the
    # session ends inside extend_landmarks, so that frame is not a complete plain frame; the next one is)
    assert started == [1, 4, 8] and ex._lazy.alive, started


def test_lazy_history_beyond_the_device_ring_is_complete():
    """`Keypoint.uv_history` is unbounded in the reference (state/keypoint.py:4-21); the device keeps the last 32 entries of a row.  Over 44 frames
    (histories of up to 47 entries: the first 15 have left the ring) the lazy classes hand out the WHOLE history of every keypoint -- state lists and
    dead lists (rows the device COPIED where the reference deep-copies), deep copies, and after the session has ended -- equal, entry for entry, to what the
    plain object path accumulates (the session archives what leaves the ring: `lazy._HistArchive`)."""
    from vo_mi355x import synthetic as syn
    sc = syn.sway_scene(48, w=256, h=160, f=260.0, seed=2024, pose_fn=lambda t: syn.sway_pose(t, period=24.0))

    def run(lazy, n):
        ctx = _oracle_ctx(256, 160)
        ctx.max_pts, ctx.batch = 2048, 1
        state, t1 = ph.gt_bootstrap(ctx, sc, 0, 3)
        loop = ph.ObjectLoop(ctx, sc["K"], state, sc["frames"][t1], t_step=1, ba_window=4, literal=True, lazy=lazy,
                             lazy_backend=lambda c, K, prm, ww, hh: ModelBackend(c, K, prm, ww, hh, cap=2048))
        for k in range(n):
            loop.step(sc["frames"][t1 + 1 + k])
        return loop

    n = 44
    a = run(True, n)
    b = run(False, n)
    sess = a.extractor._lazy
    assert sess is not None and sess.alive and len(sess._arch.blocks) >= 2

    def hist(k):
        return np.array([np.asarray(h, np.float64).reshape(2) for h in k.uv_history])

    def same(ka, kb, what):
        ha, hb = hist(ka), hist(kb)
        assert ha.shape == hb.shape and np.array_equal(ha, hb), (what, ha.shape, hb.shape)
        return len(ha)
    longest = 0
    for la, lb, what in ((a.state._landmarks_kp, b.state._landmarks_kp, "landmark keypoints"), (a.state._candidates_kp, b.state._candidates_kp, "candidates"),
                         (a.dead_kp, b.dead_kp, "dead keypoints")):
        assert len(la) == len(lb), what
        for ka, kb in zip(la, lb):
            longest = max(longest, same(ka, kb, what))
    assert longest > 40                                          # histories well beyond the 32-entry ring occurred
    old = [k for k in a.state._landmarks_kp if len(k.uv_history) > 36][:5]
    ref = [kb for ka, kb in zip(a.state._landmarks_kp, b.state._landmarks_kp) if len(ka.uv_history) > 36][:5]
    cp = copy.deepcopy(old)                                      # a copy of an attached proxy carries the archived part with it
    a.state._candidates_kp[0].t_total = int(a.state._candidates_kp[0].t_total)      # an outside write ends the session: every proxy becomes a plain object
    assert not sess.alive
    for ka, kc, kb in zip(old, cp, ref):
        same(ka, kb, "after the session ended"); same(kc, kb, "deep copy")
        assert not np.isnan(hist(ka)).any()


def test_lazy_off_is_the_plain_path():
    g = pg.load("w4")
    loop, sc, fos, t0 = _loop(g, lazy=False)
    for s in range(1, 4):
        loop.step(sc["frames"][fos[t0 + s]])
        pg.assert_entries(pg.frame(g, s), pg.loop_entries(loop), "plain step %d" % s)
    assert loop.extractor._lazy is None


def test_lazy_helpers():
    from vo_mi355x.lazy import InlierList, LazyList
    a = InlierList([0, 2, 5])
    assert 2 in a and 3 not in a and len(a) == 3 and a[1] == 2 and isinstance(a, list) and list(a) == [0, 2, 5]
    b = LazyList([1, 2]) + [3]
    assert b == [1, 2, 3]
    c = LazyList([object(), object()])
    c._copies = ["x", "y"]
    assert copy.deepcopy(c) == ["x", "y"]


@pytest.mark.parametrize("name,n_steps", [("w4", 12), ("groups", 5)])
def test_the_references_unmodified_pipeline_step_over_the_lazy_classes(name, n_steps):
    """build container only: /root/reference/src/pipeline/pipeline.py, imported as it lies, calls the lazy drop-in classes (ref_caller_check.py)"""
    import os
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/src/pipeline"):
        pytest.skip("the reference is not on this machine")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "ref_caller_check.py"), name, str(n_steps)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_history_archive_is_indexed_and_bounded_in_bytes():
    """`lazy._HistArchive`: a lookup touches only the passes that hold the row, host memory is bounded by a byte budget (oldest passes first:
    their entries come back as NaN), a recycled row's old tenant is told apart by its identity, a copy finds its source's record by identity"""
    from vo_mi355x import lazy
    HIST, EV = lazy.HIST, lazy.ARCH_EVERY
    dt = np.dtype([("t_first", np.int32), ("uv_first", np.float32, (2,)), ("hist_len", np.int32), ("hist", np.float32, (HIST, 2))])
    truth = lambda row, i: np.array([row * 1000.0 + i, -float(i)], np.float32)         # entry i of the keypoint living in `row`

    def recs(rows, n, t_first=0):
        r = np.zeros(len(rows), dt)
        for k, row in enumerate(rows):
            r["t_first"][k], r["uv_first"][k], r["hist_len"][k] = t_first, truth(row, 0), n
            for i in range(max(0, n - HIST), n):
                r["hist"][k][i % HIST] = truth(row, i)
        return r
    rows = np.arange(50)
    a = lazy._HistArchive()
    for n in range(EV, 10 * EV + 1, EV):                     # ten passes, every 16 frames
        a.add(rows, recs(rows, n))
    n = 10 * EV
    got = a.older(7, 0, truth(7, 0), n, recs([7], n)["hist"][0])
    assert got.shape == (n - HIST, 2) and np.array_equal(got, np.stack([truth(7, i) for i in range(n - HIST)]))
    assert len(a.by_row[7]) == 10 and a._relatives(7, 0, truth(7, 0)) == []
    # the row is recycled: a new keypoint (other birth frame) lives there; its lookups see its own entries, not the old tenant's
    new_tenant = recs([7], EV, t_first=200)
    new_tenant["hist"] += 0.5
    a.add(np.array([7]), new_tenant)
    r2 = recs([7], HIST + 4, t_first=200)
    r2["hist"] += 0.5
    got = a.older(7, 200, truth(7, 0), HIST + 4, r2["hist"][0])
    assert np.array_equal(got, np.stack([truth(7, i) + 0.5 for i in range(4)]))
    # a copy (row 60) of row 3's keypoint: no record of its own before the copy -> row 3's entries by identity
    cp = recs([3], n)
    a.add(np.array([60]), cp)
    got = a.older(60, 0, truth(3, 0), n, cp["hist"][0])
    assert np.array_equal(got, np.stack([truth(3, i) for i in range(n - HIST)]))
    # byte budget: only the newest passes stay
    b = lazy._HistArchive(budget_bytes=3 * (50 * EV * 2 * 4 + 50 * (8 + 8 + 8 + 8)))
    for n in range(EV, 10 * EV + 1, EV):
        b.add(rows, recs(rows, n))
    assert b.bytes <= b.budget and len(b.blocks) == 3 and b.first_id == 7
    got = b.older(7, 0, truth(7, 0), 10 * EV, recs([7], 10 * EV)["hist"][0])
    assert np.isnan(got[:7 * EV]).all() and np.array_equal(got[7 * EV:], np.stack([truth(7, i) for i in range(7 * EV, 10 * EV - HIST)]))
    assert len(b.by_row[7]) == 3                              # index entries of the passes that are gone were pruned on sight
