"""The lazy object boundary on the device (vo_mi355x/lazy.py over csrc/vo_pipeline.hip): the reference's call sequence, written as
pipeline.py:92-167 writes it (`if i in inliers` on the returned list), over vo_mi355x.Extractor / BundleAdjuster with a GPU context.  Frame 1
takes the plain path (arrays gathered from / scattered to Python objects) and seeds the device tables at the end of `adjust`; from then on
every call is one stage of vo_pipe_step and the caller's lists are views.  After every frame the caller's objects, read back through the
proxies, must equal the reference's own run (G5 goldens): integers and float32 pixels exact, float64 quantities to the closed loop's tolerance
(tests/test_gpu_pipe_golden.py)."""
import numpy as np
import pytest

import pipe_golden as pg
import pipe_helpers as ph
from test_gpu_pipe_golden import P_TOL, POSE_TOL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n_steps", [("w4", 12), ("w10", 10), ("groups", 5)])
def test_lazy_views_of_the_device_tables_equal_the_references_own_pipeline_step(name, n_steps):
    from vo_mi355x import VoContext
    from vo_mi355x.lazy import LazyKeypoint, LazyLandmark
    g = pg.load(name)
    sc = pg.scene_frames(g)
    w, h, W, t0 = int(g["w"]), int(g["h"]), int(g["ba_window"]), int(g["t_step0"])
    fos = g["frame_of_step"]
    with VoContext(w, h, max_pts=2048) as ctx:
        state, dead, dead_kp = pg.seed_objects(g)
        loop = ph.ObjectLoop(ctx, sc["K"], state, sc["frames"][fos[t0]], t_step=t0, ba_window=W, literal=True, lazy=True)
        loop.dead, loop.dead_kp = dead, dead_kp
        ex = loop.extractor
        for s in range(1, n_steps + 1):
            loop.step(sc["frames"][fos[t0 + s]])
            sess = ex._lazy
            assert sess is not None and sess.alive, (s, getattr(ex, "_lazy_error", None), sess and sess.reason)
            assert sess.stats["fast"] == (1 if s == 1 else 1 + 6 * (s - 1)), (s, sess.stats)
            pg.assert_entries(pg.frame(g, s), pg.loop_entries(loop), "lazy on the device, %s step %d" % (name, s), p_tol=P_TOL, pose_tol=POSE_TOL)
        st = loop.state
        assert all(type(l) is LazyLandmark for l in st._landmarks) and all(type(k) is LazyKeypoint for k in st._landmarks_kp + st._candidates_kp)


def test_lazy_desync_on_the_device_keeps_the_results():
    """a write from outside ends the session (every proxy becomes a plain object with its values), the plain path continues, the next complete
    frame seeds a new session -- results as before"""
    from vo_mi355x import VoContext
    g = pg.load("w4")
    sc = pg.scene_frames(g)
    w, h, W, t0 = int(g["w"]), int(g["h"]), int(g["ba_window"]), int(g["t_step0"])
    fos = g["frame_of_step"]
    with VoContext(w, h, max_pts=2048) as ctx:
        state, dead, dead_kp = pg.seed_objects(g)
        loop = ph.ObjectLoop(ctx, sc["K"], state, sc["frames"][fos[t0]], t_step=t0, ba_window=W, literal=True, lazy=True)
        loop.dead, loop.dead_kp = dead, dead_kp
        started, before = [], None
        for s in range(1, 9):
            if s == 4:
                k = loop.state._landmarks_kp[5]
                k.uv = np.array(k.uv)                      # same value, written from outside
            loop.step(sc["frames"][fos[t0 + s]])
            pg.assert_entries(pg.frame(g, s), pg.loop_entries(loop), "lazy/desync on the device, step %d" % s, p_tol=P_TOL, pose_tol=POSE_TOL)
            if loop.extractor._lazy is not None and loop.extractor._lazy is not before:
                started.append(s)
            before = loop.extractor._lazy
        assert started == [1, 4]
