"""Run by tests/test_lazy.py::test_the_references_unmodified_pipeline_step_over_the_lazy_classes in a child process, in the build container only
(it imports /root/reference/src/pipeline/pipeline.py, which never travels): the reference's OWN `Pipeline.step`, unmodified, calling
vo_mi355x.Extractor / BundleAdjuster (the import swap of INTEGRATION.md section 1, done here by handing the two objects to a Pipeline made
with __new__), lazy boundary on, tables = the CPU model.  After every frame the Pipeline's own lists must equal the G5 golden."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "ref_stub"), "/root/reference/src", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "visual-odom-pipeline_amd"),
                os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

visu = types.ModuleType("visu")
visu.Visualizer = type("Visualizer", (), {"__init__": lambda self, *a, **k: None, "update": lambda self, *a, **k: None, "render": lambda self, *a, **k: None})
sys.modules["visu"] = visu
import pipeline.pipeline as ref_pipe_mod  # noqa: E402   (the reference's file, as it lies)

import pipe_golden as pg  # noqa: E402
from lazy_model_backend import ModelBackend  # noqa: E402
from oracle_context import OracleContext  # noqa: E402
from vo_mi355x import BundleAdjuster, Extractor  # noqa: E402


def main(name, n_steps):
    g = pg.load(name)
    sc = pg.scene_frames(g)
    w, h, W, t0 = int(g["w"]), int(g["h"]), int(g["ba_window"]), int(g["t_step0"])
    fos = g["frame_of_step"]
    ctx = OracleContext(w, h)
    ctx.max_pts, ctx.batch = 2048, 1
    state, dead, dead_kp = pg.seed_objects(g)

    class FakeLoader:
        _name = "synthetic"

        def getFrame(self, t):
            return sc["frames"][t], None

    pl = ref_pipe_mod.Pipeline.__new__(ref_pipe_mod.Pipeline)
    pl._loader, pl._K = FakeLoader(), sc["K"]
    pl._ba, pl._ba_window_size, pl._ba_frequency, pl._min_kp_dist = True, W, 1, 7
    pl._max_bidir_error, pl._max_reprojection_error, pl._min_landmark_angle, pl._kp_method = np.inf, 2.0, 0.5, 'shi-tomasi'
    pl._extractor = Extractor(min_kp_dist=7, ctx=ctx, lazy=True, lazy_backend=lambda c, K, prm, ww, hh: ModelBackend(c, K, prm, ww, hh, cap=2048))
    pl._bundle_adjuster = BundleAdjuster(verbosity=0, window_size=W, method='trf', xtol=1e-3, ftol=1e-3, ctx=ctx)
    pl._visu = visu.Visualizer()
    pl._t_step, pl._landmarks_dead, pl._landmarks_kp_dead, pl._state = t0, dead, dead_kp, state
    pl._t_loader = int(fos[t0])
    pl._extractor._im_prev = sc["frames"][pl._t_loader]
    view = types.SimpleNamespace(ba_window=W)
    for s in range(1, n_steps + 1):
        assert int(fos[t0 + s]) == pl._t_loader + 1
        pl.step()
        view.state, view.dead, view.dead_kp, view.t_step = pl._state, pl._landmarks_dead, pl._landmarks_kp_dead, pl._t_step
        pg.assert_entries(pg.frame(g, s), pg.loop_entries(view), "reference caller, lazy classes, %s step %d" % (name, s))
        sess = pl._extractor._lazy
        assert sess is not None and sess.alive and sess.stats["fast"] == (1 if s == 1 else 1 + 6 * (s - 1)), (s, sess and sess.stats)
    print("OK %s: %d frames of the reference's Pipeline.step over the lazy classes = golden; %d fast calls, %d gathers" % (name, n_steps, sess.stats["fast"], sess.stats["gathers"]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
