"""Shared by the closed-loop pipeline tests: (i) the reference's Pipeline.step (pipeline.py:92-167) restated over the drop-in
Extractor / BundleAdjuster / State classes one step at a time, (ii) a rendered two-plane scene with a camera that sways back
and forth (so a sequence of any length stays in view) and a ground-truth bootstrap state for it, (iii) the object-by-object
comparison of that loop's lists with a table state (the array model of oracle/pipe_oracle.py or the device tables)."""
import copy

import numpy as np


from vo_mi355x.synthetic import gt_bootstrap, sway_pose, sway_scene as scene  # noqa: E402,F401  (the scene lives with the product's synthetic inputs: bench.py uses it too)


# ---------------------------------------------------------------------------------------------------------
# the reference's loop over the drop-in classes
# ---------------------------------------------------------------------------------------------------------
class ObjectLoop:
    """Pipeline.__init__ / step (pipeline.py:13-40, 92-167) over vo_mi355x.Extractor / BundleAdjuster"""

    def __init__(self, ctx, K, state, im_prev, t_step=1, ba_window=4, min_kp_dist=7, max_reproj=2.0, min_angle=0.5, ba_max_iters=50, literal=False,
                 **extractor_kw):
        """literal: `if i in inliers` on the list camera_pose returned, as pipeline.py:130 writes it (default: a set, which a plain list of
        3 000 inlier indices needs to stay out of O(n^2)); extractor_kw: lazy=..., lazy_backend=... (tests of vo_mi355x/lazy.py)"""
        from vo_mi355x import BundleAdjuster, Extractor
        self.K, self.state, self.t_step, self.literal = K, state, t_step, literal
        self.ba_window, self.min_kp_dist, self.max_reproj, self.min_angle = ba_window, min_kp_dist, max_reproj, min_angle
        extractor_kw.setdefault("lazy", False)
        self.extractor = Extractor(min_kp_dist=min_kp_dist, ctx=ctx, **extractor_kw)
        self.adjuster = BundleAdjuster(verbosity=0, window_size=ba_window, method='trf', xtol=1e-3, ftol=1e-3, ctx=ctx, max_iters=ba_max_iters)
        self.dead, self.dead_kp = [], []
        self.extractor._im_prev = im_prev
        self.info = {}

    def step(self, im):
        ex, st, K = self.extractor, self.state, self.K
        self.t_step += 1
        st._candidates_kp = ex.extend_tracks(im, st._candidates_kp, max_bidir_error=np.inf)
        st._landmarks, st._landmarks_kp, ld, lkd = ex.extend_landmarks(im, st._landmarks, st._landmarks_kp, max_bidir_error=np.inf)
        self.dead += copy.deepcopy(ld); self.dead_kp += copy.deepcopy(lkd)
        ex._im_prev = im.copy()
        inl, Hk = ex.camera_pose(K, st._landmarks, st._landmarks_kp, corr='3D-2D', max_err_reproj=self.max_reproj)
        inl_set = inl if self.literal else set(inl)
        lms, lkp = [], []
        for i in range(len(st._landmarks)):
            if i in inl_set:
                lms.append(st._landmarks[i]); lkp.append(st._landmarks_kp[i])
            else:
                self.dead.append(copy.deepcopy(st._landmarks[i])); self.dead_kp.append(copy.deepcopy(st._landmarks_kp[i]))
        st._landmarks, st._landmarks_kp = lms, lkp
        st._trajectory.append(self.t_step, Hk)
        l_new, lk_new, st._candidates_kp = ex.triangulate_tracks(K, st._candidates_kp, st._trajectory, t_curr=self.t_step, min_track_length=3,
                                                                 min_bearing_angle=self.min_angle, max_err_reproj=self.max_reproj, refine=True)
        st._landmarks_kp += lk_new; st._landmarks += l_new
        n_before = len(st._landmarks)
        self.state, self.dead, self.dead_kp = self.adjuster.adjust(st, self.dead, self.dead_kp, K, self.t_step)
        st = self.state
        new_c = ex.extract(im, self.t_step, st._landmarks_kp + st._candidates_kp, detector='shi-tomasi', mask_radius=self.min_kp_dist, describe=False)
        st._candidates_kp += new_c
        self.info = dict(n_inliers=len(inl), n_new=len(l_new), n_resurrected=len(st._landmarks) - n_before, n_detected=len(new_c),
                         ba=self.adjuster.last_stats)


# ---------------------------------------------------------------------------------------------------------
# comparison: object lists vs table state
# ---------------------------------------------------------------------------------------------------------
def _kp_fields(k, hist_depth):
    h = np.array(k.uv_history, np.float64).reshape(-1, 2)
    return (int(k.t_first), int(k.t_total), np.asarray(k.uv_first, np.float64).reshape(2), np.asarray(k.uv, np.float64).reshape(2), len(h),
            h[max(0, len(h) - hist_depth):])


def compare_lists(loop, table_entries, hist_depth=32, p_tol=0.0, what=""):
    """table_entries: dict(cand=[entry], lm=[entry], dead=[entry]) with entry = (t_latest | None, p | None, t_first, t_total, uv_first, uv,
    hist_len, hist tail) in list order (PipeModel.entry / the device read-back); loop: ObjectLoop.  Exact on integers and float32
    pixel coordinates, `p_tol` (relative) on landmark positions."""
    st = loop.state
    # the tables drop dead entries that can never be resurrected again (window test failed and the Landmark object is not in the
    # state's list: its t_latest is frozen, the test only fails harder) and count them; the same filter on the object side
    live = {id(l) for l in st._landmarks}
    dead = [(l, k) for l, k in zip(loop.dead, loop.dead_kp)
            if (loop.t_step - (l.t_latest - (len(k.uv_history) - 1))) < loop.ba_window or id(l) in live]
    if "n_dead_total" in table_entries:
        assert table_entries["n_dead_total"] == len(loop.dead), (what, table_entries["n_dead_total"], len(loop.dead))
    pairs = dict(cand=[(None, k) for k in st._candidates_kp], lm=list(zip(st._landmarks, st._landmarks_kp)), dead=dead)
    for name in ("cand", "lm", "dead"):
        ref, got = pairs[name], table_entries[name]
        assert len(ref) == len(got), (what, name, len(ref), len(got))
        for i, ((l, k), e) in enumerate(zip(ref, got)):
            tl, p, tf, tt, uvf, uv, n, hist = e
            rf = _kp_fields(k, hist_depth)
            assert (tf, tt, n) == (rf[0], rf[1], rf[4]), (what, name, i, (tf, tt, n), rf[:2] + (rf[4],))
            assert np.array_equal(np.float64(uvf), rf[2]) and np.array_equal(np.float64(uv), rf[3]), (what, name, i, uv, rf[3])
            assert np.array_equal(np.float64(hist), rf[5]), (what, name, i)
            if l is not None:
                assert tl == int(l.t_latest), (what, name, i, tl, l.t_latest)
                pr = np.asarray(l.p, np.float64).reshape(3)
                if p_tol == 0.0:
                    assert np.array_equal(p, pr), (what, name, i, p, pr)
                else:
                    assert np.linalg.norm(p - pr) <= p_tol * max(np.linalg.norm(pr), 1e-12), (what, name, i, p, pr)


def sharing_signature(loop):
    """how the loop's lists share objects: for every state landmark entry the index of the first dead entry holding the same Landmark /
    the same Keypoint object (-1: none) -- the structure the table model must reproduce for later frames to agree"""
    dl = {id(l): i for i, l in reversed(list(enumerate(loop.dead)))}
    dk = {id(k): i for i, k in reversed(list(enumerate(loop.dead_kp)))}
    st = loop.state
    return [(dl.get(id(l), -1), dk.get(id(k), -1)) for l, k in zip(st._landmarks, st._landmarks_kp)]
