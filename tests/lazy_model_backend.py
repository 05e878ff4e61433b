"""TEST DOUBLE: the backend interface of vo_mi355x/lazy.py (`DeviceBackend`) over the table model of oracle/pipe_oracle.py, so that the
lazy object boundary -- proxies, list mirrors, validation, retirement, desynchronisation -- runs on the CPU (`-m "not gpu"`) against the
same goldens as everything else.  The numerical calls go to the VoContext look-alike the Extractor was given (tests/oracle_context.py)."""
import numpy as np

import pipe_oracle as po
from vo_mi355x.resident import ADJUST, DETECT, HIST, POSE, TRACK, TRACK_CANDIDATES, TRACK_LANDMARKS, TRIANGULATE, ResidentPipeline


class ModelBackend:
    def __init__(self, ctx, K, prm, width, height, cap=2048):
        self.N = cap
        self.m = po.PipeModel(ctx, K, width, height, cap=cap,
                              params=po.Params(ba_window=prm["ba_window"], min_track_length=prm["min_track_length"], mask_radius=prm["mask_radius"],
                                               max_new=prm["max_new"], max_reproj_err=prm["max_reproj_err"], min_bearing_angle=prm["min_bearing_angle"],
                                               ba_max_iters=prm["ba_max_iters"], ba_ftol=prm["ba_ftol"], ba_xtol=prm["ba_xtol"],
                                               min_distance=prm["min_kp_dist"]))
        self._img, self._mask = None, None
        self.stages = []

    def seed(self, state, dead, dead_kp, t_step):
        self.m.seed(state, dead, dead_kp, t_step)

    def push_frame(self, img):
        self._img = img                         # (the model pushes it into the context itself when it tracks)

    def stage(self, stages):
        m = self.m
        self.stages.append(stages)
        if stages & TRACK:
            m.info = {}
            m.track_points(self._img)
        halves = (1 if stages & TRACK_CANDIDATES else 0) | (2 if stages & TRACK_LANDMARKS else 0)
        if (stages & TRACK) or halves:
            m.extend(m._p1, halves or 3)
            if (halves or 3) & 2:
                self._mask = m.keep_mask
        if stages & POSE:
            m.localize()
            self._mask = getattr(m, "inlier_mask", None)
        if stages & TRIANGULATE and not m.status:
            m.triangulate()
        if stages & ADJUST and not m.status:
            m.adjust()
        if stages & DETECT and not m.status:
            m.detect()
            m.sweep()                           # rows are recycled once per frame, like the device behind VO_PIPE_KEEP_FREE_LISTS
        ba = m.info.get("ba") or {}
        H = m.poses.get(m.t, np.eye(4))
        return dict(status=m.status, overflow=m.info.get("overflow", 0), t=m.t, n_new=m.info.get("n_new", 0), n_resurrected=m.info.get("n_resurrected", 0),
                    n_detected=m.info.get("n_detected", 0), H=H, ba_cost0=ba.get("cost0", 0.0), ba_cost=ba.get("cost", 0.0), ba_iters=ba.get("iters", 0),
                    ba_accepted=ba.get("accepted", 0), ba_status=ba.get("status", 0), ba_observations=ba.get("n_obs", 0))

    def lists(self):
        m = self.m
        poses = np.zeros((HIST, 12))
        for t, H in m.poses.items():
            if m.t - HIST < t <= m.t:
                poses[t % HIST] = np.asarray(H, np.float64)[:3].reshape(12)
        a = lambda v: np.array(v, np.int32).reshape(-1)
        return dict(cand=a(m.cand), lm_l=a(m.lm_L), lm_k=a(m.lm_K), dead_l=a(m.dead_L), dead_k=a(m.dead_K), poses=poses, t=m.t, status=m.status)

    def mask(self, n):
        return np.asarray(self._mask[:n], bool)

    def rows(self, kind, rows):
        m = self.m
        rows = np.asarray(rows, np.int64)
        if kind == "K":
            out = np.zeros(len(rows), ResidentPipeline.K_ROW)
            out["t_first"], out["t_total"], out["hist_len"] = m.k_tf[rows], m.k_tt[rows], m.k_len[rows]
            out["uv"], out["uv_first"], out["hist"] = m.k_uv[rows], m.k_first[rows], m.k_hist[rows]
        else:
            out = np.zeros(len(rows), ResidentPipeline.L_ROW)
            out["t_latest"], out["p"] = m.l_tl[rows], m.l_p[rows]
        return out

    def close(self):
        pass
