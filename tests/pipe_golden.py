"""G5 goldens -- the reference's OWN `Pipeline.step` (src/pipeline/pipeline.py:92-167), imported unmodified by
tests/golden/gen_golden.py and dumped after every frame -- decoded for the closed-loop tests: the seed state as drop-in objects
(with the reference's object sharing), and, per frame, the lists in the tuple form `pipe_helpers.compare_lists` /
`PipeModel.entry` / `ResidentPipeline.entries` use, plus who-shares-which-object as first-appearance numbers."""
import hashlib
import os

import numpy as np

from vo_mi355x import synthetic as syn
from vo_mi355x.state import Keypoint, Landmark, State, Trajectory

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HIST = 32


def load(name):
    return dict(np.load(os.path.join(GOLDEN, "pipe_%s.npz" % name)))


def scene_frames(g):
    """the frames the generator fed the reference (re-rendered from the seed; the file holds their SHA-256)"""
    n = int(g["frame_of_step"].max()) + 1
    sc = syn.sway_scene(n, w=int(g["w"]), h=int(g["h"]), f=260.0, seed=2024, pose_fn=lambda t: syn.sway_pose(t, amp=(0.9, 0.25, -0.5), period=24.0))
    sha = np.frombuffer(hashlib.sha256(sc["frames"].tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, g["frames_sha256"]), "the rendered scene differs from the one the golden was generated on"
    return sc


def _kps(g, prefix):
    lens = g[prefix + "hist_len"]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return [dict(t_first=int(g[prefix + "t_first"][i]), t_total=int(g[prefix + "t_total"][i]), uv_first=g[prefix + "uv_first"][i], uv=g[prefix + "uv"][i],
                 hist=g[prefix + "hist"][off[i]:off[i + 1]]) for i in range(len(lens))]


def seed_objects(g):
    """the seed dump (step 0) -> (State, dead, dead_kp) of drop-in objects; entries with equal ids share one object"""
    p = "s0_"
    kobj, lobj = {}, {}

    def K_of(kid, d):
        if kid not in kobj:
            hist = [np.float32(h).reshape(2, 1) for h in d["hist"]]
            kobj[kid] = Keypoint(d["t_first"], d["t_total"], np.float32(d["uv_first"]).reshape(2, 1), np.float32(d["uv"]).reshape(2, 1), np.zeros((1, 1)), hist)
        return kobj[kid]

    def L_of(lid, tl, pos):
        if lid not in lobj:
            lobj[lid] = Landmark(int(tl), np.float64(pos).reshape(3, 1).copy(), np.zeros((1, 1)))
        return lobj[lid]
    lm = [L_of(i, t, q) for i, t, q in zip(g[p + "lm_Lid"], g[p + "lm_t_latest"], g[p + "lm_p"])]
    dead = [L_of(i, t, q) for i, t, q in zip(g[p + "dead_Lid"], g[p + "dead_t_latest"], g[p + "dead_p"])]
    lmk = [K_of(i, d) for i, d in zip(g[p + "lm_Kid"], _kps(g, p + "lmk_"))]
    cand = [K_of(i, d) for i, d in zip(g[p + "cand_Kid"], _kps(g, p + "cand_"))]
    deadk = [K_of(i, d) for i, d in zip(g[p + "dead_Kid"], _kps(g, p + "deadk_"))]
    traj = Trajectory({})
    for t, H in enumerate(g[p + "traj"]):
        traj.append(t, H.copy())
    return State(lm, lmk, cand, traj), dead, deadk


def _first_appearance(*lists):
    ids, out = {}, []
    for lst in lists:
        out.append([ids.setdefault(int(v), len(ids)) for v in lst])
    return out


def frame(g, s, hist_depth=HIST):
    """step s of the golden -> dict(cand, lm, dead = entry tuples in list order; n_dead_total; share = (L numbers of lm + dead, K numbers
    of lm + cand + dead) by first appearance; traj [t + 1, 4, 4]).  The dead list is the FILTERED one the tables keep: an entry stays
    while its track lies inside the window or its Landmark object is in the state's list (oracle/pipe_oracle.py header)."""
    p = "s%d_" % s
    W = int(g["ba_window"])
    t = int(g["t_step0"]) + s
    cand, lmk, deadk = _kps(g, p + "cand_"), _kps(g, p + "lmk_"), _kps(g, p + "deadk_")

    def entry(tl, pos, k):
        n = len(k["hist"])
        return (tl, pos, k["t_first"], k["t_total"], np.float64(k["uv_first"]), np.float64(k["uv"]), n, np.float64(k["hist"][max(0, n - hist_depth):]))
    live = set(int(v) for v in g[p + "lm_Lid"])
    keep = [(t - (int(tl) - (len(k["hist"]) - 1))) < W or int(lid) in live
            for tl, k, lid in zip(g[p + "dead_t_latest"], deadk, g[p + "dead_Lid"])] if s > 0 else [True] * len(deadk)
    kd = np.nonzero(keep)[0]
    share = (_first_appearance(g[p + "lm_Lid"], g[p + "dead_Lid"][kd]), _first_appearance(g[p + "lm_Kid"], g[p + "cand_Kid"], g[p + "dead_Kid"][kd]))
    return dict(cand=[entry(None, None, k) for k in cand],
                lm=[entry(int(tl), np.float64(q), k) for tl, q, k in zip(g[p + "lm_t_latest"], g[p + "lm_p"], lmk)],
                dead=[entry(int(g[p + "dead_t_latest"][i]), np.float64(g[p + "dead_p"][i]), deadk[i]) for i in kd],
                n_dead_total=len(deadk), share=share, traj=g[p + "traj"], t=t)


def assert_entries(ref, got, what, p_tol=0.0, pose_tol=0.0):
    """ref: frame(g, s); got: dict(cand, lm, dead, n_dead_total[, share][, poses {t: H}]).  Integers and float32 pixel coordinates exact,
    landmark positions within p_tol (relative; 0 = bit-equal), poses within pose_tol (absolute)"""
    assert got["n_dead_total"] == ref["n_dead_total"], (what, "dead total", got["n_dead_total"], ref["n_dead_total"])
    for name in ("cand", "lm", "dead"):
        a, b = ref[name], got[name]
        assert len(a) == len(b), (what, name, "length", len(a), len(b))
        for i, (ea, eb) in enumerate(zip(a, b)):
            assert (ea[2], ea[3], ea[6]) == (eb[2], eb[3], eb[6]), (what, name, i, "t_first / t_total / history length", ea[2:4] + (ea[6],), eb[2:4] + (eb[6],))
            assert np.array_equal(ea[4], np.float64(eb[4])) and np.array_equal(ea[5], np.float64(eb[5])), (what, name, i, "uv", ea[5], eb[5])
            assert np.array_equal(ea[7], np.float64(eb[7])), (what, name, i, "history")
            if ea[0] is not None:
                assert ea[0] == eb[0], (what, name, i, "t_latest", ea[0], eb[0])
                pa, pb = ea[1], np.float64(eb[1]).reshape(3)
                if p_tol == 0.0:
                    assert np.array_equal(pa, pb), (what, name, i, "p", pa, pb)
                else:
                    assert np.linalg.norm(pa - pb) <= p_tol * max(np.linalg.norm(pa), 1e-12), (what, name, i, "p", pa, pb)
    if "share" in got:
        assert got["share"][0] == ref["share"][0], (what, "Landmark objects shared differently")
        assert got["share"][1] == ref["share"][1], (what, "Keypoint objects shared differently")
    if "poses" in got:
        for t, H in got["poses"].items():
            d = np.abs(np.asarray(H)[:3] - ref["traj"][t][:3]).max()
            assert d <= pose_tol, (what, "pose", t, d)


# ---- the three implementations in the common form ----------------------------------------------------------------------------------
def loop_entries(loop, hist_depth=HIST):
    """tests/pipe_helpers.ObjectLoop (Pipeline.step over the drop-in classes)"""
    st, W, t = loop.state, loop.ba_window, loop.t_step

    def entry(l, k):
        h = np.array(k.uv_history, np.float64).reshape(-1, 2)
        return (None if l is None else int(l.t_latest), None if l is None else np.asarray(l.p, np.float64).reshape(3), int(k.t_first), int(k.t_total),
                np.asarray(k.uv_first, np.float64).reshape(2), np.asarray(k.uv, np.float64).reshape(2), len(h), h[max(0, len(h) - hist_depth):])
    live = {id(l) for l in st._landmarks}
    dead = [(l, k) for l, k in zip(loop.dead, loop.dead_kp) if (t - (l.t_latest - (len(k.uv_history) - 1))) < W or id(l) in live]
    share = (_first_appearance([id(l) for l in st._landmarks], [id(l) for l, _ in dead]),
             _first_appearance([id(k) for k in st._landmarks_kp], [id(k) for k in st._candidates_kp], [id(k) for _, k in dead]))
    return dict(cand=[entry(None, k) for k in st._candidates_kp], lm=[entry(l, k) for l, k in zip(st._landmarks, st._landmarks_kp)],
                dead=[entry(l, k) for l, k in dead], n_dead_total=len(loop.dead), share=share,
                poses={t_: st._trajectory[t_] for t_ in range(len(st._trajectory))})


def model_entries(m):
    """oracle/pipe_oracle.PipeModel"""
    share = (_first_appearance(m.lm_L, m.dead_L), _first_appearance(m.lm_K, m.cand, m.dead_K))
    return dict(cand=[m.entry(None, k) for k in m.cand], lm=[m.entry(l, k) for l, k in zip(m.lm_L, m.lm_K)],
                dead=[m.entry(l, k) for l, k in zip(m.dead_L, m.dead_K)], n_dead_total=len(m.dead_L) + m.n_dead_inert, share=share,
                poses={t: H for t, H in m.poses.items()})


def device_entries(rp, b=0):
    """vo_mi355x.resident.ResidentPipeline (the device tables read back)"""
    e = rp.entries(b)
    r = e["rows"]
    e["share"] = (_first_appearance(r["lm_l"], r["dead_l"]), _first_appearance(r["lm_k"], [], r["dead_k"]))
    # (the candidates' rows are not in `rows`; a candidate keypoint is never shared, so number them after the landmarks' like the golden does)
    T = rp.read_table("cand")[b][:len(e["cand"])]
    e["share"] = (e["share"][0], _first_appearance(r["lm_k"], T, r["dead_k"]))
    return e
