"""Host side of the device-resident `Pipeline.step` (C ABI `vo_pipe_*`, csrc/vo_pipeline.hip).

The reference's per-frame state -- `State(landmarks, landmarks_kp, candidates_kp, trajectory)` plus the pipeline's two dead
lists (src/state/state.py:4-10, src/pipeline/pipeline.py:31) -- lives in device tables; `ResidentPipeline.step` enqueues one
frame of src/pipeline/pipeline.py:92-167 with no host synchronisation and `fetch` returns the small per-frame records.

    rp = ResidentPipeline(ctx, K, ba_window=4)
    rp.seed(state, landmarks_dead, landmarks_kp_dead, t_step=1)      # the bootstrap's objects (pipeline.py:42-90) -> tables
    ctx.push_frame(im_prev)                                          # = extractor._im_prev (pipeline.py:36)
    for im in frames:
        ctx.push_frame(im); rp.step()                                # or rp.step(frame_idx) on an uploaded sequence
        rec = rp.fetch()
    state, dead, dead_kp = rp.objects()                              # tables -> the reference's objects, sharing included

Object identity is part of the reference's behaviour (BundleAdjuster.adjust appends dead landmarks to the state's lists
without copying them, bundle_adjuster.py:142-147): the tables are rows of Keypoint / Landmark OBJECTS plus ordered lists of row
indices, and `seed` / `objects` translate between the two forms keeping who-shares-what.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PipeParams, PipeRecord
from .state import Keypoint, Landmark, State, Trajectory

HIST = 32
TRACK, POSE, TRIANGULATE, ADJUST, DETECT, ALL = 1, 2, 4, 8, 16, 31
TRACK_CANDIDATES, TRACK_LANDMARKS, KEEP_FREE_LISTS = 32, 64, 128      # the halves of TRACK's bookkeeping as the reference calls them (include/vo_mi355x.h)
LOST, CAPACITY, GROUPS = 1, 2, 4
# table ids (include/vo_mi355x.h) -> (dtype, per-sequence shape as a function of N = max_pts, R = 4 N)
_TABLES = {
    "k_tfirst": (0, np.int32, lambda N, R: (R,)), "k_ttotal": (1, np.int32, lambda N, R: (R,)), "k_histlen": (2, np.int32, lambda N, R: (R,)),
    "k_uv": (3, np.float32, lambda N, R: (R, 2)), "k_uvfirst": (4, np.float32, lambda N, R: (R, 2)), "k_hist": (5, np.float32, lambda N, R: (HIST, R, 2)),
    "l_tlatest": (6, np.int32, lambda N, R: (R,)), "l_p": (7, np.float64, lambda N, R: (R, 3)),
    "cand": (8, np.int32, lambda N, R: (N,)), "lm_l": (9, np.int32, lambda N, R: (N,)), "lm_k": (10, np.int32, lambda N, R: (N,)),
    "lm_kshared": (11, np.int32, lambda N, R: (N,)), "dead_l": (12, np.int32, lambda N, R: (N,)), "dead_k": (13, np.int32, lambda N, R: (N,)),
    "counts": (14, np.int32, lambda N, R: (32,)), "poses": (15, np.float64, lambda N, R: (HIST, 12)),
}
_RECORD_FIELDS = [n for n, _ in PipeRecord._fields_ if n not in ("H", "H_final")]


class ResidentPipeline:
    """ba_max_iters: the LM's own iteration cap; ba_budget: LM iterations ENQUEUED per frame (default = ba_max_iters, which never cuts a
    solve).  An enqueued iteration whose solve has already stopped exits at once but still costs its four launches (~16 us per frame for
    one sequence, ~45 us for a batch of 32), so a caller that tracks one camera may pass the few iterations a warm-started window needs
    (2-3 typical; e.g. ba_budget=6) and watch rec["ba_done"]: 0 flags a solve the budget cut (set_ba_budget raises it between frames).
    resurrect=False leaves dead landmarks dead (not the reference; see include/vo_mi355x.h)."""

    def __init__(self, ctx, K, ba_window=4, min_track_length=3, mask_radius=7, max_new=1000, max_reproj_err=2.0, min_bearing_angle=0.5,
                 ba_max_iters=50, ba_budget=None, ba_ftol=1e-3, ba_xtol=1e-3, pnp_blind_batches=4, pnp_seed=0, min_kp_dist=7, resurrect=True):
        self.ctx, self._L = ctx, ctx._L
        B = ctx.batch
        K = np.ascontiguousarray(np.broadcast_to(np.asarray(K, np.float64).reshape(-1, 3, 3), (B, 3, 3)))
        self.K = K
        p = PipeParams()
        ctx._ck(self._L.vo_pipe_default_params(C.byref(p)))
        p.ba_window, p.min_track_length, p.mask_radius, p.max_new = ba_window, min_track_length, mask_radius, max_new
        p.max_reproj_err, p.min_bearing_angle, p.pnp_blind_batches = max_reproj_err, min_bearing_angle, pnp_blind_batches
        p.ba.max_iters, p.ba.ftol, p.ba.xtol = ba_max_iters, ba_ftol, ba_xtol
        p.ba_budget = ba_max_iters if ba_budget is None else ba_budget
        p.resurrect = 1 if resurrect else 0
        p.pnp.reproj_err, p.pnp.seed = max_reproj_err, pnp_seed
        p.st.min_distance = float(min_kp_dist)
        self.params = p
        ctx._ck(self._L.vo_pipe_create(ctx._h, K.ctypes.data_as(C.POINTER(C.c_double)), C.byref(p)))
        self.N, self.R, self.B = ctx.max_pts, 4 * ctx.max_pts, B
        self.ba_window = ba_window
        self._inflight = 0
        ctx._ba_shape = (ba_window, ctx.max_pts)          # VoContext.ba_probe reads the resident problem the ADJUST stage builds

    # ---- tables ---------------------------------------------------------------------------------
    def read_table(self, name):
        tid, dt, shp = _TABLES[name]
        a = np.empty((self.B,) + shp(self.N, self.R), dt)
        self.ctx._ck(self._L.vo_pipe_table_read(self.ctx._h, tid, a.ctypes.data_as(C.c_void_p)))
        return a

    def write_table(self, name, a):
        tid, dt, shp = _TABLES[name]
        a = np.ascontiguousarray(a, dt)
        assert a.shape == (self.B,) + shp(self.N, self.R), (name, a.shape)
        self.ctx._ck(self._L.vo_pipe_table_write(self.ctx._h, tid, a.ctypes.data_as(C.c_void_p)))

    def read_tables(self):
        return {n: self.read_table(n) for n in _TABLES}

    # ---- stepping -------------------------------------------------------------------------------
    def step(self, frame_idx=-1, stages=ALL):
        """enqueue one frame (frame_idx of the uploaded sequence, or -1: the caller has pushed it); returns at once"""
        self.ctx._ck(self._L.vo_pipe_step(self.ctx._h, int(frame_idx), int(stages)))
        self._inflight += 1

    def step_host(self, frames, stages=ALL):
        """enqueue one frame whose images the HOST hands over (one uint8 [h, w] array per sequence, or the tuple `VoContext.host_frames` returns):
        Pipeline.step(img) of the reference (pipeline.py:98,171-172).  The upload runs on the copy stream, pyramid + tracking on the side stream.
        Arrays over page-locked memory (`VoContext.host_alloc`) must stay untouched until the step has been fetched (they are kept referenced)."""
        ptrs, stride, frames = frames if isinstance(frames, tuple) and len(frames) == 3 and isinstance(frames[1], int) else self.ctx.host_frames(frames)
        self.ctx._ck(self._L.vo_pipe_step_host(self.ctx._h, ptrs, int(stride), int(stages)))
        self._inflight += 1
        self._host_refs = (getattr(self, "_host_refs", []) + [frames])[-4:]

    def fetch(self):
        """records of the OLDEST step in flight: list of dicts, one per sequence (a dict when batch == 1)"""
        rec = (PipeRecord * self.B)()
        self.ctx._ck(self._L.vo_pipe_fetch(self.ctx._h, rec))
        self._inflight -= 1
        out = []
        for r in rec:
            d = {n: getattr(r, n) for n in _RECORD_FIELDS}
            H = np.eye(4); H[:3] = np.array(r.H[:]).reshape(3, 4)
            d["H"] = H
            Hf = np.eye(4); Hf[:3] = np.array(r.H_final[:]).reshape(3, 4)
            d["H_final"] = Hf                      # pose of step d["t_final"]: out of the window now, final
            out.append(d)
        return out[0] if self.B == 1 else out

    def fetch_raw(self):
        """the OLDEST step's records as the ctypes array (no dicts, no arrays: the per-stage caller of lazy.py reads four or five fields)"""
        rec = (PipeRecord * self.B)()
        self.ctx._ck(self._L.vo_pipe_fetch(self.ctx._h, rec))
        self._inflight -= 1
        return rec

    # ---- read-backs of the object boundary (lazy.py): the lists in one copy, object rows by index, the consensus mask -------------
    _LISTS = ("cand", "lm_l", "lm_k", "lm_kshared", "dead_l", "dead_k", "counts", "poses")
    K_ROW = np.dtype([("t_first", "<i4"), ("t_total", "<i4"), ("hist_len", "<i4"), ("pad", "<i4"), ("uv", "<f4", 2), ("uv_first", "<f4", 2),
                      ("hist", "<f4", (HIST, 2))])
    L_ROW = np.dtype([("t_latest", "<i4"), ("pad", "<i4"), ("p", "<f8", 3)])

    def read_lists(self):
        """-> dict(cand, lm_l, lm_k, lm_kshared, dead_l, dead_k [B, N] i32, counts [B, 32] i32, poses [B, 32, 12] f64): ONE device-to-host copy"""
        if not hasattr(self, "_lists_buf"):
            nb = C.c_uint64()
            self.ctx._ck(self._L.vo_pipe_lists_bytes(self.ctx._h, C.byref(nb)))
            self._lists_buf = np.empty(nb.value, np.uint8)
            off, self._lists_off = 0, {}
            for n in self._LISTS:
                _, dt, shp = _TABLES[n]
                size = int(np.prod((self.B,) + shp(self.N, self.R))) * np.dtype(dt).itemsize
                self._lists_off[n] = (off, size, dt, (self.B,) + shp(self.N, self.R))
                off += (size + 255) & ~255
            assert off == nb.value, (off, nb.value)
        self.ctx._ck(self._L.vo_pipe_lists_read(self.ctx._h, self._lists_buf.ctypes.data_as(C.c_void_p)))
        return {n: self._lists_buf[o:o + sz].view(dt).reshape(shape) for n, (o, sz, dt, shape) in self._lists_off.items()}

    def read_rows(self, kind, rows):
        """object rows by index (batch 1): kind 'K' -> structured array K_ROW (history in ring-slot order), 'L' -> L_ROW"""
        if self.B != 1:
            raise ValueError("read_rows copies the rows of ONE sequence (vo_pipe_rows_read moves [batch][n] records): batch is %d" % self.B)
        rows = np.ascontiguousarray(rows, np.int32).reshape(-1)
        dt = self.K_ROW if kind == "K" else self.L_ROW
        out = np.empty(len(rows), dt)
        for i in range(0, len(rows), self.N):
            part = rows[i:i + self.N]
            self.ctx._ck(self._L.vo_pipe_rows_read(self.ctx._h, 0 if kind == "K" else 1, part.ctypes.data_as(C.POINTER(C.c_int32)), len(part),
                                                   out[i:i + len(part)].ctypes.data_as(C.c_void_p)))
        return out

    def read_inliers(self, n):
        """consensus mask of the last POSE stage over the first n entries of the landmark list as it was before the pruning (batch 1)"""
        if self.B != 1:
            raise ValueError("read_inliers copies the mask of ONE sequence (vo_pipe_inliers_read moves [batch][n] bytes): batch is %d" % self.B)
        m = np.zeros(max(n, 1), np.uint8)
        self.ctx._ck(self._L.vo_pipe_inliers_read(self.ctx._h, m.ctypes.data_as(C.POINTER(C.c_uint8)), int(n)))
        return m[:n].astype(bool)

    def set_ba_budget(self, budget):
        self.ctx._ck(self._L.vo_pipe_set_ba_budget(self.ctx._h, int(budget)))

    # ---- objects -> tables (Pipeline._get_init_state's State, or any later state) ------------------------------------
    def seed(self, states, dead=None, dead_kp=None, t_step=1):
        """states: State (batch 1) or one per sequence; dead / dead_kp: the pipeline's dead lists (default empty)"""
        B, N, R = self.B, self.N, self.R
        states = [states] if B == 1 and not isinstance(states, (list, tuple)) else list(states)
        dead = [[] for _ in range(B)] if dead is None else ([dead] if B == 1 and (not dead or not isinstance(dead[0], (list, tuple))) else dead)
        dead_kp = [[] for _ in range(B)] if dead_kp is None else ([dead_kp] if B == 1 and (not dead_kp or not isinstance(dead_kp[0], (list, tuple))) else dead_kp)
        T = {n: np.zeros((B,) + shp(N, R), dt) for n, (_, dt, shp) in _TABLES.items()}
        for b in range(B):
            st = states[b]
            kmap, lmap = {}, {}

            def K_of(k):
                i = kmap.get(id(k))
                if i is None:
                    i = kmap[id(k)] = len(kmap)
                    if i >= R:
                        raise ValueError("more keypoint objects than table rows")
                    h = np.array(k.uv_history, np.float64).reshape(-1, 2)
                    T["k_tfirst"][b, i], T["k_ttotal"][b, i], T["k_histlen"][b, i] = k.t_first, k.t_total, len(h)
                    T["k_uv"][b, i], T["k_uvfirst"][b, i] = np.asarray(k.uv).reshape(2), np.asarray(k.uv_first).reshape(2)
                    for idx in range(max(0, len(h) - HIST), len(h)):
                        T["k_hist"][b, idx % HIST, i] = h[idx]
                return i

            def L_of(l):
                i = lmap.get(id(l))
                if i is None:
                    i = lmap[id(l)] = len(lmap)
                    T["l_tlatest"][b, i], T["l_p"][b, i] = l.t_latest, np.asarray(l.p, np.float64).reshape(3)
                return i
            n_l, n_c, n_d = len(st._landmarks), len(st._candidates_kp), len(dead[b])
            if n_l + n_c > N or n_d > N:
                raise ValueError("state does not fit max_pts = %d" % N)
            T["lm_l"][b, :n_l] = [L_of(l) for l in st._landmarks]
            T["lm_k"][b, :n_l] = [K_of(k) for k in st._landmarks_kp]
            T["cand"][b, :n_c] = [K_of(k) for k in st._candidates_kp]
            T["dead_l"][b, :n_d] = [L_of(l) for l in dead[b]]
            T["dead_k"][b, :n_d] = [K_of(k) for k in dead_kp[b]]
            dk = set(T["dead_k"][b, :n_d].tolist())
            T["lm_kshared"][b, :n_l] = [1 if k in dk else 0 for k in T["lm_k"][b, :n_l].tolist()]
            T["counts"][b, :6] = [n_c, n_l, n_d, 0, 0, t_step]
            for t, H in st._trajectory._poses.items():
                if t_step - HIST < t <= t_step:
                    T["poses"][b, t % HIST] = np.asarray(H, np.float64)[:3].reshape(12)
        for n, a in T.items():
            self.write_table(n, a)
        self.ctx._ck(self._L.vo_pipe_commit(self.ctx._h))

    # ---- tables -> objects ----------------------------------------------------------------------------------------------
    def entries(self, b=0, tables=None):
        """list-order dump of sequence b for comparisons: dict(cand, lm, dead = lists of (t_latest | None, p | None, t_first, t_total,
        uv_first, uv, hist_len, last min(len, 32) history entries oldest first), n_dead_total, t, status, poses {t: H 4x4})"""
        T = tables or self.read_tables()
        cn = T["counts"][b]
        n_c, n_l, n_d, n_inert, status, t = [int(x) for x in cn[:6]]

        def entry(L, K):
            n = int(T["k_histlen"][b, K])
            hist = np.array([T["k_hist"][b, i % HIST, K] for i in range(max(0, n - HIST), n)], np.float32).reshape(-1, 2)
            return (None if L is None else int(T["l_tlatest"][b, L]), None if L is None else T["l_p"][b, L].copy(), int(T["k_tfirst"][b, K]),
                    int(T["k_ttotal"][b, K]), T["k_uvfirst"][b, K].copy(), T["k_uv"][b, K].copy(), n, hist)
        poses = {}
        for tt in range(max(0, t - HIST + 1), t + 1):
            H = np.eye(4); H[:3] = T["poses"][b, tt % HIST].reshape(3, 4)
            poses[tt] = H
        return dict(cand=[entry(None, int(k)) for k in T["cand"][b, :n_c]],
                    lm=[entry(int(l), int(k)) for l, k in zip(T["lm_l"][b, :n_l], T["lm_k"][b, :n_l])],
                    dead=[entry(int(l), int(k)) for l, k in zip(T["dead_l"][b, :n_d], T["dead_k"][b, :n_d])],
                    n_dead_total=n_d + n_inert, t=t, status=status, poses=poses,
                    rows=dict(lm_l=T["lm_l"][b, :n_l].copy(), lm_k=T["lm_k"][b, :n_l].copy(), dead_l=T["dead_l"][b, :n_d].copy(),
                              dead_k=T["dead_k"][b, :n_d].copy(), lm_kshared=T["lm_kshared"][b, :n_l].copy()))

    def objects(self, b=0):
        """-> (State, landmarks_dead, landmarks_kp_dead) as the reference's objects; entries that share a table row share the object.
        Histories hold the last 32 positions (what the window can reach); dead entries dropped as inert are not materialised."""
        T = self.read_tables()
        cn = T["counts"][b]
        n_c, n_l, n_d, _, _, t = [int(x) for x in cn[:6]]
        kobj, lobj = {}, {}
        zero = np.zeros((1, 1))

        def K_of(i):
            if i not in kobj:
                n = int(T["k_histlen"][b, i])
                hist = [T["k_hist"][b, j % HIST, i].astype(np.float32).reshape(2, 1) for j in range(max(0, n - HIST), n)]
                kobj[i] = Keypoint(int(T["k_tfirst"][b, i]), int(T["k_ttotal"][b, i]), T["k_uvfirst"][b, i].reshape(2, 1).copy(),
                                   T["k_uv"][b, i].reshape(2, 1).copy(), zero.copy(), hist)
            return kobj[i]

        def L_of(i):
            if i not in lobj:
                lobj[i] = Landmark(int(T["l_tlatest"][b, i]), T["l_p"][b, i].reshape(3, 1).copy(), zero.copy())
            return lobj[i]
        traj = Trajectory({})
        for tt in range(max(0, t - HIST + 1), t + 1):
            H = np.eye(4); H[:3] = T["poses"][b, tt % HIST].reshape(3, 4)
            traj.append(tt, H)
        st = State([L_of(int(i)) for i in T["lm_l"][b, :n_l]], [K_of(int(i)) for i in T["lm_k"][b, :n_l]],
                   [K_of(int(i)) for i in T["cand"][b, :n_c]], traj)
        return st, [L_of(int(i)) for i in T["dead_l"][b, :n_d]], [K_of(int(i)) for i in T["dead_k"][b, :n_d]]
