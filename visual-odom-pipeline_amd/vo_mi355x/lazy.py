"""The reference's object interface as VIEWS of the device tables (SURVEY.md section 7 "keep SoA arrays and only materialise objects lazily").

`Pipeline.step` (src/pipeline/pipeline.py:92-167) talks to `Extractor` / `BundleAdjuster` in lists of `Keypoint` / `Landmark` objects
(src/extractor/extractor.py:38-277, src/bundle_adjuster/bundle_adjuster.py:127-215).  The plain drop-in classes gather every field of every
object into arrays on the way in and scatter the results back on the way out: ~30 ms per 1241 x 376 frame, ~1 ms of it on the GPU.  Here the
state lives in the tables of csrc/vo_pipeline.hip and the caller holds

  * `LazyKeypoint` / `LazyLandmark`: one PROXY OBJECT per table row (same object for the same row, so `is` / sharing behave as in the
    reference); a field is read from the device only when somebody asks for it (bulk gather, cached until the next stage);
  * `LazyList`: a real `list` of proxies (C-speed `len`, indexing, `+`, `+=`, iteration); `copy.deepcopy` of the lists
    `extend_landmarks` returns hands out the copies the device made (pipeline.py:101-102);
  * `InlierList`: the list `camera_pose` returns, with O(1) `in` (pipeline.py:130 asks `i in inliers` for every landmark).

Each method of the reference's call sequence maps to one stage of `vo_pipe_step`:

    extend_tracks        TRACK | TRACK_CANDIDATES   pyramid + KLT of every keypoint, the candidates' keep rule / bookkeeping
    extend_landmarks     TRACK_LANDMARKS            the landmarks' half on the same tracked set; deaths -> dead lists
    camera_pose('3D-2D') POSE                       P3P-RANSAC + refinement, pruning, trajectory.append on the device
    triangulate_tracks   TRIANGULATE                DLT per birth group, filters, gate, promotion
    BundleAdjuster.adjust ADJUST                    resurrection, observation table, LM solve, write-back
    extract('shi-tomasi') DETECT                    exclusion discs + Shi-Tomasi, corners appended as candidates

A call takes the fast path only if its arguments ARE the session's current lists (checked by identity, C-speed list comparison) and its
parameters are the ones the session was created with; anything else -- another call order, foreign objects in a list, an attribute written
from outside, a list that no longer fits the tables -- DESYNCHRONISES the session: every proxy is filled with its values and becomes a
plain object, and the call (and all later ones) run the plain drop-in path, until the next `BundleAdjuster.adjust` at the end of a frame
that came through the reference's call order seeds a new session from the caller's objects.  So results are those of the plain classes
whatever the caller does; only the speed differs.

`DeviceBackend` is the only part that touches the GPU (vo_mi355x.resident.ResidentPipeline + the read-backs vo_pipe_lists_read /
vo_pipe_rows_read / vo_pipe_inliers_read); tests drive the same `Session` over the table model of oracle/pipe_oracle.py.
"""
import copy
import os
import sys
import weakref
from itertools import compress as _compress

import numpy as np

from .state import Keypoint, Landmark

HIST = 32
_K_FIELDS = ("t_first", "t_total", "uv_first", "uv", "des", "uv_history")
_L_FIELDS = ("t_latest", "p", "des")
_EXTRACTORS = weakref.WeakSet()           # drop-in Extractors alive in this process (BundleAdjuster.adjust looks for its partner here)


def enabled():
    return os.environ.get("VO_LAZY", "1") != "0"


# ------------------------------------------------------------------------------------------------------------------------------------
# proxies
# ------------------------------------------------------------------------------------------------------------------------------------
class _Proxy:
    """attached: `_sess` / `_row` set and no field in the instance dict -> __getattr__ fetches; detached (`_row` = -1): a plain object"""

    def __getattr__(self, name):
        # only reached when normal lookup fails: a field of an ATTACHED proxy (read from the tables), or of a detached one that still holds the
        # raw table row it left with (`_rec`: its fields are built the first time somebody asks)
        if name in self._FIELDS:
            d = self.__dict__
            row = d.get("_row", -1)
            if row >= 0:
                return d["_sess"]._field(self, name)
            rec = d.get("_rec")
            if rec is not None:
                v = _field_of(self._KIND, rec, name, d.get("_older"))
                d[name] = v
                return v
            if row == -2:
                raise RuntimeError("stale %s proxy: the object left the pipeline's lists and nobody held it then" % type(self).__name__)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        d = self.__dict__
        if name in self._FIELDS and d.get("_row", -1) >= 0:
            d["_sess"].desync("an attribute of a %s was written from outside" % type(self).__name__)
        d[name] = value

    __eq__ = object.__eq__
    __hash__ = object.__hash__

    def __deepcopy__(self, memo):
        d = self.__dict__
        q = d.get("_copy")
        if q:                                      # the copy the device made when this entry was pruned (pipeline.py:133-134); one per list entry
            hint = q.pop(0)                        # that held this object, handed out in list order like the reference's per-entry deepcopy calls
            if not q:
                del d["_copy"]
            return hint
        new = type(self).__new__(type(self))
        memo[id(self)] = new
        nd = new.__dict__
        nd["_row"] = -1
        if d.get("_row", -1) >= 0:                 # attached: the copy is an independent plain object with the values of now
            nd["_rec"] = d["_sess"]._gather(self._KIND, [d["_row"]])[0].copy()
            if self._KIND == "K" and int(nd["_rec"]["hist_len"]) > HIST:
                nd["_older"] = _LazyOlder(d["_sess"]._arch, d["_row"])
        else:
            if "_rec" in d:
                nd["_rec"] = d["_rec"].copy()
                if d.get("_older") is not None:
                    nd["_older"] = d["_older"]
            for f in self._FIELDS:
                if f in d:
                    nd[f] = copy.deepcopy(d[f], memo)
        return new

    def __repr__(self):
        return "<%s row %d>" % (type(self).__name__, self.__dict__.get("_row", -1))


def _field_of(kind, rec, name, older=None):
    """one field of a packed table row (resident.ResidentPipeline.K_ROW / L_ROW) as the reference's objects hold it; older: the history entries
    that have left the row's ring ([hist_len - 32, 2] float32 from the session's archive, NaN where it has none)"""
    if kind == "L":
        if name == "t_latest":
            return int(rec["t_latest"])
        if name == "p":
            return np.array(rec["p"], np.float64).reshape(3, 1)
        return np.zeros((1, 1))
    if name == "t_first":
        return int(rec["t_first"])
    if name == "t_total":
        return int(rec["t_total"])
    if name == "uv_first":
        return np.array(rec["uv_first"], np.float32).reshape(2, 1)
    if name == "uv":
        return np.array(rec["uv"], np.float32).reshape(2, 1)
    if name == "uv_history":
        n = int(rec["hist_len"])
        ring = np.array(rec["hist"], np.float32)
        lo = max(0, n - HIST)
        # entries older than the device's ring (32; a window reaches 20) come from the host-side archive of the session (`_HistArchive`); without
        # one (a row read outside a session) NaN stands in
        if isinstance(older, _LazyOlder):
            older = older.resolve(rec)
        if older is not None and len(older) == lo:
            head = [older[i].reshape(2, 1).copy() for i in range(lo)]
        else:
            head = [np.full((2, 1), np.nan, np.float32) for _ in range(lo)]
        return head + [ring[i % HIST].reshape(2, 1).copy() for i in range(lo, n)]
    return np.zeros((1, 1))


class LazyKeypoint(_Proxy, Keypoint):
    _FIELDS = frozenset(_K_FIELDS)
    _KIND = "K"


class LazyLandmark(_Proxy, Landmark):
    _FIELDS = frozenset(_L_FIELDS)
    _KIND = "L"


class LazyList(list):
    """a list of proxies; `_copies`: what deepcopy hands out (the device's copies of entries that died)"""
    _copies = None

    def __deepcopy__(self, memo):
        if self._copies is not None:
            out = LazyList(self._copies)
            for a, b in zip(self, self._copies):
                memo[id(a)] = b
            return out
        return LazyList(copy.deepcopy(x, memo) for x in self)


class InlierList(list):
    """`inliers.reshape((-1,)).tolist()` of extractor.py:191 with a set behind `in`"""

    def __init__(self, it=()):
        super().__init__(it)
        self._set = frozenset(self)

    def __contains__(self, i):
        return i in self._set


# ------------------------------------------------------------------------------------------------------------------------------------
# the GPU side
# ------------------------------------------------------------------------------------------------------------------------------------
class DeviceBackend:
    def __init__(self, ctx, K, prm):
        from .resident import ResidentPipeline
        self.ctx = ctx
        self.rp = ResidentPipeline(ctx, K, ba_window=prm["ba_window"], min_track_length=prm["min_track_length"], mask_radius=prm["mask_radius"],
                                   max_new=prm["max_new"], max_reproj_err=prm["max_reproj_err"], min_bearing_angle=prm["min_bearing_angle"],
                                   ba_max_iters=prm["ba_max_iters"], ba_ftol=prm["ba_ftol"], ba_xtol=prm["ba_xtol"], min_kp_dist=prm["min_kp_dist"],
                                   pnp_blind_batches=prm.get("pnp_blind_batches", 4))
        self.N = self.rp.N

    def seed(self, state, dead, dead_kp, t_step):
        self.rp.seed(state, dead, dead_kp, t_step)

    def push_frame(self, img):
        self.ctx.push_frame(img)

    def stage(self, stages):
        from .resident import DETECT, KEEP_FREE_LISTS
        from .resident import POSE
        self.rp.step(-1, stages | (0 if stages & DETECT else KEEP_FREE_LISTS))     # rows are recycled once per frame, behind the DETECT stage
        r = self.rp.fetch_raw()[0]
        rec = dict(status=r.status, overflow=r.overflow, t=r.t, n_new=r.n_new, n_resurrected=r.n_resurrected, n_detected=r.n_detected,
                   ba_cost0=r.ba_cost0, ba_cost=r.ba_cost, ba_iters=r.ba_iters, ba_accepted=r.ba_accepted, ba_status=r.ba_status,
                   ba_observations=r.ba_observations)
        if stages & POSE:
            H = np.eye(4)
            H[:3] = np.frombuffer(r.H, np.float64).reshape(3, 4)
            rec["H"] = H
        return rec

    def lists(self):
        T = self.rp.read_lists()
        c = T["counts"][0]
        n_c, n_l, n_d = int(c[0]), int(c[1]), int(c[2])
        return dict(cand=T["cand"][0, :n_c].copy(), lm_l=T["lm_l"][0, :n_l].copy(), lm_k=T["lm_k"][0, :n_l].copy(), dead_l=T["dead_l"][0, :n_d].copy(),
                    dead_k=T["dead_k"][0, :n_d].copy(), poses=T["poses"][0].copy(), t=int(c[5]), status=int(c[4]))

    def mask(self, n):
        return self.rp.read_inliers(n)

    def rows(self, kind, rows):
        return self.rp.read_rows(kind, rows)

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------------------------------------------
# the session
# ------------------------------------------------------------------------------------------------------------------------------------
ARCH_EVERY = 16          # frames between two archive passes (<= half the ring: consecutive passes overlap)
ARCH_BYTES = 256 << 20   # host bytes the archive may hold (4 000 tracked rows: ~0.5 MB per pass -> ~8 000 frames); older entries come back as NaN


class _HistArchive:
    """`Keypoint.uv_history` is unbounded in the reference (state/keypoint.py:4-21: one entry per frame since detection, candidate phase included);
    the device keeps the last 32 entries per row.  Every ARCH_EVERY frames the session copies the newest ARCH_EVERY entries of every keypoint row
    that has a proxy to the host (one bulk gather), so that a caller who reads an old history gets all of it -- the window of the bundle
    adjustment never reaches that far, nothing on the device needs it.
    Rows are COPIED on the device where the reference deep-copies a keypoint (deaths, pruned non-inliers, survivors that shared a row): the copy
    starts life in a new row with the history of its source.  What such a row has no record of itself is taken from the rows that share its
    identity (birth frame and first position): of those, the one whose entries agree longest with the row's own oldest known entries is its
    source (two lines of descent of one keypoint -- a resurrected dead copy tracked beside the original -- differ from where they parted).
    Host memory is bounded by a BYTE budget (`ARCH_BYTES`: the oldest passes go first, their entries come back as NaN); a lookup touches only the
    passes that hold the row (`by_row`) or its identity (`by_id`), and nothing is looked up before somebody reads a `uv_history` (`_LazyOlder`)."""

    def __init__(self, budget_bytes=None):
        self.blocks = {}          # pass id -> (rows [m] sorted, t_first [m], uv_first [m, 2], n [m], data [m, ARCH_EVERY, 2]): entries n - ARCH_EVERY .. n - 1 then
        self.by_row = {}          # row -> [(pass id, index in the pass)], oldest first
        self.by_id = {}           # (t_first, u_first, v_first) -> set of rows that held a keypoint of that identity
        self.seeded = {}          # row -> (t_first, uv_first (2,), [n, 2]): the whole history of an object the session was seeded with
        self.next_id, self.first_id, self.bytes = 0, 0, 0
        self.budget = ARCH_BYTES if budget_bytes is None else budget_bytes

    @staticmethod
    def _key(t_first, uv_first):
        return (int(t_first), float(uv_first[0]), float(uv_first[1]))

    def seed(self, row, t_first, uv_first, hist):
        self.seeded[row] = (t_first, uv_first, hist)
        self.by_id.setdefault(self._key(t_first, uv_first), set()).add(row)

    def add(self, rows, recs):
        if not len(rows):
            return
        n = recs["hist_len"].astype(np.int64)
        idx = n[:, None] - ARCH_EVERY + np.arange(ARCH_EVERY)[None, :]
        data = np.array(recs["hist"], np.float32)[np.arange(len(rows))[:, None], idx % HIST]
        data[idx < 0] = np.nan
        rows = np.asarray(rows, np.int64)
        tf, uvf = recs["t_first"].astype(np.int64), np.array(recs["uv_first"], np.float32).reshape(-1, 2)
        bid = self.next_id
        self.next_id += 1
        self.blocks[bid] = (rows, tf, uvf, n, data)
        self.bytes += data.nbytes + rows.nbytes + tf.nbytes + uvf.nbytes + n.nbytes
        for i, r in enumerate(rows.tolist()):
            self.by_row.setdefault(r, []).append((bid, i))
            self.by_id.setdefault((int(tf[i]), float(uvf[i, 0]), float(uvf[i, 1])), set()).add(r)
        while self.bytes > self.budget and len(self.blocks) > 1:          # the oldest passes go first (index entries of a gone pass are skipped and pruned on sight)
            old = self.blocks.pop(self.first_id)
            self.bytes -= old[4].nbytes + old[0].nbytes + old[1].nbytes + old[2].nbytes + old[3].nbytes
            self.first_id += 1

    def _own(self, row, t_first, uv_first, n_hi):
        """what the archive holds of the keypoint (born at t_first at uv_first: rows are recycled, and a copy keeps its source's birth frame) while
        it lived in `row`: entries 0 .. n_hi - 1, NaN where it has none"""
        out = np.full((n_hi, 2), np.nan, np.float32)
        sd = self.seeded.get(row)
        if sd is not None and sd[0] == t_first and sd[1][0] == uv_first[0] and sd[1][1] == uv_first[1]:
            m = min(len(sd[2]), n_hi)
            out[:m] = sd[2][:m]
        ent = self.by_row.get(row)
        if ent:
            if ent[0][0] < self.first_id:
                ent[:] = [e for e in ent if e[0] >= self.first_id]
            for bid, i in ent:
                rows, tf, uvf, n, data = self.blocks[bid]
                if tf[i] == t_first and uvf[i, 0] == uv_first[0] and uvf[i, 1] == uv_first[1]:
                    lo = int(n[i]) - ARCH_EVERY
                    a, b = max(lo, 0), min(int(n[i]), n_hi)
                    if b > a:
                        out[a:b] = data[i, a - lo:b - lo]
        return out

    def _relatives(self, row, t_first, uv_first):
        """other rows that hold (or held) a keypoint with this birth frame and first position"""
        return [r for r in self.by_id.get(self._key(t_first, uv_first), ()) if r != row]

    def older(self, row, t_first, uv_first, n, ring):
        """entries 0 .. n - 33 of the keypoint in `row` (history length n > 32, `ring` = its 32-entry ring)"""
        n_old = n - HIST
        own = self._own(row, t_first, uv_first, n)
        for i in range(n_old, n):
            own[i] = ring[i % HIST]
        miss = np.isnan(own[:n_old, 0])
        if miss.any():
            first = int(np.argmax(~np.isnan(own[:, 0])))         # the oldest entry the row knows of itself
            best, best_len = None, 0
            for r in self._relatives(row, t_first, uv_first):
                cand = self._own(r, t_first, uv_first, n)
                k = first
                while k < n and not np.isnan(cand[k, 0]) and cand[k, 0] == own[k, 0] and cand[k, 1] == own[k, 1]:
                    k += 1
                if k - first > best_len:
                    best, best_len = cand, k - first
            if best is not None:
                fill = miss & ~np.isnan(best[:n_old, 0])
                own[:n_old][fill] = best[:n_old][fill]
        return own[:n_old].copy()


class _LazyOlder:
    """the part of a detached keypoint's history that had left the device's ring when it was detached: looked up in the session's archive the
    first time its `uv_history` is read (most detached keypoints are never asked), from the table row the proxy took with it"""
    __slots__ = ("arch", "row")

    def __init__(self, arch, row):
        self.arch, self.row = arch, int(row)

    def resolve(self, rec):
        n = int(rec["hist_len"])
        if n <= HIST:
            return None
        return self.arch.older(self.row, int(rec["t_first"]), np.array(rec["uv_first"], np.float32).reshape(2), n, np.array(rec["hist"], np.float32))


def _idle_refcount():
    """sys.getrefcount of an object that has just left the only container that held it and is bound to ONE local name -- measured on this
    interpreter instead of assumed (CPython 3.10: the local name + getrefcount's argument = 2; an interpreter that borrows references on
    its evaluation stack reports less, and `_retire` would then take a proxy somebody still holds for an idle one)"""
    table = {0: object()}
    p = table.pop(0)
    return sys.getrefcount(p)


_REFCOUNT_IDLE = _idle_refcount()


class Session:
    def __init__(self, backend, K, prm, width, height):
        self.be, self.K, self.prm, self.w, self.h = backend, np.asarray(K, np.float64), dict(prm), width, height
        self.alive = True
        self.reason = None
        self.Kp, self.Lp = {}, {}                         # row -> proxy
        R = 4 * backend.N
        self._hasK, self._hasL = np.zeros(R, bool), np.zeros(R, bool)
        self.cand, self.lm_L, self.lm_K, self.dead_L, self.dead_K = [], [], [], [], []      # lists of proxies = the device lists
        self.rows = {}                                    # the same as row arrays
        self.t = 0
        self.last_H = None
        self.cur_img = None                               # what the device frame store holds as its current frame
        self.track_img = None                             # frame whose KLT has run and whose landmark half is still to come
        self._cache = {}                                  # kind -> (row -> index, structured array) of the last bulk gather
        self.stats = dict(fast=0, stages=0, gathers=0)
        self._hinted = []         # proxies carrying a `_copy` hint of the last camera_pose (see _drop_hints)
        self._arch = _HistArchive()
        self._arch_frames = 0     # frames since the last archive pass

    # ---- proxies / mirrors -----------------------------------------------------------------------------------------------------------
    def _proxies(self, kind, rows):
        table, cls = (self.Kp, LazyKeypoint) if kind == "K" else (self.Lp, LazyLandmark)
        has = self._hasK if kind == "K" else self._hasL
        new = rows[~has[rows]] if len(rows) else rows
        for r in np.unique(new).tolist():
            p = cls.__new__(cls)
            d = p.__dict__
            d["_sess"], d["_row"] = self, r
            table[r] = p
        if len(new):
            has[new] = True
        get = table.__getitem__
        return list(map(get, rows.tolist()))

    def _mirror(self, kind, old_list, old_rows, new_rows, mask=None):
        """the list of proxies for `new_rows`, made from the previous mirror where the stage only compacted (mask) / appended / replaced a few
        rows: C-speed list operations instead of one dict look-up per entry"""
        if old_rows is None:
            return self._proxies(kind, new_rows)
        if mask is not None:
            base, base_rows = list(_compress(old_list, mask.tolist())), old_rows[mask]
        else:
            base, base_rows = old_list, old_rows
        n = len(base)
        if len(new_rows) < n:
            return self._proxies(kind, new_rows)
        diff = np.nonzero(new_rows[:n] != base_rows)[0]
        if len(diff) > 64 + n // 8:
            return self._proxies(kind, new_rows)
        if len(diff):
            base = list(base)
            for i, p in zip(diff.tolist(), self._proxies(kind, new_rows[diff])):
                base[i] = p
        return base + self._proxies(kind, new_rows[n:]) if len(new_rows) > n else list(base)

    def _refresh(self, what, retire=False, lm_mask=None):
        """lists after a stage -> mirrors (`what`: which lists the stage may have changed).  retire: the stage was the frame's last one -- the
        device recycles the rows no list refers to (VO_PIPE_KEEP_FREE_LISTS on every other stage), so their proxies are settled now"""
        L = self.be.lists()
        old = self.rows
        self.rows = L
        self.t = L["t"]
        self._cache = {}
        if "cand" in what:
            self.cand = self._mirror("K", self.cand, old.get("cand"), L["cand"])
        if "lm" in what:
            self.lm_L = self._mirror("L", self.lm_L, old.get("lm_l"), L["lm_l"], lm_mask)
            self.lm_K = self._mirror("K", self.lm_K, old.get("lm_k"), L["lm_k"], lm_mask)
        if "dead" in what:
            self.dead_L = self._mirror("L", self.dead_L, old.get("dead_l"), L["dead_l"])
            self.dead_K = self._mirror("K", self.dead_K, old.get("dead_k"), L["dead_k"])
        if retire:
            R = len(self._hasK)
            liveK, liveL = np.zeros(R, bool), np.zeros(R, bool)
            liveK[L["cand"]] = True; liveK[L["lm_k"]] = True; liveK[L["dead_k"]] = True
            liveL[L["lm_l"]] = True; liveL[L["dead_l"]] = True
            self._retire("K", np.nonzero(self._hasK & ~liveK)[0])
            self._retire("L", np.nonzero(self._hasL & ~liveL)[0])
        return L

    def _retire(self, kind, rows):
        """proxies whose row has left every list: somebody outside still holds one -> it gets its values now (the row is still intact) and
        lives on as a plain object, like the reference's; otherwise it is simply forgotten"""
        if not len(rows):
            return
        table, has = (self.Kp, self._hasK) if kind == "K" else (self.Lp, self._hasL)
        held = []
        for r in rows.tolist():
            p = table.pop(r)
            if sys.getrefcount(p) > _REFCOUNT_IDLE:
                held.append(p)
            else:
                p.__dict__["_row"] = -2
        has[rows] = False
        if held:
            self._fill(held, detach=True)

    # ---- values ----------------------------------------------------------------------------------------------------------------------
    def _gather(self, kind, rows):
        self.stats["gathers"] += 1
        return self.be.rows(kind, np.asarray(rows, np.int32))

    def _older(self, rec, row):
        """the part of a keypoint row's history that has left its ring, from the archive (None: the ring holds everything)"""
        n = int(rec["hist_len"])
        if n <= HIST:
            return None
        return self._arch.older(int(row), int(rec["t_first"]), np.array(rec["uv_first"], np.float32).reshape(2), n, np.array(rec["hist"], np.float32))

    def _archive_pass(self):
        rows = np.nonzero(self._hasK)[0]
        if len(rows):
            self._arch.add(rows, self._gather("K", rows))
        self._arch_frames = 0

    def _fill(self, proxies, detach=True):
        """the proxies take the table rows they stand for with them (`_rec`) and become plain objects; their fields are built on first use"""
        for kind in ("K", "L"):
            ps = [p for p in proxies if p._KIND == kind and p.__dict__.get("_row", -1) >= 0]
            if not ps:
                continue
            recs = self._gather(kind, [p.__dict__["_row"] for p in ps])
            for p, rec in zip(ps, recs):
                d = p.__dict__
                d["_rec"] = rec.copy()
                if kind == "K" and int(rec["hist_len"]) > HIST:
                    d["_older"] = _LazyOlder(self._arch, d["_row"])
                d["_row"] = -1
                d.pop("_sess", None)
                d.pop("_copy", None)               # a plain object copies itself

    def _field(self, proxy, name):
        """one field of an attached proxy: the first miss after a stage gathers every row of that kind that has a proxy"""
        kind = proxy._KIND
        c = self._cache.get(kind)
        if c is None:
            rows = np.nonzero(self._hasK if kind == "K" else self._hasL)[0]
            recs = self._gather(kind, rows)
            c = self._cache[kind] = ({int(r): i for i, r in enumerate(rows.tolist())}, recs, {})
        index, recs, vals = c
        key = (proxy.__dict__["_row"], name)
        v = vals.get(key)
        if v is None:
            rec = recs[index[key[0]]]
            v = vals[key] = _field_of(kind, rec, name, self._older(rec, key[0]) if (kind == "K" and name == "uv_history") else None)
        return v

    def _drop_hints(self):
        """the deepcopy hints camera_pose attached (the copies the device made of the pruned entries) belong to the caller's deepcopy calls that
        follow it directly (pipeline.py:130-134).  Once the next stage starts -- or the session ends -- a leftover hint would hand a stale dead-row
        proxy to an unrelated copy.deepcopy of an object that is plain by then: dropped."""
        for p in self._hinted:
            p.__dict__.pop("_copy", None)
        self._hinted = []

    # ---- leaving the fast path -------------------------------------------------------------------------------------------------------
    def desync(self, reason):
        """every proxy becomes a plain object holding its current values; the session is over"""
        if not self.alive:
            return
        self._drop_hints()
        self.alive, self.reason = False, reason
        self._fill(list(self.Kp.values()) + list(self.Lp.values()), detach=True)
        self.Kp, self.Lp = {}, {}
        self.be.close()

    def _fail(self, reason):
        self.desync(reason)
        return NotImplemented

    def _check_record(self, rec, in_place=False):
        """in_place: the stage has rewritten rows the caller's objects stand for (tracked positions, adjusted landmarks), so the plain path
        cannot take over from here -- its capacity was checked BEFORE the stage ran (`_room`), a failure now is a defect"""
        if rec["status"] & 1:
            self.desync("the pose stage found no consensus")
            raise RuntimeError("solvePnPRansac found no pose")          # what the plain class raises (the reference crashes in cv2.Rodrigues(None))
        if rec["status"] or rec["overflow"]:
            if in_place:
                self.desync("capacity after an in-place stage")
                raise RuntimeError("vo_mi355x.lazy: the device tables overflowed in a stage whose capacity had been checked (status %d, overflow %d)"
                                   % (rec["status"], rec["overflow"]))
            return False
        return True

    def _room(self, extra_dead=0, extra_lm=0):
        """capacity the next stage may need: the dead list takes what can die, the state's lists what can be resurrected"""
        N = self.be.N
        return len(self.dead_L) + extra_dead <= N and len(self.lm_L) + len(self.cand) + extra_lm <= N

    @staticmethod
    def _same_image(a, b):
        return a is b or (a is not None and b is not None and a.shape == b.shape and np.array_equal(a, b))

    # ---- seeding ---------------------------------------------------------------------------------------------------------------------
    def seed(self, state, dead, dead_kp, t_step, cur_img):
        """objects -> tables; the caller's lists are rewritten IN PLACE with the proxies; -> the two dead lists as lists of proxies"""
        self.be.seed(state, dead, dead_kp, t_step)
        self.cur_img = cur_img
        L = self._refresh(("cand", "lm", "dead"), retire=True)
        if not (len(L["lm_l"]) == len(state._landmarks) and len(L["cand"]) == len(state._candidates_kp) and len(L["dead_l"]) == len(dead)):
            raise RuntimeError("seed: the tables do not hold the lists they were seeded with")
        for objs, rows in ((state._landmarks_kp, L["lm_k"]), (state._candidates_kp, L["cand"]), (dead_kp, L["dead_k"])):
            for k, r in zip(objs, rows.tolist()):
                # (every history, however short: an archive pass keeps the newest ARCH_EVERY entries of a row, the passes are ARCH_EVERY frames
                #  apart, so everything older than the first pass's reach has to be on the host already)
                self._arch.seed(r, int(k.t_first), np.asarray(k.uv_first, np.float32).reshape(2),
                                np.array([np.asarray(h, np.float32).reshape(2) for h in k.uv_history], np.float32).reshape(-1, 2))
        state._landmarks[:] = self.lm_L
        state._landmarks_kp[:] = self.lm_K
        state._candidates_kp[:] = self.cand
        T = len(state._trajectory)
        self.last_H = state._trajectory[T - 1] if T else None
        return list(self.dead_L), list(self.dead_K)

    # ---- the reference's calls -------------------------------------------------------------------------------------------------------
    def extend_tracks(self, im_prev, im_curr, kp, max_bidir_error):
        self._drop_hints()
        if not (np.isinf(max_bidir_error) and kp == self.cand and self._same_image(im_prev, self.cur_img)
                and im_curr.shape == (self.h, self.w) and self.track_img is None):
            return self._fail("extend_tracks: not the session's candidate list / frame pair")
        from .resident import TRACK, TRACK_CANDIDATES
        img = np.ascontiguousarray(im_curr, np.uint8)
        self.be.push_frame(img)
        self.cur_img = self.track_img = img.copy()
        rec = self.be.stage(TRACK | TRACK_CANDIDATES)
        self._check_record(rec, in_place=True)
        self._refresh(("cand",))
        self.stats["fast"] += 1
        return LazyList(self.cand)

    def extend_landmarks(self, im_prev, im_curr, landmarks, landmarks_kp, max_bidir_error):
        self._drop_hints()
        from .resident import TRACK, TRACK_LANDMARKS
        if not (np.isinf(max_bidir_error) and landmarks == self.lm_L and landmarks_kp == self.lm_K and self._room(extra_dead=len(self.lm_L))):
            return self._fail("extend_landmarks: not the session's landmark lists (or the dead list could overflow)")
        if self.track_img is not None:
            if not self._same_image(im_curr, self.track_img):
                return self._fail("extend_landmarks: another frame than extend_tracks")
            stages = TRACK_LANDMARKS
        else:                                              # the caller tracks its landmarks without having tracked candidates this frame
            if not (self._same_image(im_prev, self.cur_img) and im_curr.shape == (self.h, self.w)):
                return self._fail("extend_landmarks: not the session's frame pair")
            img = np.ascontiguousarray(im_curr, np.uint8)
            self.be.push_frame(img)
            self.cur_img = img.copy()
            stages = TRACK | TRACK_LANDMARKS
        self.track_img = None
        old_L, old_K, n_dead0 = self.lm_L, self.lm_K, len(self.dead_L)
        rec = self.be.stage(stages)
        self._check_record(rec, in_place=True)
        keep = self.be.mask(len(old_L))
        self._refresh(("lm", "dead"), lm_mask=keep)
        died = np.nonzero(~keep)[0].tolist()
        if len(self.lm_L) != int(keep.sum()) or len(self.dead_L) != n_dead0 + len(died):
            self.desync("extend_landmarks: the device lists do not match the keep mask")
            raise RuntimeError("vo_mi355x.lazy: extend_landmarks: the device lists do not match the keep mask")
        ld, lkd = LazyList(old_L[i] for i in died), LazyList(old_K[i] for i in died)
        ld._copies, lkd._copies = self.dead_L[n_dead0:], self.dead_K[n_dead0:]
        self.stats["fast"] += 1
        return LazyList(self.lm_L), LazyList(self.lm_K), ld, lkd

    def camera_pose(self, K, list_1, list_2, max_err_reproj):
        self._drop_hints()
        from .resident import POSE
        if not (list_1 == self.lm_L and list_2 == self.lm_K and max_err_reproj == self.prm["max_reproj_err"] and np.array_equal(np.asarray(K, np.float64), self.K)
                and self.track_img is None and self._room(extra_dead=len(self.lm_L))):
            return self._fail("camera_pose: not the session's landmark lists / parameters")
        old_L, old_K, n_dead0 = self.lm_L, self.lm_K, len(self.dead_L)
        rec = self.be.stage(POSE)
        if not self._check_record(rec):
            return self._fail("camera_pose: capacity")
        mask = self.be.mask(len(old_L))
        self._refresh(("lm", "dead"), lm_mask=mask)
        out = np.nonzero(~mask)[0].tolist()
        if len(self.lm_L) != int(mask.sum()) or len(self.dead_L) != n_dead0 + len(out):
            return self._fail("camera_pose: the device lists do not match the consensus mask")
        for j, i in enumerate(out):                        # deepcopy(state._landmarks[i]) / (..._kp[i]) of pipeline.py:133-134 = the device's copies
            old_L[i].__dict__.setdefault("_copy", []).append(self.dead_L[n_dead0 + j])
            old_K[i].__dict__.setdefault("_copy", []).append(self.dead_K[n_dead0 + j])
            self._hinted += [old_L[i], old_K[i]]
        self.last_H = np.array(rec["H"], np.float64)
        self.stats["fast"] += 1
        return InlierList(np.nonzero(mask)[0].tolist()), self.last_H

    def triangulate_tracks(self, K, candidates_kp, trajectory, t_curr, min_track_length, min_bearing_angle, max_err_reproj):
        self._drop_hints()
        from .resident import TRIANGULATE
        p = self.prm
        T = len(trajectory)
        if not (candidates_kp == self.cand and t_curr == self.t and T == t_curr + 1 and trajectory[T - 1] is self.last_H
                and (min_track_length, min_bearing_angle, max_err_reproj) == (p["min_track_length"], p["min_bearing_angle"], p["max_reproj_err"])
                and np.array_equal(np.asarray(K, np.float64), self.K) and self.track_img is None):
            return self._fail("triangulate_tracks: not the session's candidate list / trajectory / parameters")
        n_l0 = len(self.lm_L)
        rec = self.be.stage(TRIANGULATE)
        if not self._check_record(rec):
            return self._fail("triangulate_tracks: capacity")
        self._refresh(("cand", "lm"))
        if len(self.lm_L) != n_l0 + rec["n_new"]:
            return self._fail("triangulate_tracks: list length")
        self.stats["fast"] += 1
        return LazyList(self.lm_L[n_l0:]), LazyList(self.lm_K[n_l0:]), LazyList(self.cand)

    def adjust(self, state, landmarks_dead, landmarks_kp_dead, K, t_now, window, ftol, xtol, max_iters):
        self._drop_hints()
        from .resident import ADJUST
        p = self.prm
        # the caller's dead list = the device's dead list + the entries the device has dropped as inert (they can never return: flagged then)
        live, foreign = [], False
        for i, k in enumerate(landmarks_kp_dead):
            d = getattr(k, "__dict__", {})
            if d.get("_row", -1) >= 0 and d.get("_sess") is self:
                live.append(i)
            elif not d.get("_inert"):
                foreign = True
        ok = (not foreign and self._room(extra_lm=len(self.dead_L)) and state._landmarks == self.lm_L and state._landmarks_kp == self.lm_K and state._candidates_kp == self.cand and t_now == self.t
              and (window, ftol, xtol, max_iters) == (p["ba_window"], p["ba_ftol"], p["ba_xtol"], p["ba_max_iters"])
              and len(landmarks_dead) == len(landmarks_kp_dead) and [landmarks_dead[i] for i in live] == self.dead_L
              and [landmarks_kp_dead[i] for i in live] == self.dead_K and np.array_equal(np.asarray(K, np.float64), self.K) and self.track_img is None)
        T = len(state._trajectory)
        ok = ok and T == t_now + 1 and state._trajectory[T - 1] is self.last_H
        if not ok:
            return self._fail("adjust: not the session's lists / trajectory / parameters")
        n_l0 = len(self.lm_L)
        old_dead_K = self.dead_K
        rec = self.be.stage(ADJUST)
        self._check_record(rec, in_place=True)
        L = self._refresh(("lm", "dead"))
        n_res = rec["n_resurrected"]
        if len(self.lm_L) != n_l0 + n_res:
            self.desync("adjust: list length")
            raise RuntimeError("vo_mi355x.lazy: adjust: the device lists do not match the record")
        stay = set(map(id, self.dead_K))
        for k in old_dead_K:                               # entries the device dropped for good (window test failed, landmark not in the state's list)
            if id(k) not in stay:
                k.__dict__["_inert"] = True
        # recently dead landmarks are appended to the state's lists as the same objects (bundle_adjuster.py:142-147) ...
        state._landmarks.extend(self.lm_L[n_l0:])
        state._landmarks_kp.extend(self.lm_K[n_l0:])
        # ... and lead the dead lists the caller gets back, the others keep their order (:203-204), the ones the device dropped included
        taken = set(map(id, self.dead_K[:n_res]))
        front = [i for i in live if id(landmarks_kp_dead[i]) in taken]
        if len(front) != n_res:
            self.desync("adjust: resurrected entries not found in the caller's dead list")
            raise RuntimeError("vo_mi355x.lazy: adjust: resurrected entries not found in the caller's dead list")
        fs = set(front)
        rest = [i for i in range(len(landmarks_dead)) if i not in fs]
        dead_l = [landmarks_dead[i] for i in front] + [landmarks_dead[i] for i in rest]
        dead_k = [landmarks_kp_dead[i] for i in front] + [landmarks_kp_dead[i] for i in rest]
        # window poses back into the trajectory (:206-213)
        poses = L["poses"]
        for s in range(window):
            t = t_now - s
            if t not in state._trajectory._poses:
                break
            H = np.eye(4)
            H[:3] = poses[t % HIST].reshape(3, 4)
            state._trajectory._poses[t] = H
        self.last_H = state._trajectory._poses[t_now]
        self.stats["fast"] += 1
        stats = dict(cost0=rec["ba_cost0"], cost=rec["ba_cost"], iters=rec["ba_iters"], accepted=rec["ba_accepted"], status=rec["ba_status"],
                     n_obs=rec["ba_observations"]) if rec["ba_observations"] > 0 else None
        return state, dead_l, dead_k, stats

    def extract(self, img, t, current_kp, mask_radius):
        self._drop_hints()
        from .resident import DETECT
        if not (t == self.t and mask_radius == self.prm["mask_radius"] and self._same_image(img, self.cur_img) and current_kp == self.lm_K + self.cand
                and self.track_img is None):
            return self._fail("extract: not the session's keypoints / frame / parameters")
        n_c0 = len(self.cand)
        rec = self.be.stage(DETECT)
        if not self._check_record(rec):
            return self._fail("extract: capacity")
        new = self.be.lists()["cand"][n_c0:]             # (before the mirrors move: a failure here leaves the caller's objects untouched)
        if len(new) != rec["n_detected"]:
            return self._fail("extract: list length")
        self._refresh(("cand",), retire=True)
        self._arch_frames += 1
        if self._arch_frames >= ARCH_EVERY:
            self._archive_pass()
        self.stats["fast"] += 1
        return LazyList(self.cand[n_c0:])


# ------------------------------------------------------------------------------------------------------------------------------------
# how BundleAdjuster.adjust finds / starts a session
# ------------------------------------------------------------------------------------------------------------------------------------
def session_of(state, landmarks_kp_dead=()):
    """the live session whose proxies the caller's lists hold, or None"""
    for lst in (state._landmarks_kp, state._candidates_kp, landmarks_kp_dead):
        if len(lst):
            s = getattr(lst[0], "__dict__", {}).get("_sess")
            if s is not None and s.alive:
                return s
    return None


def seed_after_adjust(adjuster, state, dead_l, dead_k, K, t_now):
    """Called at the end of a plain `adjust`.  If the frame came through the reference's call order on ONE drop-in Extractor that owns a device
    context holding the frame, the caller's objects move into device tables: the state's lists are rewritten in place with proxies and the
    two dead lists come back as lists of proxies (None: nothing changed)."""
    if not enabled() or adjuster._loss != 'huber':
        return None
    cands = [e for e in _EXTRACTORS if e._lazy_on and e._frame_is_reference_order() and e._dev_cur is not None]
    if len(cands) != 1:
        return None
    ext = cands[0]
    ext._trace = []
    ctx = ext._ctx
    if ctx is None or (ext._lazy_backend is None and not hasattr(ctx, "_h")):
        return None
    N = getattr(ctx, "max_pts", 0)
    T = len(state._trajectory)
    n_l, n_c, n_d = len(state._landmarks), len(state._candidates_kp), len(dead_l)
    if not (0 < N <= 8192 and getattr(ctx, "batch", 1) == 1 and n_l + n_c + n_d <= N and n_l == len(state._landmarks_kp) and T == t_now + 1
            and all(t in state._trajectory._poses for t in range(max(0, t_now - HIST + 1), t_now + 1))):
        return None
    seen = ext._seen
    if not np.isinf(seen.get("max_bidir_error", 0.0)):
        return None
    prm = dict(ba_window=adjuster._window_size, ba_ftol=adjuster._ftol, ba_xtol=adjuster._xtol, ba_max_iters=adjuster._max_iters,
               min_track_length=seen.get("min_track_length", 3), min_bearing_angle=seen.get("min_bearing_angle", 0.5),
               max_reproj_err=seen.get("pose_max_err", 2.0), mask_radius=seen.get("mask_radius", int(ext._shitomasi_params["minDistance"])),
               max_new=ext._shitomasi_params["maxCorners"], min_kp_dist=ext._shitomasi_params["minDistance"])
    if seen.get("tri_max_err", prm["max_reproj_err"]) != prm["max_reproj_err"]:
        return None                     # one reprojection threshold on the device (pipeline.py:23 uses one for both)
    h, w = ext._dev_cur.shape
    try:
        be = ext._lazy_backend(ctx, K, prm, w, h) if ext._lazy_backend is not None else DeviceBackend(ctx, K, prm)
        sess = Session(be, K, prm, w, h)
        out = sess.seed(state, dead_l, dead_k, t_now, ext._dev_cur)
    except Exception as e:              # a state the tables cannot hold (capacity, histories ...): stay on the plain path
        if os.environ.get("VO_LAZY_STRICT"):
            raise
        ext._lazy_error = repr(e)
        return None
    ext._lazy = sess
    return out
