"""TEST INFRASTRUCTURE -- not part of the product.  Array model of the device-resident Pipeline.step.

The reference keeps its per-frame state in Python lists of `Keypoint` / `Landmark` OBJECTS
(/root/reference/src/pipeline/pipeline.py:92-167, src/state/*.py) and relies on object identity in three places:
`BundleAdjuster.adjust` appends recently dead landmarks to `state._landmarks` WITHOUT copying them
(src/bundle_adjuster/bundle_adjuster.py:142-147), `Extractor.extend_landmarks` deep-copies the
keypoint of a survivor but keeps the landmark object (src/extractor/extractor.py:80-86), and `Pipeline.step`
deep-copies what dies (pipeline.py:101-102, 133-134; one `deepcopy` call memoises, so two dying entries that share
a landmark object share the copy).  The device keeps the same state as TABLES OF OBJECTS with stable rows plus
ORDERED LISTS OF ROW INDICES, which represents sharing exactly:

    K objects  t_first, t_total, uv_first, uv, hist_len, hist ring (entry idx at slot idx % HIST)      Keypoint
    L objects  t_latest, p                                                                            Landmark
    cand       [K]                                                   state._candidates_kp
    lm         [(L, K, k_shared)]                                    state._landmarks / _landmarks_kp (parallel lists)
    dead       [(L, K)]                                              Pipeline._landmarks_dead / _landmarks_kp_dead
    poses      t -> H (4x4)                                          state._trajectory

Dead entries that can never be resurrected again (their landmark object is referenced by no state entry and the
window test of bundle_adjuster.py:144-150 has failed: it can only fail harder) are dropped from the list and only
counted -- the reference keeps them for its visualiser.

This file restates that algorithm with numpy arrays, phase by phase as the HIP kernels of csrc/vo_pipeline.hip do
it, over a numerical back end with the VoContext API (tests pass the CPU OracleContext or a GPU VoContext).
tests/test_pipe_model.py checks it against the reference's loop over the drop-in classes, object by object."""
import numpy as np

HIST = 32

# per-sequence status bits (same values as VO_PIPE_* in include/vo_mi355x.h)
ST_LOST = 1          # the 3D-2D pose found no consensus (the reference crashes in cv2.Rodrigues(None))
ST_CAPACITY = 2      # a list or an object table is full
ST_GROUPS = 4        # a ripe candidate's birth pose has left the 32-frame trajectory ring


def _rot_to_vec(R):
    """cv2.Rodrigues matrix -> vector (project onto SO(3), axis from the antisymmetric part, the pi branch from the
    diagonal), the log map the reference packs its window poses with (bundle_adjuster.py:173).  The oracle's OWN statement: nothing under
    oracle/ imports the product."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(0.25 * (v @ v))
    c = min(1.0, max(-1.0, (R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        rx = np.sqrt(max((R[0, 0] + 1) * 0.5, 0.0))
        ry = np.sqrt(max((R[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        rz = np.sqrt(max((R[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        w = np.array([rx, ry, rz])
        n = np.linalg.norm(w)
        return w * (theta / n) if n > 0 else np.zeros(3)
    return v * (0.5 * theta / s)


def _vec_to_rot(r):
    """cv2.Rodrigues vector -> matrix (bundle_adjuster.py:48,211)"""
    r = np.asarray(r, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    cs, sn = np.cos(theta), np.sin(theta)
    return cs * np.eye(3) + (1.0 - cs) * np.outer(k, k) + sn * Kx


def cpython_set_order(keys):
    """Iteration order of `set(keys)` for small non-negative ints in CPython 3 (Objects/setobject.c), which is the order the
    reference walks the birth-frame groups in (`for t_first in set([k.t_first ...])`, extractor.py:210-211): an open-addressing table
    of 8 slots (x4 whenever fill * 5 >= mask * 3: 32 at the 5th key, 128 at the 19th), slot = hash & mask with hash(i) = i, on a collision
    up to 9 linear probes while they stay inside the table, then i = (5 i + 1 + perturb) & mask with perturb >>= 5; a resize re-inserts
    the keys in slot order; iteration walks the slots.  The device does the same walk in k_pipe_promote (vo_pipeline.hip)."""
    def insert(table, mask, key):
        perturb, i = key, key & mask
        while True:
            if table[i] is None or table[i] == key:
                table[i] = key
                return
            if i + 9 <= mask:
                for j in range(i + 1, i + 10):
                    if table[j] is None or table[j] == key:
                        table[j] = key
                        return
            perturb >>= 5
            i = (i * 5 + 1 + perturb) & mask
    mask, table, fill = 7, [None] * 8, 0
    for key in keys:
        key = int(key)
        assert key >= 0
        if key in table:
            continue
        insert(table, mask, key)
        fill += 1
        if fill * 5 >= mask * 3:
            old = [k for k in table if k is not None]
            size = 8
            while size <= fill * 4:
                size <<= 1
            mask, table = size - 1, [None] * size
            for k in old:
                insert(table, mask, k)
    return [k for k in table if k is not None]


class Params:
    def __init__(self, ba_window=4, min_track_length=3, mask_radius=7, max_new=1000, max_reproj_err=2.0, min_bearing_angle=0.5,
                 ba_max_iters=50, ba_ftol=1e-3, ba_xtol=1e-3, pnp_seed=0, min_distance=7, resurrect=True):
        self.ba_window, self.min_track_length, self.mask_radius, self.max_new = ba_window, min_track_length, mask_radius, max_new
        self.max_reproj_err, self.min_bearing_angle = max_reproj_err, min_bearing_angle
        self.ba_max_iters, self.ba_ftol, self.ba_xtol, self.pnp_seed, self.min_distance = ba_max_iters, ba_ftol, ba_xtol, pnp_seed, min_distance
        self.resurrect = resurrect       # False: dead landmarks stay dead (not the reference; see vo_pipe_params.resurrect)


class PipeModel:
    """ONE sequence.  cap: capacity of the lists (cand + lm <= cap: they share the KLT point buffer); cap_obj: object rows."""

    def __init__(self, ctx, K, width, height, cap=4096, cap_obj=None, params=None):
        self.ctx, self.K, self.w, self.h = ctx, np.asarray(K, np.float64), width, height
        self.cap, self.cap_obj = cap, cap_obj or 4 * cap
        self.prm = params or Params()
        n = self.cap_obj
        self.k_tf, self.k_tt, self.k_len = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.k_uv, self.k_first = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
        self.k_hist = np.full((n, HIST, 2), np.nan, np.float32)
        self.l_tl, self.l_p = np.zeros(n, np.int32), np.zeros((n, 3), np.float64)
        self.freeK, self.freeL = list(range(n - 1, -1, -1)), list(range(n - 1, -1, -1))      # stacks: pop() hands out row 0 first
        self.cand = []
        self.lm_L, self.lm_K, self.lm_ksh = [], [], []
        self.dead_L, self.dead_K = [], []
        self.n_dead_inert = 0
        self.poses = {}
        self.t = 0
        self.status = 0
        self.info = {}

    # ---- object rows ---------------------------------------------------------------------------
    def _alloc(self, stack):
        if not stack:
            self.status |= ST_CAPACITY
            raise MemoryError("object table full")
        return stack.pop()

    def new_K(self, t_first, t_total, uv_first, uv, hist):
        k = self._alloc(self.freeK)
        self.k_tf[k], self.k_tt[k] = t_first, t_total
        self.k_first[k], self.k_uv[k] = np.float32(uv_first).reshape(2), np.float32(uv).reshape(2)
        hist = np.asarray(hist, np.float32).reshape(-1, 2)
        self.k_len[k] = len(hist)
        self.k_hist[k] = np.nan
        for idx in range(max(0, len(hist) - HIST), len(hist)):
            self.k_hist[k, idx % HIST] = hist[idx]
        return k

    def copy_K(self, src):
        k = self._alloc(self.freeK)
        for a in (self.k_tf, self.k_tt, self.k_len, self.k_uv, self.k_first, self.k_hist):
            a[k] = a[src]
        return k

    def new_L(self, t_latest, p):
        l = self._alloc(self.freeL)
        self.l_tl[l], self.l_p[l] = t_latest, np.asarray(p, np.float64).reshape(3)
        return l

    def copy_L(self, src):
        return self.new_L(self.l_tl[src], self.l_p[src])

    def _push_dead(self, make_L, make_K):
        """append a dead entry; a full dead list drops it (it can then never be resurrected: counted with the inert ones)"""
        if len(self.dead_L) >= self.cap:
            self.n_dead_inert += 1
            self.info["overflow"] = self.info.get("overflow", 0) | 1
            return
        self.dead_L.append(make_L()); self.dead_K.append(make_K())

    def _append_hist(self, k, uv):
        self.k_hist[k, self.k_len[k] % HIST] = uv
        self.k_len[k] += 1

    def hist_at(self, k, idx):
        """k.uv_history[idx] (valid while the ring still holds it)"""
        assert 0 <= idx < self.k_len[k] and idx >= self.k_len[k] - HIST
        return self.k_hist[k, idx % HIST]

    def sweep(self):
        """end of a step: rows no list refers to go back to the free stacks (ascending, so pop() hands out the lowest first)"""
        usedK, usedL = np.zeros(self.cap_obj, bool), np.zeros(self.cap_obj, bool)
        usedK[self.cand] = True; usedK[self.lm_K] = True; usedK[self.dead_K] = True
        usedL[self.lm_L] = True; usedL[self.dead_L] = True
        self.freeK = [int(i) for i in np.nonzero(~usedK)[0][::-1]]
        self.freeL = [int(i) for i in np.nonzero(~usedL)[0][::-1]]

    # ---- seeding from the reference's objects (Pipeline._get_init_state, pipeline.py:42-90, or any later state) ----
    def seed(self, state, landmarks_dead, landmarks_kp_dead, t_step):
        kmap, lmap = {}, {}

        def K_of(k):
            if id(k) not in kmap:
                kmap[id(k)] = self.new_K(k.t_first, k.t_total, k.uv_first, k.uv, np.array(k.uv_history, np.float64).reshape(-1, 2))
            return kmap[id(k)]

        def L_of(l):
            if id(l) not in lmap:
                lmap[id(l)] = self.new_L(l.t_latest, l.p)
            return lmap[id(l)]
        self.cand = [K_of(k) for k in state._candidates_kp]
        self.lm_L = [L_of(l) for l in state._landmarks]
        self.lm_K = [K_of(k) for k in state._landmarks_kp]
        self.dead_L = [L_of(l) for l in landmarks_dead]
        self.dead_K = [K_of(k) for k in landmarks_kp_dead]
        dead_k = set(self.dead_K)
        self.lm_ksh = [k in dead_k for k in self.lm_K]
        self.poses = {int(t): np.array(H, np.float64) for t, H in state._trajectory._poses.items()}
        self.t = t_step
        self.n_dead_inert = 0

    # ---- stage 1: KLT + the keep rule (extractor.py:38-88, pipeline.py:98-103) ------------------------------
    def dense_points(self):
        """the resident point set: landmark entries first, then candidates"""
        ids = self.lm_K + self.cand
        return self.k_uv[ids].reshape(-1, 2).copy() if ids else np.zeros((0, 2), np.float32)

    def track(self, img):
        self.extend(self.track_points(img))

    def track_points(self, img):
        """pyramid + KLT of every keypoint of the state (landmark entries first): the tracked positions both halves of `extend` read"""
        self.ctx.push_frame(img)
        p0 = self.dense_points()
        self._p1 = self.ctx.klt_track(p0)[0] if len(p0) else p0
        self._p1_nl = len(self.lm_L)
        return self._p1

    def _inside(self, p):
        return (p[:, 0] >= 0) & (p[:, 0] <= self.w) & (p[:, 1] >= 0) & (p[:, 1] <= self.h)     # NaN fails, ends included

    def extend(self, p1, which=3):
        """which: bit 0 the candidates (extend_tracks), bit 1 the landmarks (extend_landmarks; the step counter advances here) -- the two
        halves of k_pipe_extend, which the object boundary runs one after the other on the same tracked point set"""
        nl = len(self.lm_L)
        keep = self._inside(p1)
        if which & 1:
            # candidates (extend_tracks)
            out = []
            for i, k in enumerate(self.cand):
                if keep[nl + i]:
                    self.k_uv[k] = p1[nl + i]; self.k_tt[k] += 1; self._append_hist(k, p1[nl + i])
                    out.append(k)
            self.cand = out
        if not (which & 2):
            return
        self.t += 1
        self.keep_mask = keep[:nl].copy()
        # landmarks (extend_landmarks): phase A -- survivors update their keypoint and their landmark's t_latest
        for j in range(nl):
            if keep[j]:
                k = self.lm_K[j]
                self.k_uv[k] = p1[j]; self.k_tt[k] += 1; self._append_hist(k, p1[j])
                self.l_tl[self.lm_L[j]] += 1
        # phase B -- deepcopy(k) of a survivor (extractor.py:85): observable only when the keypoint object is also in the dead list
        for j in range(nl):
            if keep[j] and self.lm_ksh[j]:
                self.lm_K[j] = self.copy_K(self.lm_K[j]); self.lm_ksh[j] = False
        # phase C -- what died is deep-copied into the dead lists AFTER the loop (pipeline.py:101-102): one deepcopy call per list,
        # so entries that share a landmark object share its copy
        memo = {}

        def shared_copy(L):
            if L not in memo:
                memo[L] = self.copy_L(L)
            return memo[L]
        for j in range(nl):
            if not keep[j]:
                self._push_dead(lambda: shared_copy(self.lm_L[j]), lambda: self.copy_K(self.lm_K[j]))
        self.lm_L = [v for v, kp in zip(self.lm_L, keep[:nl]) if kp]
        self.lm_K = [v for v, kp in zip(self.lm_K, keep[:nl]) if kp]
        self.lm_ksh = [False] * len(self.lm_L)

    # ---- stage 2: 3D-2D pose + pruning (extractor.py:174-191, pipeline.py:124-140) --------------------------
    def localize(self):
        X = self.l_p[self.lm_L].astype(np.float32).reshape(-1, 3)
        uv = self.k_uv[self.lm_K].reshape(-1, 2)
        if len(X) < 4:
            self.status |= ST_LOST
            return
        rvec, tvec, inl, st = self.ctx.pnp_ransac(self.K, X, uv, reproj_err=self.prm.max_reproj_err, confidence=0.9999,
                                                  max_iters=1000000, seed=self.prm.pnp_seed)
        self.info["pnp"] = st
        if st["status"] != 0:
            self.status |= ST_LOST
            return
        mask = np.zeros(len(X), bool)
        mask[np.asarray(inl, np.int64).reshape(-1)] = True
        self.inlier_mask = mask.copy()
        for j in range(len(X)):
            if not mask[j]:       # deepcopy per entry (pipeline.py:133-134): no shared copies
                self._push_dead(lambda: self.copy_L(self.lm_L[j]), lambda: self.copy_K(self.lm_K[j]))
        self.lm_L = [v for v, m in zip(self.lm_L, mask) if m]
        self.lm_K = [v for v, m in zip(self.lm_K, mask) if m]
        self.lm_ksh = [False] * len(self.lm_L)
        H = np.eye(4)
        H[:3, :3] = _vec_to_rot(rvec); H[:3, 3] = np.asarray(tvec).reshape(3)
        self.poses[self.t] = H

    # ---- stage 3: triangulation of ripe candidates (extractor.py:193-277, triangulate.py:82-146) ------------
    def triangulate(self):
        prm = self.prm
        ripe = [k for k in self.cand if self.k_tt[k] >= prm.min_track_length]
        self.cand = [k for k in self.cand if self.k_tt[k] < prm.min_track_length]
        self.info["n_new"] = 0
        if not ripe:
            return
        H1 = self.poses[self.t]
        # groups by birth frame in the order CPython iterates the set the reference builds from the ripe list (extractor.py:210-212)
        for born in cpython_set_order(int(self.k_tf[k]) for k in ripe):
            grp = [k for k in ripe if int(self.k_tf[k]) == born]
            if born not in self.poses or not (0 <= self.t - born < HIST):
                self.status |= ST_GROUPS
                return
            H0 = self.poses[born]
            P0, P1 = np.float32(self.K @ H0[:3]), np.float32(self.K @ H1[:3])
            X4, depth1, reproj = self.ctx.triangulate(P0, P1, self.k_first[grp], self.k_uv[grp], self.K, H0, H1)
            X4 = X4.reshape(4, -1)
            pts = (X4[:3] / X4[3]).T
            rows = np.nonzero((np.asarray(depth1) > 0) & (np.asarray(reproj) < prm.max_reproj_err))[0]
            if not len(rows):
                continue
            # the reference's "bearing angle" of the group's first landmark (extractor.py:231-240)
            ray = np.zeros(4); ray[:3] = np.float64(pts[rows[0]])
            a = np.linalg.norm(H1 @ np.linalg.inv(H0))
            b, c = np.linalg.norm(H0 @ ray), np.linalg.norm(H1 @ ray)
            with np.errstate(invalid='ignore', divide='ignore'):
                theta = np.degrees(np.arccos((b * b + c * c - a * a) / (2 * b * c)))
            if not (theta > prm.min_bearing_angle):
                continue
            room = self.cap - len(self.lm_L) - len(self.cand)          # capacity policy: the lists share the KLT point buffer
            if len(rows) > room:
                rows = rows[:max(room, 0)]
                self.info["overflow"] = self.info.get("overflow", 0) | 2
            for i in rows:
                self.lm_L.append(self.new_L(self.t, np.float64(pts[i]))); self.lm_K.append(grp[i]); self.lm_ksh.append(False)
            self.info["n_new"] += len(rows)

    # ---- stage 4: sliding-window bundle adjustment (bundle_adjuster.py:127-215) -----------------------------
    def ba_problem(self):
        W, t_now = self.prm.ba_window, self.t
        n_active = len(self.lm_L)
        window = [self.prm.resurrect and (t_now - (int(self.l_tl[L]) - (int(self.k_len[K]) - 1))) < W for L, K in zip(self.dead_L, self.dead_K)]
        room = self.cap - len(self.lm_L) - len(self.cand)          # capacity policy: resurrect in dead-list order while there is room
        crit, taken = [], 0
        for ok in window:
            take = ok and taken < room
            taken += 1 if take else 0
            crit.append(take)
        if sum(window) > taken:
            self.info["overflow"] = self.info.get("overflow", 0) | 4
        for ok, L, K in zip(crit, self.dead_L, self.dead_K):
            if ok:          # appended to the state's lists as the SAME objects (:142-147)
                self.lm_L.append(L); self.lm_K.append(K); self.lm_ksh.append(True)
        live_L = set(self.lm_L)
        live_L |= {L for ok, win, L in zip(crit, window, self.dead_L) if win and not ok}      # (capacity policy: in the window, no room this time)
        front = [(L, K) for ok, L, K in zip(crit, self.dead_L, self.dead_K) if ok]
        # an entry that stays dead is kept while it may still be resurrected: the window test holds (no room this time) or its
        # Landmark object is in the state's list (its t_latest still advances); otherwise it is inert and only counted
        kept = [(L, K) for ok, win, L, K in zip(crit, window, self.dead_L, self.dead_K) if not ok and (win or L in live_L)]
        self.n_dead_inert += len(self.dead_L) - len(front) - len(kept)
        self.dead_L = [L for L, _ in front + kept]; self.dead_K = [K for _, K in front + kept]
        self.info["n_resurrected"] = len(front)
        N = len(self.lm_L)
        obs = np.full((W, N, 2), np.nan)
        for j, (L, K) in enumerate(zip(self.lm_L, self.lm_K)):
            n = int(self.k_len[K])
            for s in range(W):
                idx = (t_now - s) - int(self.l_tl[L]) + n - 1
                if 0 <= idx <= n - 1:
                    obs[s, j] = self.hist_at(K, idx)
        points = self.l_p[self.lm_L].reshape(N, 3).copy()
        poses = np.zeros((W, 6))
        for i in range(W):
            if t_now - i < 0:
                break
            H = self.poses[t_now - i]
            poses[i, :3] = _rot_to_vec(H[:3, :3]); poses[i, 3:] = H[:3, 3]
        return poses, points, obs, n_active

    def adjust(self):
        prm = self.prm
        poses, points, obs, n_active = self.ba_problem()
        N, W = len(points), prm.ba_window
        self.info["ba"] = None
        if N > 0 and not np.isnan(obs[..., 0]).all():
            bp = self.ctx.ba_params(max_iters=prm.ba_max_iters, ftol=prm.ba_ftol, xtol=prm.ba_xtol)
            poses, points, self.info["ba"] = self.ctx.ba_adjust(self.K, poses, points, obs, bp)
        for i in range(N):                 # in list order: entries sharing a landmark object -- the last one wins (:197-201)
            self.l_p[self.lm_L[i]] = points[i]
        for i in range(W):
            t = self.t - i
            if t < 0:
                break
            H = np.eye(4)
            H[:3, :3] = _vec_to_rot(poses[i, :3]); H[:3, 3] = poses[i, 3:]
            self.poses[t] = H

    # ---- stage 5: re-detection (extractor.py:90-132, pipeline.py:159-163) -----------------------------------
    def detect(self):
        prm = self.prm
        cur = self.dense_points()
        sp = self.ctx.st_params(max_corners=1000, quality_level=0.03, min_distance=prm.min_distance, block_size=31)
        corners = self.ctx.shi_tomasi(cur if len(cur) else None, mask_radius=prm.mask_radius, params=sp)
        corners = np.asarray(corners, np.float32).reshape(-1, 2)[:prm.max_new]
        room = self.cap - len(self.lm_L) - len(self.cand)
        if len(corners) > room:
            corners = corners[:max(room, 0)]
            self.info["overflow"] = self.info.get("overflow", 0) | 8
        for q in corners:
            self.cand.append(self.new_K(self.t, 1, q, q, q[None]))
        self.info["n_detected"] = len(corners)

    def step(self, img):
        if self.status:
            return
        self.info = {}
        self.track(img)
        self.localize()
        if self.status:
            return
        self.triangulate()
        if self.status:
            return
        self.adjust()
        self.detect()
        self.sweep()

    # ---- what a test compares ---------------------------------------------------------------------
    def counts(self):
        return dict(n_lm=len(self.lm_L), n_cand=len(self.cand), n_dead=len(self.dead_L) + self.n_dead_inert)

    def entry(self, L, K):
        """(t_latest, p, t_first, t_total, uv_first, uv, hist_len, last min(len, HIST) history entries oldest first)"""
        n = int(self.k_len[K])
        hist = np.array([self.hist_at(K, i) for i in range(max(0, n - HIST), n)], np.float32).reshape(-1, 2)
        return (None if L is None else int(self.l_tl[L]), None if L is None else self.l_p[L].copy(), int(self.k_tf[K]), int(self.k_tt[K]),
                self.k_first[K].copy(), self.k_uv[K].copy(), n, hist)
