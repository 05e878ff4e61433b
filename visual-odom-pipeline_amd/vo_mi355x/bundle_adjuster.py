"""Drop-in for the reference's `BundleAdjuster` (src/bundle_adjuster/bundle_adjuster.py).

`adjust(state, landmarks_dead, landmarks_kp_dead, K, t_now)` keeps the reference's landmark / observation
selection, x0 packing, write-back and list aliasing (:127-215) and replaces the scipy least_squares call (:189-194)
by the HIP bundle adjustment (VoContext.ba_adjust: same Huber cost, analytic Jacobians, Schur + LM)."""
import numpy as np

from .so3 import rodrigues_mat_to_vec, rodrigues_vec_to_mat


class BundleAdjuster:
    def __init__(self, xtol=1e-2, ftol=1e-4, method='trf', verbosity=2, loss='huber', window_size=3,
                 max_err_reproj=2.0, ctx=None, device=0, max_iters=50):
        self._ftol, self._xtol = ftol, xtol
        self._method, self._verbosity = method, verbosity
        self._window_size = window_size
        self._max_err_reproj = max_err_reproj
        self._loss = loss
        self._ctx, self._device, self._max_iters = ctx, device, max_iters
        self.last_stats = None

    def _context(self):
        if self._ctx is None:
            from .context import VoContext
            self._ctx = VoContext(64, 64, max_pts=64, device=self._device)   # BA needs no frame store
        return self._ctx

    def build_problem(self, state, landmarks_dead, landmarks_kp_dead, t_now):
        """The reference's selection rules (:132-176) -> dense problem (poses [W,6], points [N,3], obs [W,N,2]).
        Like the reference this APPENDS the recently-dead landmarks to state._landmarks / state._landmarks_kp (the lists the
        solve works on ARE the state's lists, :142).  -> (poses, points, obs, number of live landmarks, dead pairs that stay out)"""
        W = self._window_size
        lms, kps = state._landmarks, state._landmarks_kp          # aliases, deliberately
        n_live = len(lms)
        stay_out = []
        for l, k in zip(landmarks_dead, landmarks_kp_dead):
            born = l.t_latest - (len(k.uv_history) - 1)           # frame of the track's first history entry (:136-139)
            if t_now - born < W:                                  # the whole track lies inside the window (:144-147)
                lms.append(l)
                kps.append(k)
            else:
                stay_out.append((l, k))
        N = len(lms)
        # observation of landmark j in window slot s <-> history index (t_now - s) - t_latest + L - 1 (:56-59, :150-158):
        # the valid slots are one contiguous run read back to front -> one slice per landmark instead of W lookups
        obs = np.full((W, N, 2), np.nan)
        for j, (l, k) in enumerate(zip(lms, kps)):
            L = len(k.uv_history)
            newest = t_now - l.t_latest + (L - 1)                 # history index of slot 0
            s_lo, s_hi = max(0, newest - (L - 1)), min(W - 1, newest)
            if s_lo <= s_hi:
                run = np.array(k.uv_history[newest - s_hi:newest - s_lo + 1], dtype=np.float64).reshape(-1, 2)
                obs[s_lo:s_hi + 1, j] = run[::-1]
        points = np.array([l.p for l in lms], dtype=np.float64).reshape(N, 3) if N else np.zeros((0, 3))
        poses = np.zeros((W, 6))                                  # poses missing at the start of a sequence stay identity (:169-171)
        n_traj = len(state._trajectory)
        for s in range(min(W, n_traj)):
            H = state._trajectory[n_traj - 1 - s]
            poses[s, :3] = rodrigues_mat_to_vec(H[:3, :3])
            poses[s, 3:] = H[:3, 3]
        return poses, points, obs, n_live, [p[0] for p in stay_out], [p[1] for p in stay_out]

    def adjust(self, state, landmarks_dead, landmarks_kp_dead, K, t_now):
        from . import lazy as _lz
        sess = _lz.session_of(state, landmarks_kp_dead)
        if sess is not None:          # the lists are views of the device tables (lazy.py): resurrection, solve and write-back happen there
            r = sess.adjust(state, landmarks_dead, landmarks_kp_dead, K, t_now, self._window_size, self._ftol, self._xtol, self._max_iters) \
                if self._loss == 'huber' else sess._fail("adjust: loss")
            if r is not NotImplemented:
                state, dead_l, dead_k, self.last_stats = r
                return state, dead_l, dead_k
        state, dead_l, dead_k = self._adjust_plain(state, landmarks_dead, landmarks_kp_dead, K, t_now)
        # a frame that came through the reference's call order ends here: from now on the state can live in device tables
        seeded = _lz.seed_after_adjust(self, state, dead_l, dead_k, K, t_now)
        return (state,) + seeded if seeded is not None else (state, dead_l, dead_k)

    def _adjust_plain(self, state, landmarks_dead, landmarks_kp_dead, K, t_now):
        poses, points, obs, n_live, out_l, out_k = self.build_problem(state, landmarks_dead, landmarks_kp_dead, t_now)
        lms, kps = state._landmarks, state._landmarks_kp
        N, W = len(lms), self._window_size
        if N > 0 and not np.isnan(obs[..., 0]).all():
            c = self._context()
            delta = 1.0 if self._loss == 'huber' else 1e30       # 'linear' loss == Huber with an unreachable knee
            prm = c.ba_params(max_iters=self._max_iters, ftol=self._ftol, xtol=self._xtol, huber_delta=delta)
            poses, points, self.last_stats = c.ba_adjust(np.asarray(K, np.float64), poses, points, obs, prm)
        # positions back into the landmark objects, every landmark its own (3, 1) array (:197-201)
        fresh = np.ascontiguousarray(points, np.float64).reshape(N, 3, 1).copy()
        for l, p in zip(lms, fresh):
            l.p = p
        # the resurrected ones lead the dead lists the caller gets back (:203-204)
        dead_l, dead_k = lms[n_live:] + out_l, kps[n_live:] + out_k
        # window poses back into the trajectory, newest first, as far as it reaches back (:206-213)
        for s in range(W):
            if (t_now - s) not in state._trajectory._poses:
                break
            H = np.eye(4)
            H[:3, :3] = rodrigues_vec_to_mat(poses[s, :3])
            H[:3, 3] = poses[s, 3:]
            state._trajectory._poses[t_now - s] = H
        return state, dead_l, dead_k
