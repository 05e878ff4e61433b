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
        Like the reference this APPENDS the recently-dead landmarks to state._landmarks / state._landmarks_kp."""
        W = self._window_size
        n_landmarks_active = len(state._landmarks)
        t_earliest = [l.t_latest - (len(k.uv_history) - 1) for l, k in zip(landmarks_dead, landmarks_kp_dead)]
        refine_landmarks, refine_landmarks_kp = state._landmarks, state._landmarks_kp     # aliases (:142)
        unrefined_landmarks, unrefined_landmarks_kp = [], []
        for i, l in enumerate(landmarks_dead):
            if (t_now - t_earliest[i]) < W:
                refine_landmarks.append(l)
                refine_landmarks_kp.append(landmarks_kp_dead[i])
            else:
                unrefined_landmarks.append(l)
                unrefined_landmarks_kp.append(landmarks_kp_dead[i])
        N = len(refine_landmarks)
        # observation of landmark j in window slot t <-> history index (t_now - t) - t_latest + L - 1 (:56-59, :150-158):
        # valid slots are one contiguous run, read back to front -> one slice per landmark instead of W lookups
        obs = np.full((W, N, 2), np.nan)
        for j, (l, k) in enumerate(zip(refine_landmarks, refine_landmarks_kp)):
            L = len(k.uv_history)
            base = t_now - l.t_latest + (L - 1)              # history index of slot 0
            t_lo, t_hi = max(0, base - (L - 1)), min(W - 1, base)
            if t_lo <= t_hi:
                h = np.array(k.uv_history[base - t_hi:base - t_lo + 1], dtype=np.float64).reshape(-1, 2)
                obs[t_lo:t_hi + 1, j] = h[::-1]
        points = (np.array([l.p for l in refine_landmarks], dtype=np.float64).reshape(N, 3) if N else np.zeros((0, 3)))
        poses = np.zeros((W, 6))        # poses missing at the start of a sequence stay identity (:169-171)
        T = len(state._trajectory)
        for i in range(W):
            if T - 1 - i < 0:
                break
            H = state._trajectory[T - 1 - i]
            poses[i, :3] = rodrigues_mat_to_vec(H[:3, :3])
            poses[i, 3:] = H[:3, 3]
        return poses, points, obs, n_landmarks_active, unrefined_landmarks, unrefined_landmarks_kp

    def adjust(self, state, landmarks_dead, landmarks_kp_dead, K, t_now):
        poses, points, obs, n_active, unref_l, unref_k = self.build_problem(state, landmarks_dead, landmarks_kp_dead, t_now)
        refine_landmarks, refine_landmarks_kp = state._landmarks, state._landmarks_kp
        N, W = len(refine_landmarks), self._window_size
        if N > 0 and not np.isnan(obs[..., 0]).all():
            c = self._context()
            delta = 1.0 if self._loss == 'huber' else 1e30       # 'linear' loss == Huber with an unreachable knee
            prm = c.ba_params(max_iters=self._max_iters, ftol=self._ftol, xtol=self._xtol, huber_delta=delta)
            poses_out, points_out, self.last_stats = c.ba_adjust(np.asarray(K, np.float64), poses, points, obs, prm)
        else:
            poses_out, points_out = poses, points
        # write back exactly like the reference (:197-213)
        p_new = np.ascontiguousarray(points_out, np.float64).reshape(N, 3, 1).copy()     # each landmark gets its own row
        for i in range(N):
            refine_landmarks[i].p = p_new[i]
        landmarks_dead = [refine_landmarks[i] for i in range(n_active, N)] + unref_l
        landmarks_kp_dead = [refine_landmarks_kp[i] for i in range(n_active, N)] + unref_k
        for i in range(W):
            t = t_now - i
            if not (t in state._trajectory._poses):
                break
            H_i = np.eye(4)
            H_i[:3, :3] = rodrigues_vec_to_mat(poses_out[i, :3])
            H_i[:3, 3] = poses_out[i, 3:]
            state._trajectory._poses[t] = H_i
        return state, landmarks_dead, landmarks_kp_dead
