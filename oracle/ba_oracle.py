"""Sliding-window bundle adjustment -- CPU ORACLE (numpy, float64).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

What it restates
----------------
* `residual_norm`  : the reference objective, /root/reference/src/bundle_adjuster/
  bundle_adjuster.py:18-65 (+ `_project` :68-83): one residual per observation,
  r = || dehom(K [R(rvec_i)|t_i] [X_j;1]) - uv ||_2, stacked slot-major (slot 0 =
  newest frame) and, inside a slot, by ascending landmark index (:153-158).
  PINNED against the reference itself: tests/golden/ba_*.npz were produced by
  importing the reference module in the build container (tests/golden/gen_golden.py).
* `pack_x0 / sparsity_coo`: x0 layout and Jacobian structure of `adjust`
  (:165-176) and `_jacobian_sparsity` (:85-124).  PINNED likewise.
* the robust cost: scipy's Huber, rho(z) = z (z<=1), 2 sqrt(z) - 1 (z>1), cost =
  0.5 sum rho(r^2), f_scale = 1 (scipy/optimize/_lsq/least_squares.py:169-178).

What it defines (the build's solver; `north_star` asks for an analytic-Jacobian
J^T J normal-equation solve instead of the reference's finite-difference TRF/LSMR)
* analytic 2x6 / 2x3 Jacobian blocks of the 2-vector pixel error e,
* IRLS-weighted normal equations  H = sum w J^T J, g = sum w J^T e, w = rho'(|e|^2),
* Schur complement on the landmark blocks and a Levenberg-Marquardt loop with
  Marquardt (diag H) scaling and Nielsen's gain-ratio damping update; termination
  mirrors scipy's ftol / xtol / gtol tests (scipy/optimize/_lsq/common.py:705-717).
The GPU implementation follows exactly this algorithm; iterates are compared
step by step in tests/test_gpu_ba.py.
"""
import numpy as np

HUBER_DELTA = 1.0


def skew(v):
    v = np.asarray(v, np.float64)
    out = np.zeros(v.shape[:-1] + (3, 3))
    out[..., 0, 1], out[..., 0, 2] = -v[..., 2], v[..., 1]
    out[..., 1, 0], out[..., 1, 2] = v[..., 2], -v[..., 0]
    out[..., 2, 0], out[..., 2, 1] = -v[..., 1], v[..., 0]
    return out


def rodrigues_exp(r):
    """rvec (3,) -> R (3,3); the closed form of cv2.Rodrigues."""
    r = np.asarray(r, np.float64).reshape(3)
    th = np.sqrt(r @ r)
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / th
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * skew(k)


def right_jacobian(r):
    """J_r(r) of SO(3):  R(r + dr) ~ R(r) Exp(J_r dr)."""
    r = np.asarray(r, np.float64).reshape(3)
    th2 = r @ r
    S = skew(r)
    if th2 < 1e-8:
        a, b = 0.5 - th2 / 24.0, 1.0 / 6.0 - th2 / 120.0
    else:
        th = np.sqrt(th2)
        a, b = (1 - np.cos(th)) / th2, (th - np.sin(th)) / (th2 * th)
    return np.eye(3) - a * S + b * (S @ S)


def project(K, pose, X):
    """pose = (rvec, tvec) world->camera; X (N,3) -> uv (N,2), Xc (N,3), p = K Xc (N,3)"""
    R = rodrigues_exp(pose[:3])
    Xc = X @ R.T + pose[3:]
    p = Xc @ np.asarray(K, np.float64).T
    return p[:, :2] / p[:, 2:3], Xc, p


def valid_mask(obs):
    return ~np.isnan(obs[..., 0])


def residual_vec(K, poses, points, obs):
    """e [W,N,2] (0 where not observed)"""
    W, N = obs.shape[:2]
    e = np.zeros((W, N, 2))
    m = valid_mask(obs)
    for i in range(W):
        uv, _, _ = project(K, poses[i], points)
        e[i][m[i]] = uv[m[i]] - obs[i][m[i]]
    return e


def residual_norm(K, poses, points, obs):
    """The reference's residual vector (bundle_adjuster.py:18-65): slot-major, ascending j."""
    e = residual_vec(K, poses, points, obs)
    m = valid_mask(obs)
    return np.concatenate([np.linalg.norm(e[i][m[i]], axis=1) for i in range(obs.shape[0])])


def pack_x0(poses, points):
    """x0 = [X_0 .. X_{N-1} | (rvec, tvec) slot 0 .. slot W-1]  (bundle_adjuster.py:165-176)"""
    return np.concatenate([np.asarray(points, np.float64).reshape(-1), np.asarray(poses, np.float64).reshape(-1)])


def unpack_x(x, n_pts, n_slots):
    return x[3 * n_pts:].reshape(n_slots, 6).copy(), x[:3 * n_pts].reshape(n_pts, 3).copy()


def sparsity_coo(obs):
    """(rows, cols) of the reference's jac_sparsity (bundle_adjuster.py:85-124)"""
    W, N = obs.shape[:2]
    m = valid_mask(obs)
    rows, cols = [], []
    k = 0
    for i in range(W):
        for j in np.nonzero(m[i])[0]:
            for c in range(3):
                rows.append(k); cols.append(3 * j + c)
            for c in range(6):
                rows.append(k); cols.append(3 * N + 6 * i + c)
            k += 1
    return np.array(rows), np.array(cols)


def huber_rho(s, delta=HUBER_DELTA):
    """rho(s) for s = |e|^2 (scipy convention with f_scale = delta)"""
    d2 = delta * delta
    return np.where(s <= d2, s, 2 * delta * np.sqrt(np.maximum(s, 1e-300)) - d2)


def huber_weight(s, delta=HUBER_DELTA):
    d2 = delta * delta
    return np.where(s <= d2, 1.0, delta / np.sqrt(np.maximum(s, 1e-300)))


def cost(K, poses, points, obs, delta=HUBER_DELTA):
    e = residual_vec(K, poses, points, obs)
    s = (e * e).sum(-1)
    return 0.5 * huber_rho(s, delta)[valid_mask(obs)].sum()


def jacobian_blocks(K, poses, points, obs):
    """-> e [W,N,2], Jp [W,N,2,6] (d e / d(rvec,tvec)), Jl [W,N,2,3] (d e / d X), mask [W,N]"""
    K = np.asarray(K, np.float64)
    W, N = obs.shape[:2]
    m = valid_mask(obs)
    e = np.zeros((W, N, 2)); Jp = np.zeros((W, N, 2, 6)); Jl = np.zeros((W, N, 2, 3))
    for i in range(W):
        R = rodrigues_exp(poses[i, :3])
        Jr = right_jacobian(poses[i, :3])
        uv, Xc, p = project(K, poses[i], points)
        # d uv / d Xc  (general 3x3 K)
        A = np.zeros((N, 2, 3))
        A[:, 0, :] = (K[0][None, :] - uv[:, 0:1] * K[2][None, :]) / p[:, 2:3]
        A[:, 1, :] = (K[1][None, :] - uv[:, 1:2] * K[2][None, :]) / p[:, 2:3]
        dXc_dr = -np.einsum('ab,nbc,cd->nad', R, skew(points), Jr)  # -R [X]x J_r
        Jp_i = np.concatenate([np.einsum('nab,nbc->nac', A, dXc_dr), A], axis=2)
        Jl_i = np.einsum('nab,bc->nac', A, R)
        e[i][m[i]] = (uv - obs[i])[m[i]]
        Jp[i][m[i]] = Jp_i[m[i]]
        Jl[i][m[i]] = Jl_i[m[i]]
    return e, Jp, Jl, m


def dense_jacobian_norm_form(K, poses, points, obs):
    """Jacobian of the reference's norm residuals (rows in residual_norm order):
    d|e|/dx = (e^T/|e|) J_e.  Used to compare with the reference's finite-difference J."""
    W, N = obs.shape[:2]
    e, Jp, Jl, m = jacobian_blocks(K, poses, points, obs)
    rows = int(m.sum())
    J = np.zeros((rows, 3 * N + 6 * W))
    k = 0
    for i in range(W):
        for j in np.nonzero(m[i])[0]:
            eh = e[i, j] / np.linalg.norm(e[i, j])
            J[k, 3 * j:3 * j + 3] = eh @ Jl[i, j]
            J[k, 3 * N + 6 * i:3 * N + 6 * i + 6] = eh @ Jp[i, j]
            k += 1
    return J


def normal_equations(K, poses, points, obs, delta=HUBER_DELTA):
    """IRLS-weighted Gauss-Newton system.
    -> dict(Hpp [W,6,6], Hpl [W,N,6,3], Hll [N,3,3], gp [W,6], gl [N,3], cost)"""
    e, Jp, Jl, m = jacobian_blocks(K, poses, points, obs)
    s = (e * e).sum(-1)
    w = huber_weight(s, delta) * m
    Hpp = np.einsum('wn,wnka,wnkb->wab', w, Jp, Jp)
    Hpl = np.einsum('wn,wnka,wnkb->wnab', w, Jp, Jl)
    Hll = np.einsum('wn,wnka,wnkb->nab', w, Jl, Jl)
    gp = np.einsum('wn,wnka,wnk->wa', w, Jp, e)
    gl = np.einsum('wn,wnka,wnk->na', w, Jl, e)
    c = 0.5 * (huber_rho(s, delta) * m).sum()
    return dict(Hpp=Hpp, Hpl=Hpl, Hll=Hll, gp=gp, gl=gl, cost=c)


def schur_system(ne, lam):
    """Damped (Marquardt, diag H) reduced camera system.
    -> S [6W,6W], rhs [6W] (S dp = rhs), Minv [N,3,3] (inverse damped landmark blocks), z = Minv gl"""
    Hpp, Hpl, Hll, gp, gl = ne['Hpp'], ne['Hpl'], ne['Hll'], ne['gp'], ne['gl']
    W, N = Hpl.shape[:2]
    Hll_d = Hll.copy()
    idx = np.arange(3)
    Hll_d[:, idx, idx] += lam * np.maximum(Hll[:, idx, idx], 1e-12)
    Minv = np.linalg.inv(Hll_d)
    S = np.zeros((6 * W, 6 * W))
    idx6 = np.arange(6)
    for i in range(W):
        blk = Hpp[i].copy()
        blk[idx6, idx6] += lam * np.maximum(Hpp[i][idx6, idx6], 1e-12)
        S[6 * i:6 * i + 6, 6 * i:6 * i + 6] = blk
    B = Hpl.transpose(0, 2, 1, 3).reshape(6 * W, N, 3)  # [6W, N, 3]
    BM = np.einsum('pnc,ncd->pnd', B, Minv)
    S -= np.einsum('pnd,qnd->pq', BM, B)
    z = np.einsum('ncd,nd->nc', Minv, gl)
    rhs = -gp.reshape(-1) + np.einsum('pnc,nc->p', B, z)
    return S, rhs, Minv, z, B


def lm_step(ne, lam):
    """Solve (H + lam diag H) d = -g.  -> dposes [W,6], dpoints [N,3], predicted reduction"""
    S, rhs, Minv, z, B = schur_system(ne, lam)
    W = ne['Hpp'].shape[0]
    dp = np.linalg.solve(S, rhs)
    dl = -z - np.einsum('ncd,nd->nc', Minv, np.einsum('pnc,p->nc', B, dp))
    # predicted reduction 0.5 d^T (lam D d - g)
    idx = np.arange(3); idx6 = np.arange(6)
    Dl = np.maximum(ne['Hll'][:, idx, idx], 1e-12)
    Dp = np.maximum(ne['Hpp'][:, idx6, idx6], 1e-12)
    dpw = dp.reshape(W, 6)
    pred = 0.5 * (lam * ((Dl * dl * dl).sum() + (Dp * dpw * dpw).sum())
                  - (ne['gl'] * dl).sum() - (ne['gp'] * dpw).sum())
    return dpw, dl, pred


def solve(K, poses0, points0, obs, max_iters=50, lam0=1e-4, ftol=1e-3, xtol=1e-3, gtol=1e-8,
          delta=HUBER_DELTA, trace=None, lam_min=1e-3):
    """Levenberg-Marquardt with Nielsen damping and a damping floor `lam_min` (the reference fixes no gauge: without a
    floor lambda decays to ~1e-5 and the solver then burns iterations on rejected steps along the 7 gauge directions).  One 'iteration' = linearise at the current x,
    solve one damped step, evaluate the trial cost, accept/reject -- exactly the GPU's loop.
    status: 1 gtol, 2 ftol, 3 xtol, 0 max_iters, 4 damping overflow."""
    poses, points = np.array(poses0, np.float64), np.array(points0, np.float64)
    lam, nu = lam0, 2.0
    F = cost(K, poses, points, obs, delta)
    F0 = F
    status, it, n_acc = 0, 0, 0
    for it in range(1, max_iters + 1):
        ne = normal_equations(K, poses, points, obs, delta)
        ginf = max(np.abs(ne['gp']).max(), np.abs(ne['gl']).max())
        if ginf < gtol:
            status = 1; it -= 1
            break
        dp, dl, pred = lm_step(ne, lam)
        tp, tl = poses + dp, points + dl
        Ft = cost(K, tp, tl, obs, delta)
        step = np.sqrt((dp * dp).sum() + (dl * dl).sum())
        xn = np.sqrt((poses * poses).sum() + (points * points).sum())
        rho = (F - Ft) / pred if pred > 0 else -1.0
        if trace is not None:
            trace.append(dict(it=it, lam=lam, F=F, Ft=Ft, rho=rho, step=step))
        if Ft < F and rho > 0:
            dF = F - Ft
            poses, points, F = tp, tl, Ft
            n_acc += 1
            lam = lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3); nu = 2.0
            lam = max(lam, lam_min)
            if dF < ftol * F:
                status = 2
                break
            if step < xtol * (xtol + xn):
                status = 3
                break
        else:
            if step < xtol * (xtol + xn):
                status = 3
                break
            lam *= nu; nu *= 2.0
            if lam > 1e12:
                status = 4
                break
    return dict(poses=poses, points=points, cost=F, cost0=F0, iters=it, accepted=n_acc, status=status, lam=lam)
