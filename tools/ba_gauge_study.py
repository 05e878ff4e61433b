"""Round-4 study: does motion along the 7 gauge directions cause the LM's long tails?  CPU, numpy (oracle/ba_oracle.py).

For problems of the bench's BA bank (2000 landmarks, 10 poses) and of the closed loop (tests/golden/pipe_w4.npz replayed through the table model):
  * every LM step is split into its component inside the span of the 7 generators of the world similarity (translation, rotation, scale --
    poses AND points) and the rest; printed with the cost the gauge-free step alone reaches;
  * three solvers are run to their own stop: the shipped one (Marquardt damping with floor 1e-3), the same without a floor, and one whose reduced
    camera system is made regular by mu G G^T (G = pose part of the generators: a gauge-free Gauss-Newton step, floor 1e-9).
Usage: python tools/ba_gauge_study.py > profiles/r04_ba_gauge_study.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "visual-odom-pipeline_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import ba_oracle as bo  # noqa: E402
from vo_mi355x import synthetic as syn  # noqa: E402

BA_VARIANTS = [(0.3, 0.3, 0.02, 1.0), (0.3, 0.6, 0.05, 1.0), (0.5, 0.3, 0.02, 0.9), (0.3, 1.0, 0.10, 1.0), (0.3, 0.3, 0.02, 0.8),
               (0.8, 0.5, 0.05, 1.0), (0.3, 0.8, 0.02, 0.95), (0.4, 0.4, 0.08, 0.85)]


def gauge_full(poses, points):
    """the 7 generators over [poses (rvec, tvec) | points]: x_w' = x_w + eps;  x_w' = x_w + omega x x_w;  x_w' = (1 + sigma) x_w"""
    W, N = len(poses), len(points)
    G = np.zeros((6 * W + 3 * N, 7))
    for i in range(W):
        R = bo.rodrigues_exp(poses[i, :3])
        Ji = np.linalg.inv(bo.right_jacobian(poses[i, :3]))
        G[6 * i + 3:6 * i + 6, 0:3] = -R
        G[6 * i:6 * i + 3, 3:6] = -Ji
        G[6 * i + 3:6 * i + 6, 6] = poses[i, 3:]
    o = 6 * W
    for k in range(3):
        G[o + k::3, k] = 1.0
        e = np.zeros(3); e[k] = 1
        G[o:, 3 + k] = np.cross(e, points).reshape(-1)
    G[o:, 6] = points.reshape(-1)
    return G


def lm_step_gauge(ne, lam, poses):
    S, rhs, Minv, z, B = bo.schur_system(ne, lam)
    W = ne['Hpp'].shape[0]
    Q, _ = np.linalg.qr(gauge_full(poses, np.zeros((0, 3)))[:6 * W])
    S2 = S + (np.trace(S) / S.shape[0]) * (Q @ Q.T)
    dp = np.linalg.solve(S2, rhs)
    dl = -z - np.einsum('ncd,nd->nc', Minv, np.einsum('pnc,p->nc', B, dp))
    idx, idx6 = np.arange(3), np.arange(6)
    Dl = np.maximum(ne['Hll'][:, idx, idx], 1e-12); Dp = np.maximum(ne['Hpp'][:, idx6, idx6], 1e-12)
    dpw = dp.reshape(W, 6)
    pred = 0.5 * (lam * ((Dl * dl * dl).sum() + (Dp * dpw * dpw).sum()) - (ne['gl'] * dl).sum() - (ne['gp'] * dpw).sum())
    return dpw, dl, pred


def solve(K, poses0, points0, obs, gauge, lam_min, max_iters=50, ftol=1e-3, xtol=1e-3, split=None):
    poses, points = np.array(poses0, float), np.array(points0, float)
    lam, nu = 1e-4, 2.0
    F = bo.cost(K, poses, points, obs)
    status, it = 0, 0
    for it in range(1, max_iters + 1):
        ne = bo.normal_equations(K, poses, points, obs)
        dp, dl, pred = lm_step_gauge(ne, lam, poses) if gauge else bo.lm_step(ne, lam)
        tp, tl = poses + dp, points + dl
        Ft = bo.cost(K, tp, tl, obs)
        step = np.sqrt((dp * dp).sum() + (dl * dl).sum()); xn = np.sqrt((poses * poses).sum() + (points * points).sum())
        rho = (F - Ft) / pred if pred > 0 else -1.0
        if split is not None:
            G = gauge_full(poses, points)
            d = np.concatenate([dp.reshape(-1), dl.reshape(-1)])
            dg = G @ np.linalg.lstsq(G, d, rcond=None)[0]
            free = d - dg
            W = len(poses)
            Ff = bo.cost(K, poses + free[:6 * W].reshape(W, 6), points + free[6 * W:].reshape(-1, 3), obs)
            split.append((it, lam, F, Ft, Ff, np.linalg.norm(d), np.linalg.norm(dg), np.linalg.norm(free), xtol * (xtol + xn)))
        if Ft < F and rho > 0:
            dF = F - Ft
            poses, points, F = tp, tl, Ft
            lam = max(lam * max(1 / 3, 1 - (2 * rho - 1) ** 3), lam_min); nu = 2.0
            if dF < ftol * F:
                status = 2; break
            if step < xtol * (xtol + xn):
                status = 3; break
        else:
            if step < xtol * (xtol + xn):
                status = 3; break
            lam *= nu; nu *= 2
    return dict(cost=F, iters=it, status=status)


def main():
    print(__doc__.split("Usage")[0])
    print("1. LM steps of the shipped solver split into gauge / gauge-free parts (bench bank problems whose solve is long)")
    for b, k in ((3, 0), (5, 5), (1, 3)):
        v = BA_VARIANTS[k]
        s = syn.make_ba_scene(n_pts=2000, n_slots=10, seed=1000 + b + 7919 * k, obs_noise=v[0], pt_noise=v[1], pose_noise=v[2], visibility=v[3])
        tr = []
        r = solve(s['K'], s['poses0'], s['points0'], s['obs'], False, 1e-3, split=tr)
        print("  problem (seed row %d, variant %d): %d iterations, status %d" % (b, k, r['iters'], r['status']))
        for t in tr:
            print("    it %2d lambda %.1e  F %10.3f -> %10.3f (gauge-free step alone: %10.3f)  |step| %7.3f = gauge %6.3f (+) free %7.3f   xtol threshold %.3f"
                  % (t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]))
    print("\\n   -> the steps are NOT gauge drift: the gauge component is 5-25 %% of the step's norm, the gauge-free step reaches the same cost, and the\\n"
          "      gauge-free norm (20-60 units) is as far above the xtol threshold (1.8) as the full norm.  What is slow is real: weakly observable\\n"
          "      depths of far points under forward motion (a step of +-0.5 m on 2 000 points has norm 22) and IRLS reweighting.")
    print("\\n2. iterations to the solver's own stop (ftol = xtol = 1e-3, cap 50): shipped (floor 1e-3) | no floor (1e-9) | gauge-regularised S + floor 1e-9")
    hist = collections.defaultdict(list)
    for b in range(6):
        for k, v in enumerate(BA_VARIANTS):
            s = syn.make_ba_scene(n_pts=2000, n_slots=10, seed=1000 + b + 7919 * k, obs_noise=v[0], pt_noise=v[1], pose_noise=v[2], visibility=v[3])
            args = (s['K'], s['poses0'], s['points0'], s['obs'])
            res = [solve(*args, False, 1e-3), solve(*args, False, 1e-9), solve(*args, True, 1e-9)]
            print("  bank %d/%d   " % (b, k) + " | ".join("%2d it  cost %9.2f (%d)" % (r['iters'], r['cost'], r['status']) for r in res), flush=True)
            for n, r in zip(("shipped", "no floor", "gauge-regularised"), res):
                hist[n].append(r['iters'])
    for n, h in hist.items():
        print("  %-18s mean %.2f  max %2d  histogram %s" % (n, np.mean(h), max(h), sorted(collections.Counter(h).items())))
    print("\\n   -> neither variant shortens the tail (means 4.9 / 5.3 / 5.2, maxima 9 / 12 / 11): a termination or step rule that ignores the gauge\\n"
          "      directions does not do what VERDICT r3 item 4a hoped; the shipped floor stays.")


if __name__ == "__main__":
    main()
