"""pnp_oracle.py -- CPU ORACLE (test infrastructure, not product code) for the 3D-2D pose step.

Reference call site: Extractor.camera_pose(..., corr='3D-2D'), /root/reference/src/extractor/extractor.py:174-191:
    cv2.solvePnPRansac(pts3d, pts2d, K, None, reprojectionError=max_err_reproj, iterationsCount=1000000, confidence=0.9999)
followed by Rodrigues; Pipeline.step prunes the non-inlier landmarks (pipeline.py:124-137).

PARITY STATUS: *** unpinned and statistical ***.  OpenCV 4.4 runs RANSAC with its own MWC generator, EPnP on 5-point
samples and a Levenberg-Marquardt refinement on the consensus set; none of it can be run here and the reference holds no
vectors for it.  What IS determined by the call's contract is reproduced: the consensus set of a pose is the set of
points with squared reprojection error <= reprojectionError^2, the search stops when a sample of inliers has been drawn
with probability `confidence` (RANSACUpdateNumIters), and the returned pose minimises the reprojection error over the
consensus set.  On data whose inlier set is unambiguous every correct implementation returns that set and that minimiser.
This file DEFINES the algorithm the GPU implements (csrc/vo_pnp.hip) so that the two can be compared hypothesis by
hypothesis:
  * hypothesis h: 4 distinct indices from a counter-based generator (splitmix64 of (seed, h, draw)); Grunert's P3P on the
    first three (quartic in the depth ratio, Ferrari + Newton polish), the fourth picks among the <= 4 solutions;
  * hypotheses are scored in batches -- `first_batch` (32), then `batch` (256) each --; best = most inliers, ties to the smallest h; after
    every batch the iteration bound is updated like OpenCV's RANSACUpdateNumIters with model_points = 4 (OpenCV updates it after every
    hypothesis: the small first batch is the closer restatement where the bound falls to a handful, as in the closed loop);
  * Gauss-Newton (step halving) on (rvec, t) over the consensus set of the best hypothesis.
"""
import math

import numpy as np

MASK64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def sample4(seed, h, n):
    """4 distinct indices in [0, n) for hypothesis h."""
    idx, k = [], 0
    while len(idx) < 4:
        r = splitmix64(((seed & 0xFFFFFF) << 40) ^ ((h & 0xFFFFFFFF) << 8) ^ (k & 0xFF)) if k < 256 else splitmix64(k)
        i = int((r >> 11) % n)
        k += 1
        if i not in idx:
            idx.append(i)
    return idx


def rodrigues(r):
    th = math.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]], float)
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + (math.sin(th) / th) * K + ((1 - math.cos(th)) / (th * th)) * (K @ K)


def log_so3(R):
    c = min(1.0, max(-1.0, (np.trace(R) - 1) / 2))
    th = math.acos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * w
    if math.pi - th < 1e-6:                      # near pi: from the symmetric part
        A = (R + np.eye(3)) / 2
        ax = np.sqrt(np.maximum(np.diag(A), 0))
        i = int(np.argmax(ax))
        v = A[:, i] / ax[i]
        if np.dot(w, v) < 0:
            v = -v
        return th * v
    return th / (2 * math.sin(th)) * w


def cubic_largest_real_root(A, B, C):
    """largest real root of z^3 + A z^2 + B z + C"""
    P = B - A * A / 3.0
    Q = 2.0 * A * A * A / 27.0 - A * B / 3.0 + C
    disc = Q * Q / 4.0 + P * P * P / 27.0
    if disc > 0:
        sq = math.sqrt(disc)
        t = math.copysign(abs(-Q / 2.0 + sq) ** (1.0 / 3.0), -Q / 2.0 + sq) + math.copysign(abs(-Q / 2.0 - sq) ** (1.0 / 3.0), -Q / 2.0 - sq)
    elif P == 0.0:
        t = 0.0
    else:
        m = 2.0 * math.sqrt(-P / 3.0)
        arg = 3.0 * Q / (P * m)
        arg = min(1.0, max(-1.0, arg))
        t = m * math.cos(math.acos(arg) / 3.0)
    return t - A / 3.0


def quartic_real_roots(c4, c3, c2, c1, c0):
    """real roots of c4 x^4 + ... + c0 (Ferrari on the depressed quartic, two Newton steps on the original)"""
    if abs(c4) < 1e-300:
        return []
    a, b, c, d = c3 / c4, c2 / c4, c1 / c4, c0 / c4
    p = b - 3.0 * a * a / 8.0
    q = c - a * b / 2.0 + a * a * a / 8.0
    r = d - a * c / 4.0 + a * a * b / 16.0 - 3.0 * a * a * a * a / 256.0
    ys = []
    if abs(q) < 1e-14 * (1.0 + abs(p) ** 1.5):
        disc = p * p - 4.0 * r
        if disc >= 0:
            sq = math.sqrt(disc)
            for y2 in ((-p + sq) / 2.0, (-p - sq) / 2.0):
                if y2 >= 0:
                    ys += [math.sqrt(y2), -math.sqrt(y2)]
    else:
        z0 = cubic_largest_real_root(2.0 * p, p * p - 4.0 * r, -q * q)
        if z0 > 0:
            s = math.sqrt(z0)
            for sg in (1.0, -1.0):
                # y^2 + sg s y + (p + z0) / 2 - sg q / (2 s) = 0
                bb, cc = sg * s, (p + z0) / 2.0 - sg * q / (2.0 * s)
                disc = bb * bb - 4.0 * cc
                if disc >= 0:
                    sq = math.sqrt(disc)
                    ys += [(-bb + sq) / 2.0, (-bb - sq) / 2.0]
    out = []
    for y in ys:
        x = y - a / 4.0
        for _ in range(2):
            f = (((c4 * x + c3) * x + c2) * x + c1) * x + c0
            fp = ((4.0 * c4 * x + 3.0 * c3) * x + 2.0 * c2) * x + c1
            if fp != 0.0:
                x -= f / fp
        out.append(x)
    return out


def p3p(f, P):
    """f: 3 unit bearings (camera frame), P: 3 world points -> list of (R, t) with x_cam = R X + t"""
    a2 = float(np.dot(P[1] - P[2], P[1] - P[2])); b2 = float(np.dot(P[0] - P[2], P[0] - P[2])); c2 = float(np.dot(P[0] - P[1], P[0] - P[1]))
    if b2 <= 0 or a2 <= 0 or c2 <= 0:
        return []
    ca, cb, cg = float(np.dot(f[1], f[2])), float(np.dot(f[0], f[2])), float(np.dot(f[0], f[1]))
    A, C = a2 / b2, c2 / b2
    A4 = A * A - 2 * A * C - 2 * A + C * C - 4 * C * ca * ca + 2 * C + 1
    A3 = -4 * (A * A * cb - 2 * A * C * cb - A * ca * cg - A * cb + C * C * cb - 2 * C * ca * ca * cb - C * ca * cg + C * cb + ca * cg)
    A2 = 2 * (2 * A * A * cb * cb + A * A - 4 * A * C * cb * cb - 2 * A * C - 4 * A * ca * cb * cg - 2 * A * cg * cg + 2 * C * C * cb * cb + C * C
              - 2 * C * ca * ca - 4 * C * ca * cb * cg + 2 * ca * ca + 2 * cg * cg - 1)
    A1 = -4 * (A * A * cb - 2 * A * C * cb - A * ca * cg - 2 * A * cb * cg * cg + A * cb + C * C * cb - C * ca * cg - C * cb + ca * cg)
    A0 = A * A - 2 * A * C - 4 * A * cg * cg + 2 * A + C * C - 2 * C + 1
    sols = []
    qq = A - C
    e1 = P[1] - P[0]; e1 = e1 / math.sqrt(np.dot(e1, e1))
    e3 = np.cross(e1, P[2] - P[0]); n3 = math.sqrt(np.dot(e3, e3))
    if n3 <= 0:
        return []
    e3 = e3 / n3
    e2 = np.cross(e3, e1)
    for v in quartic_real_roots(A4, A3, A2, A1, A0):
        den = 2.0 * (cg - v * ca)
        if v <= 0 or abs(den) < 1e-12:
            continue
        u = ((qq - 1.0) * v * v - 2.0 * qq * cb * v + 1.0 + qq) / den
        w = 1.0 + v * v - 2.0 * v * cb
        if u <= 0 or w <= 0:
            continue
        s1 = math.sqrt(b2 / w)
        Q = [s1 * f[0], u * s1 * f[1], v * s1 * f[2]]
        g1 = Q[1] - Q[0]; g1 = g1 / math.sqrt(np.dot(g1, g1))
        g3 = np.cross(g1, Q[2] - Q[0]); m3 = math.sqrt(np.dot(g3, g3))
        if m3 <= 0:
            continue
        g3 = g3 / m3
        g2 = np.cross(g3, g1)
        R = np.outer(g1, e1) + np.outer(g2, e2) + np.outer(g3, e3)
        t = Q[0] - R @ P[0]
        sols.append((R, t))
    return sols


def reproj_err2(K, R, t, X, uv):
    Xc = X @ R.T + t
    p = Xc @ K.T
    with np.errstate(divide="ignore", invalid="ignore"):
        d = p[:, :2] / p[:, 2:3] - uv
    return (d * d).sum(1)


def hypothesis(K, Kinv, X, uv, idx):
    f = []
    for i in idx[:3]:
        b = Kinv @ np.array([uv[i, 0], uv[i, 1], 1.0])
        f.append(b / math.sqrt(np.dot(b, b)))
    best, best_e = None, float("inf")
    for R, t in p3p(f, [X[i] for i in idx[:3]]):
        e = reproj_err2(K, R, t, X[idx[3]:idx[3] + 1], uv[idx[3]:idx[3] + 1])[0]
        if e < best_e:                       # NaN never wins
            best, best_e = (R, t), e
    return best


def update_num_iters(p, ep, model_points, max_iters):
    """OpenCV RANSACUpdateNumIters (calib3d/ptsetreg.cpp)"""
    p = min(max(p, 0.0), 1.0); ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, 2.2250738585072014e-308)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < 2.2250738585072014e-308:
        return 0
    num, denom = math.log(num), math.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(round(num / denom))


def refine(K, rvec, t, X, uv, iters=20):
    """Gauss-Newton on the pose (left rotation update exp(w) R, translation), step halving, over the given points
    -> rvec, t, cost (sum of squared pixel errors)"""
    def cost(Rm, tt):
        return float(reproj_err2(K, Rm, tt, X, uv).sum())
    R, tt = rodrigues(np.array(rvec, float)), np.array(t, float)
    c = cost(R, tt)
    for _ in range(iters):
        Xc = X @ R.T + tt
        p = Xc @ K.T
        ip2 = 1.0 / p[:, 2]
        u, v = p[:, 0] * ip2, p[:, 1] * ip2
        e = np.stack([u - uv[:, 0], v - uv[:, 1]], 1)
        A = np.zeros((len(X), 2, 3))                      # d(u, v) / d x_cam
        for cidx in range(3):
            A[:, 0, cidx] = (K[0, cidx] - u * K[2, cidx]) * ip2
            A[:, 1, cidx] = (K[1, cidx] - v * K[2, cidx]) * ip2
        J = np.zeros((len(X), 2, 6))
        RX = X @ R.T
        for k in range(3):                                # d(exp(w) R X) / d w_k = e_k x (R X)
            G = np.zeros((3, 3)); G[(k + 1) % 3, (k + 2) % 3] = -1; G[(k + 2) % 3, (k + 1) % 3] = 1
            J[:, :, k] = np.einsum("nij,nj->ni", A, RX @ G.T)
        J[:, :, 3:] = A
        H = np.einsum("nki,nkj->ij", J, J); g = np.einsum("nki,nk->i", J, e)
        try:
            d = np.linalg.solve(H, -g)
        except np.linalg.LinAlgError:
            break
        step, ok = 1.0, False
        for _h in range(6):
            Rn, tn = rodrigues(step * d[:3]) @ R, tt + step * d[3:]
            cn = cost(Rn, tn)
            if cn < c:
                ok = True
                break
            step *= 0.5
        if not ok:
            break
        small = (c - cn) <= 1e-12 * max(c, 1e-300)
        R, tt, c = Rn, tn, cn
        if small:
            break
    return log_so3(R), tt, c


def pnp_ransac(K, X, uv, thr=2.0, conf=0.9999, max_iters=1000000, seed=0, batch=256, return_info=False, first_batch=32):
    K = np.asarray(K, float); Kinv = np.linalg.inv(K)
    X = np.asarray(X, np.float32).astype(float).reshape(-1, 3); uv = np.asarray(uv, np.float32).astype(float).reshape(-1, 2)
    n = len(X)
    thr2 = thr * thr
    best = dict(count=0, h=-1, R=None, t=None)
    niters, h0 = max_iters, 0
    while h0 < niters and n >= 4:
        nb = first_batch if h0 == 0 else batch
        for h in range(h0, h0 + nb):
            hyp = hypothesis(K, Kinv, X, uv, sample4(seed, h, n))
            if hyp is None:
                continue
            e2 = reproj_err2(K, hyp[0], hyp[1], X, uv)
            cnt = int((e2 <= thr2).sum())
            if cnt > best["count"]:
                best = dict(count=cnt, h=h, R=hyp[0], t=hyp[1])
        h0 += nb
        if best["count"] > 0:
            niters = min(niters, update_num_iters(conf, (n - best["count"]) / n, 4, max_iters))
    if best["count"] < 4:
        return (None, None, np.zeros(0, int)) + ((dict(hyps=h0, best=-1),) if return_info else ())
    mask = reproj_err2(K, best["R"], best["t"], X, uv) <= thr2
    r, t, c = refine(K, log_so3(best["R"]), best["t"], X[mask], uv[mask])
    out = (r, t, np.nonzero(mask)[0])
    return out + ((dict(hyps=h0, best=best["h"], count=best["count"], cost=c),) if return_info else ())
