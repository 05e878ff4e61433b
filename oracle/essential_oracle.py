"""essential_oracle.py -- CPU ORACLE (test infrastructure, not product code) for the 2D-2D bootstrap pose.

Reference call site: Extractor.camera_pose(..., corr='2D-2D'), /root/reference/src/extractor/extractor.py:162-172:
    E, mask = cv2.findEssentialMat(p1, p2, K, prob=0.9999, method=cv2.RANSAC, mask=None, threshold=1.0)
    retval, R, t, mask = cv2.recoverPose(E, p1[inliers], p2[inliers], K)
(used once per sequence by Pipeline._get_init_state, pipeline.py:42-90; SURVEY.md 8f "next" row 4).

PARITY STATUS: *** unpinned and statistical ***.  OpenCV 4.4 (modules/calib3d/src/five-point.cpp, ptsetreg.cpp) draws
its samples from its own generator and solves the minimal problem with Nister's five-point method; neither OpenCV nor
vectors of it exist here.  Restated from the published algorithm and the call's contract:
  * points are normalised with fx, fy, cx, cy of K and the pixel threshold is divided by (fx + fy) / 2;
  * minimal solver: Nister, "An efficient solution to the five-point relative pose problem" (PAMI 2004): 4-dimensional
    null space of the 5 epipolar constraints, the 10 cubic constraints det E = 0 and 2 E E^T E - tr(E E^T) E = 0 as a
    10 x 20 matrix, Gauss-Jordan elimination, the 3 x 3 polynomial matrix B(z) and its 10th-degree determinant, one
    essential matrix per real root (<= 10 per sample);
  * consensus: squared Sampson distance <= threshold^2; the iteration bound follows RANSACUpdateNumIters with 5 model
    points, default bound 1000; the best model is returned as it is (OpenCV does not re-fit either);
  * recoverPose: E = U diag(1,1,0) V^T, the four (R, t) candidates, DLT triangulation of every inlier against
    [I | 0], a point is "good" when its depth is in (0, distance_thresh = 50) in both cameras, most good points win
    (OpenCV's order of preference on ties).
This file DEFINES what csrc/vo_essential.hip implements (hypothesis h = 5 indices from splitmix64(seed, h, draw),
roots in ascending order, batches of 256 hypotheses, ties to the smallest (h, root)) so that the two can be compared
hypothesis by hypothesis.  Real roots are isolated between the critical points of the polynomial (recursively through
its derivatives) and bisected -- no complex arithmetic, the same code on both sides.
"""
import math

import numpy as np

from pnp_oracle import splitmix64, update_num_iters

# exponent tuples (x, y, z) of the polynomial bases
P1 = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
P2 = [(2, 0, 0), (0, 2, 0), (0, 0, 2), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
# column order of the 10 x 20 constraint matrix: the ten monomials that are eliminated, then [x, y, 1] (x) powers of z
P3 = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
      (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]
MUL11 = [[P2.index(tuple(a + b for a, b in zip(P1[i], P1[j]))) for j in range(4)] for i in range(4)]
MUL21 = [[P3.index(tuple(a + b for a, b in zip(P2[i], P1[j]))) for j in range(4)] for i in range(10)]


def sample5(seed, h, n):
    """5 distinct indices in [0, n) for hypothesis h (same generator as pnp_oracle.sample4)."""
    idx, k = [], 0
    while len(idx) < 5:
        r = splitmix64(((seed & 0xFFFFFF) << 40) ^ ((h & 0xFFFFFFFF) << 8) ^ (k & 0xFF)) if k < 256 else splitmix64(k)
        i = int((r >> 11) % n)
        k += 1
        if i not in idx:
            idx.append(i)
    return idx


def mul11(a, b):
    o = [0.0] * 10
    for i in range(4):
        for j in range(4):
            o[MUL11[i][j]] += a[i] * b[j]
    return o


def mul21(a, b):
    o = [0.0] * 20
    for i in range(10):
        for j in range(4):
            o[MUL21[i][j]] += a[i] * b[j]
    return o


def null_space_5x9(q1, q2):
    """orthonormal basis (4 x 9) of the null space of the five epipolar constraints q2^T E q1 = 0 (E row-major):
    Gauss-Jordan with complete pivoting, free columns -> basis vectors, modified Gram-Schmidt.  None if rank < 5."""
    A = [[q2[i][0] * q1[i][0], q2[i][0] * q1[i][1], q2[i][0], q2[i][1] * q1[i][0], q2[i][1] * q1[i][1], q2[i][1],
          q1[i][0], q1[i][1], 1.0] for i in range(5)]
    piv = [-1] * 5
    used = [False] * 9
    for r in range(5):
        best, br, bc = 0.0, -1, -1
        for rr in range(r, 5):
            for c in range(9):
                if not used[c] and abs(A[rr][c]) > best:
                    best, br, bc = abs(A[rr][c]), rr, c
        if not best > 1e-12:
            return None
        A[r], A[br] = A[br], A[r]
        piv[r] = bc; used[bc] = True
        inv = 1.0 / A[r][bc]
        A[r] = [v * inv for v in A[r]]
        for rr in range(5):
            if rr != r:
                f = A[rr][bc]
                A[rr] = [A[rr][c] - f * A[r][c] for c in range(9)]
    basis = []
    for f in range(9):
        if used[f]:
            continue
        v = [0.0] * 9
        v[f] = 1.0
        for r in range(5):
            v[piv[r]] = -A[r][f]
        basis.append(v)
    for i in range(4):                                   # modified Gram-Schmidt
        for j in range(i):
            d = sum(basis[i][c] * basis[j][c] for c in range(9))
            basis[i] = [basis[i][c] - d * basis[j][c] for c in range(9)]
        nn = math.sqrt(sum(v * v for v in basis[i]))
        basis[i] = [v / nn for v in basis[i]]
    return basis


def constraint_matrix(basis):
    """10 x 20: det E (row 0) and E E^T E - tr(E E^T) E / 2 (rows 1..9), E = x B0 + y B1 + z B2 + B3."""
    e = [[[basis[k][3 * i + j] for k in range(4)] for j in range(3)] for i in range(3)]        # e[i][j] = p1
    sub = lambda a, b: [x - y for x, y in zip(a, b)]
    add = lambda a, b: [x + y for x, y in zip(a, b)]
    rows = []
    d0 = mul21(sub(mul11(e[1][1], e[2][2]), mul11(e[1][2], e[2][1])), e[0][0])
    d1 = mul21(sub(mul11(e[1][0], e[2][2]), mul11(e[1][2], e[2][0])), e[0][1])
    d2 = mul21(sub(mul11(e[1][0], e[2][1]), mul11(e[1][1], e[2][0])), e[0][2])
    rows.append(add(sub(d0, d1), d2))
    eet = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(i, 3):
            s = mul11(e[i][0], e[j][0])
            s = add(s, mul11(e[i][1], e[j][1]))
            s = add(s, mul11(e[i][2], e[j][2]))
            eet[i][j] = s; eet[j][i] = s
    tr = add(add(eet[0][0], eet[1][1]), eet[2][2])
    lam = [[sub(eet[i][j], [0.5 * v for v in tr]) if i == j else eet[i][j] for j in range(3)] for i in range(3)]
    for i in range(3):
        for j in range(3):
            s = mul21(lam[i][0], e[0][j])
            s = add(s, mul21(lam[i][1], e[1][j]))
            s = add(s, mul21(lam[i][2], e[2][j]))
            rows.append(s)
    return rows


def gauss_jordan_10(A):
    """reduced row echelon form on the first 10 columns (partial pivoting); False if singular"""
    for c in range(10):
        best, br = 0.0, -1
        for r in range(c, 10):
            if abs(A[r][c]) > best:
                best, br = abs(A[r][c]), r
        if not best > 1e-14:
            return False
        A[c], A[br] = A[br], A[c]
        inv = 1.0 / A[c][c]
        A[c] = [v * inv for v in A[c]]
        for r in range(10):
            if r != c:
                f = A[r][c]
                if f != 0.0:
                    A[r] = [A[r][k] - f * A[c][k] for k in range(20)]
    return True


def poly_mul(a, b):
    """coefficients in ASCENDING powers"""
    o = [0.0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            o[i + j] += x * y
    return o


def poly_eval(p, deg, x):
    s = p[deg]
    for k in range(deg - 1, -1, -1):
        s = s * x + p[k]
    return s


def real_roots(p):
    """real roots (ascending) of a polynomial given by ascending coefficients.  The roots of the derivative chain
    p^(d-1), ..., p', p are found in turn: between two consecutive critical points (and out to the Cauchy bound) the
    polynomial is monotone, so a sign change brackets exactly one root, which is bisected to the last bit."""
    scale = max(abs(v) for v in p)
    if not scale > 0:
        return []
    deg = len(p) - 1
    while deg > 0 and abs(p[deg]) <= 1e-14 * scale:
        deg -= 1
    if deg == 0:
        return []
    chain = [[v for v in p[:deg + 1]]]                   # chain[k] = k-th derivative
    for k in range(1, deg):
        q = chain[-1]
        chain.append([q[i] * i for i in range(1, len(q))])
    lin = chain[deg - 1]
    roots = [-lin[0] / lin[1]]
    for k in range(deg - 2, -1, -1):
        q = chain[k]
        d = len(q) - 1
        bound = 1.0 + max(abs(q[i] / q[d]) for i in range(d))
        edges = [-bound] + roots + [bound]
        new = []
        for a, b in zip(edges[:-1], edges[1:]):
            fa, fb = poly_eval(q, d, a), poly_eval(q, d, b)
            if (fa < 0) == (fb < 0):
                continue
            lo, hi = a, b
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                if mid == lo or mid == hi:
                    break
                fm = poly_eval(q, d, mid)
                if (fm < 0) == (fa < 0):
                    lo = mid
                else:
                    hi = mid
            new.append(0.5 * (lo + hi))
        roots = new
        if not roots and k > 0:
            # a derivative without real roots is monotone-free of sign changes: the next level is monotone on the
            # whole line, which the [-bound, bound] edge pair above handles with an empty list
            pass
    return roots


def five_point(q1, q2):
    """q1, q2: 5 x 2 normalised image points.  -> list of 3 x 3 essential matrices (Frobenius norm 1), ascending z."""
    basis = null_space_5x9(q1, q2)
    if basis is None:
        return []
    A = constraint_matrix(basis)
    if not gauss_jordan_10(A):
        return []
    # rows 4..9 give x^2 z, x^2, y^2 z, y^2, xyz, xy in terms of [x z^2, x z, x, y z^2, y z, y, z^3, z^2, z, 1]:
    # (row of m z) - z (row of m) = x bx(z) + y by(z) + b1(z) = 0 with deg bx = by = 3, deg b1 = 4
    B = []
    for ra, rb in ((4, 5), (6, 7), (8, 9)):
        e, f = A[ra][10:], A[rb][10:]
        bx = [e[2], e[1] - f[2], e[0] - f[1], -f[0]]                        # ascending powers of z
        by = [e[5], e[4] - f[5], e[3] - f[4], -f[3]]
        b1 = [e[9], e[8] - f[9], e[7] - f[8], e[6] - f[7], -f[6]]
        B.append((bx, by, b1))
    sub = lambda a, b: [x - y for x, y in zip(a, b)]
    add = lambda a, b: [x + y for x, y in zip(a, b)]
    m0 = sub(poly_mul(B[1][0], B[2][1]), poly_mul(B[1][1], B[2][0]))        # minors of the b1 column, degree 6
    m1 = sub(poly_mul(B[0][0], B[2][1]), poly_mul(B[0][1], B[2][0]))
    m2 = sub(poly_mul(B[0][0], B[1][1]), poly_mul(B[0][1], B[1][0]))
    det = add(sub(poly_mul(B[0][2], m0), poly_mul(B[1][2], m1)), poly_mul(B[2][2], m2))    # degree 10
    out = []
    for z in real_roots(det):
        rows = [[poly_eval(b[0], 3, z), poly_eval(b[1], 3, z), poly_eval(b[2], 4, z)] for b in B]
        best, nv = -1.0, None
        for a, b in ((0, 1), (0, 2), (1, 2)):                               # null vector of B(z): the largest cross product
            c = [rows[a][1] * rows[b][2] - rows[a][2] * rows[b][1], rows[a][2] * rows[b][0] - rows[a][0] * rows[b][2],
                 rows[a][0] * rows[b][1] - rows[a][1] * rows[b][0]]
            nn = c[0] * c[0] + c[1] * c[1] + c[2] * c[2]
            if nn > best:
                best, nv = nn, c
        if not best > 0 or abs(nv[2]) <= 1e-10 * math.sqrt(best):
            continue
        x, y = nv[0] / nv[2], nv[1] / nv[2]
        E = [x * basis[0][k] + y * basis[1][k] + z * basis[2][k] + basis[3][k] for k in range(9)]
        nn = math.sqrt(sum(v * v for v in E))
        if not nn > 0 or not math.isfinite(nn):
            continue
        out.append(np.array([v / nn for v in E]).reshape(3, 3))
    return out


def sampson_err2(E, q1, q2):
    """squared Sampson distance of every correspondence (OpenCV EMEstimatorCallback::computeError)"""
    x1 = np.concatenate([q1, np.ones((len(q1), 1))], 1); x2 = np.concatenate([q2, np.ones((len(q2), 1))], 1)
    Ex1 = x1 @ E.T; Etx2 = x2 @ E
    num = (x2 * Ex1).sum(1)
    with np.errstate(invalid="ignore", divide="ignore"):
        return num * num / (Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2)


def normalise(K, p):
    K = np.asarray(K, float)
    p = np.asarray(p, np.float32).astype(float).reshape(-1, 2)
    return np.stack([(p[:, 0] - K[0, 2]) / K[0, 0], (p[:, 1] - K[1, 2]) / K[1, 1]], 1)


def triangulate(P0, P1, a, b):
    """OpenCV cvTriangulatePoints for one correspondence of normalised points -> homogeneous 4-vector"""
    A = np.stack([a[0] * P0[2] - P0[0], a[1] * P0[2] - P0[1], b[0] * P1[2] - P1[0], b[1] * P1[2] - P1[1]])
    return np.linalg.svd(A)[2][3]


def decompose(E):
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, 1, 0], [-1, 0, 0], [0, 0, 1]])
    return U @ W @ Vt, U @ W.T @ Vt, U[:, 2].copy()


def recover_pose(E, q1, q2, dist=50.0):
    """-> R, t, number of good points of the chosen candidate, good counts of all four"""
    R1, R2, t = decompose(E)
    cands = [(R1, t), (R2, t), (R1, -t), (R2, -t)]
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    good = []
    for R, tt in cands:
        P1 = np.hstack([R, tt.reshape(3, 1)])
        g = 0
        for a, b in zip(q1, q2):
            Q = triangulate(P0, P1, a, b)
            if not Q[2] * Q[3] > 0:
                continue
            X = Q[:3] / Q[3]
            if not X[2] < dist:
                continue
            z2 = P1[2, :3] @ X + P1[2, 3]
            g += 1 if (z2 > 0 and z2 < dist) else 0
        good.append(g)
    g1, g2, g3, g4 = good
    if g1 >= g2 and g1 >= g3 and g1 >= g4:
        k = 0
    elif g2 >= g1 and g2 >= g3 and g2 >= g4:
        k = 1
    elif g3 >= g1 and g3 >= g2 and g3 >= g4:
        k = 2
    else:
        k = 3
    return cands[k][0], cands[k][1], good[k], good


def essential_ransac(K, p1, p2, thr=1.0, prob=0.9999, max_iters=1000, seed=0, dist=50.0, batch=256, return_info=False):
    """-> E (unit Frobenius norm), R, t, inlier indices [, info]; (None, None, None, []) if no model had >= 5 inliers"""
    K = np.asarray(K, float)
    q1, q2 = normalise(K, p1), normalise(K, p2)
    n = len(q1)
    t2 = (thr / ((K[0, 0] + K[1, 1]) / 2.0)) ** 2
    best = dict(count=4, h=-1, k=-1, E=None)
    niters, h0 = max_iters, 0
    while h0 < niters and n >= 5:
        for h in range(h0, h0 + batch):
            idx = sample5(seed, h, n)
            for k, E in enumerate(five_point(q1[idx].tolist(), q2[idx].tolist())):
                cnt = int((sampson_err2(E, q1, q2) <= t2).sum())
                if cnt > best["count"]:
                    best = dict(count=cnt, h=h, k=k, E=E)
        h0 += batch
        if best["E"] is not None:
            niters = min(niters, update_num_iters(prob, (n - best["count"]) / n, 5, max_iters))
    if best["E"] is None:
        return (None, None, None, np.zeros(0, int)) + ((dict(hyps=h0, best=-1, k=-1),) if return_info else ())
    mask = sampson_err2(best["E"], q1, q2) <= t2
    R, t, g, good = recover_pose(best["E"], q1[mask], q2[mask], dist)
    out = (best["E"], R, t, np.nonzero(mask)[0])
    return out + ((dict(hyps=h0, best=best["h"], k=best["k"], count=best["count"], n_good=g, good=good),) if return_info else ())
