"""Rodrigues vector <-> matrix on the host (closed form of cv2.Rodrigues, used by the reference at
src/bundle_adjuster/bundle_adjuster.py:48,173,211 to pack / unpack window poses)."""
import numpy as np


def rodrigues_vec_to_mat(r):
    r = np.asarray(r, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    c, s = np.cos(theta), np.sin(theta)
    return c * np.eye(3) + (1.0 - c) * np.outer(k, k) + s * Kx


def rodrigues_mat_to_vec(R):
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)          # project onto SO(3) first, as OpenCV does
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
