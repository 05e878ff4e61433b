"""match_oracle.py -- CPU ORACLE (test infrastructure, not product code) for the bootstrap's descriptor matching.

Reference call site: Extractor.match, /root/reference/src/extractor/extractor.py:134-145:
    matches = cv2.BFMatcher().knnMatch(desc_1, desc_2, k=2);  keep m if m.distance < 0.8 * n.distance
PARITY STATUS: unpinned at the OpenCV boundary (no OpenCV here, no vectors in the reference).  Restated from
modules/features2d/src/matchers.cpp + modules/core/src/batch_distance.cpp @ 4.4.0: NORM_L2 distance = sqrt of the sum of
squared float32 differences, the K best per query in ascending order, the first of equal distances wins.  OpenCV
accumulates the sum in float32 SIMD lanes; here it is accumulated in float64 and rounded once (order-independent).
"""
import numpy as np


def knn2(desc1, desc2):
    """-> idx (n1, 2) int32 (-1 = none), dist (n1, 2) float32"""
    a = np.asarray(desc1, np.float32).astype(np.float64); b = np.asarray(desc2, np.float32).astype(np.float64)
    n1, n2 = len(a), len(b)
    idx = np.full((n1, 2), -1, np.int32); dist = np.full((n1, 2), np.inf, np.float32)
    for q in range(n1):
        e = a[q][None, :] - b
        d = np.sqrt((e * e).sum(1).astype(np.float32))            # float32 sqrt of the once-rounded sum
        order = [j for j in np.argsort(d, kind="stable") if not np.isnan(d[j])][:2]
        for s, j in enumerate(order):
            idx[q, s] = j; dist[q, s] = d[j]
    return idx, dist


def ratio_matches(desc1, desc2, ratio=0.8):
    """-> list of (queryIdx, trainIdx, distance) passing Lowe's ratio test, as the reference's Extractor.match"""
    idx, dist = knn2(desc1, desc2)
    return [(q, int(idx[q, 0]), float(dist[q, 0])) for q in range(len(idx)) if float(dist[q, 0]) < ratio * float(dist[q, 1])]
