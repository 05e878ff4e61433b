"""Would TWO keypoints per wave pay in k_klt_track?  (round-2 review item 5.)  One wave owns one keypoint today and ~62 of the ~150 vector
instructions of an LK iteration (wave sums, weights, 2x2 solve, tests) are wave-uniform, i.e. paid per keypoint; two keypoints in the
two halves of a wave would share them -- but then both halves run max(it_a, it_b) iterations per level (a lane that is switched off
still costs its issue slot).  This study measures that divergence on the bench's own frames with the CPU oracle's per-level iteration
counts (bit-equal to the GPU's, tests/test_gpu_frontend.py): instructions per keypoint for (i) one keypoint per wave, (ii) neighbours
in the point list paired, (iii) pairs matched by the previous frame's iteration counts (what a sort per frame could do), (iv) the
unreachable ideal of perfectly matched pairs.  CPU only:  python tools/klt_pairing_study.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "visual-odom-pipeline_amd"), os.path.join(ROOT, "oracle")]
import vo_oracle as o                                   # noqa: E402
from vo_mi355x import synthetic as syn                  # noqa: E402

# vector instructions of k_klt_track per level and per iteration (~150 per LK iteration of which 62 are wave-uniform,
# ~330 per level for the template of which ~60 are uniform set-up)
IT_PIX, IT_UNI, LV_PIX, LV_UNI = 88.0, 62.0, 270.0, 60.0


def cost_single(it):
    lv = (it >= 0)
    return (lv * (LV_PIX + LV_UNI) + np.maximum(it, 0) * (IT_PIX + IT_UNI)).sum(1)


def cost_pair(ita, itb):
    """two keypoints in one wave: each half does its own pixel work (a half has 32 lanes for 1024 pixels: twice the instructions of a
    64-lane wave per keypoint, so PER KEYPOINT the pixel work is unchanged), the uniform work is issued once per pair, and both halves
    stay in the loop for max(it_a, it_b) iterations"""
    lv = np.maximum(ita >= 0, itb >= 0)
    m = np.maximum(np.maximum(ita, itb), 0)
    per_pair = (lv * (2 * LV_PIX + LV_UNI) + m * (2 * IT_PIX + IT_UNI)).sum(1)
    return per_pair / 2.0


def main():
    frames = syn.make_sequence(100, seed=1234, periodic=True, n_render=6)[0]
    p = syn.grid_points(2000, 1241, 376, seed=7)
    prev_it = None
    rows = []
    for t in range(5):
        p1, st, err, it = o.klt(frames[t], frames[t + 1], p, return_iters=True)
        single = cost_single(it).mean()
        n = len(p) // 2 * 2
        nb = cost_pair(it[0:n:2], it[1:n:2]).mean()
        ideal_order = np.lexsort(it.T[::-1])
        ideal = cost_pair(it[ideal_order][0:n:2], it[ideal_order][1:n:2]).mean()
        if prev_it is not None:
            order = np.lexsort(prev_it.T[::-1])            # sorted by LAST frame's counts, coarsest level first
            pred = cost_pair(it[order][0:n:2], it[order][1:n:2]).mean()
        else:
            pred = float("nan")
        rows.append((single, nb, pred, ideal, [float(np.maximum(it[:, l], 0).mean()) for l in range(4)],
                     [float(np.maximum(np.maximum(it[0:n:2, l], it[1:n:2, l]), 0).mean()) for l in range(4)]))
        prev_it, p = it, p1
    print("vector instructions per keypoint (model: %g + %g per iteration, %g + %g per level)" % (IT_PIX, IT_UNI, LV_PIX, LV_UNI))
    print("%-8s %10s %14s %18s %14s" % ("frame", "1 kp/wave", "neighbours", "sorted by t-1", "ideal pairs"))
    for t, r in enumerate(rows):
        print("%-8d %10.0f %8.0f (%+.1f%%) %10.0f (%+.1f%%) %8.0f (%+.1f%%)   mean iterations per level %s, of a neighbour pair's maximum %s"
              % (t, r[0], r[1], 100 * (r[1] / r[0] - 1), r[2], 100 * (r[2] / r[0] - 1), r[3], 100 * (r[3] / r[0] - 1),
                 ["%.2f" % x for x in r[4]], ["%.2f" % x for x in r[5]]))


if __name__ == "__main__":
    main()
