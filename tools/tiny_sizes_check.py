"""GPU: the pyramid kernels' interior / border index split (vo_split_index, csrc/vo_frame.hip) at sizes where a padded row has few or no interior
groups -- small windows so that small images keep several levels.  Prints one line per size; tests/test_gpu_sizes.py runs the same check."""
import sys
import numpy as np
sys.path.insert(0, "oracle"); sys.path.insert(0, "visual-odom-pipeline_amd")
import vo_oracle as o
from vo_mi355x import VoContext

SIZES = [(34, 34, 5), (47, 33, 7), (36, 90, 5), (33, 200, 9), (64, 35, 5), (17, 17, 3), (16, 40, 3), (15, 15, 3), (23, 9, 3), (130, 12, 5), (12, 130, 5)]


def check(w, h, win, seed=5):
    rng = np.random.default_rng(seed + w * 31 + h)
    f0 = rng.integers(0, 256, (h, w), dtype=np.uint8); f1 = np.roll(f0, 1, axis=1)
    with VoContext(w, h, max_pts=64, win=win) as c:
        c.push_frame(f0); c.push_frame(f1)
        lv = o.build_pyramid(f1, win=win)
        ok = True
        for l in range(len(lv)):
            img_l, der_l = c.pyramid_read(1, l)
            ok &= np.array_equal(img_l, lv[l]) and np.array_equal(der_l, o.scharr(lv[l]))
        pts = np.array([[w / 2, h / 2], [3.5, 2.25], [w - 2.5, h - 3.0], [0.5, h - 1.0]], np.float32)
        p1, st, err, it = c.klt_track(pts, params=c.klt_params(win=win), return_iters=True)
        q1, qs, qe, qi = o.klt(f0, f1, pts, winSize=(win, win), return_iters=True)
        return len(lv), ok, bool(np.array_equal(p1, q1) and np.array_equal(st, qs) and np.array_equal(err, qe) and np.array_equal(it, qi))


if __name__ == "__main__":
    for (w, h, win) in SIZES:
        try:
            print((w, h, win), "levels %d pyramid %s klt %s" % check(w, h, win))
        except Exception as e:      # noqa: BLE001
            print((w, h, win), "error:", str(e)[:120])
