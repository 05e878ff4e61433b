"""Timing of the bootstrap calls (host-synchronous API: includes the H2D of the inputs and the syncs):
descriptor matching 1000 x 1000 x 128 and the five-point RANSAC + recoverPose on 1000 matches with 30 % outliers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from vo_mi355x import VoContext
from test_gpu_essential import two_view_scene

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(0)
d1 = np.floor(rng.gamma(0.6, 30.0, (B, 1000, 128))).astype(np.float32)
d2 = np.floor(rng.gamma(0.6, 30.0, (B, 1000, 128))).astype(np.float32)
scenes = [two_view_scene(1000, 0.3, 10 + b) for b in range(B)]
K = np.stack([s[0] for s in scenes]); p1 = np.stack([s[1] for s in scenes]); p2 = np.stack([s[2] for s in scenes])
with VoContext(64, 64, max_pts=1000, batch=B) as c:
    for name, fn in (("match_knn2 1000x1000x128", lambda: c.match_knn2(d1, d2)),
                     ("essential_ransac n=1000, 30% outliers", lambda: c.essential_ransac(K, p1, p2, seed=1))):
        for _ in range(3):
            r = fn()
        t0 = time.perf_counter()
        for _ in range(20):
            r = fn()
        dt = (time.perf_counter() - t0) / 20
        extra = ""
        if name.startswith("essential"):
            st = r[4] if isinstance(r[4], list) else [r[4]]
            extra = "; inliers %d, samples %d" % (st[0]["n_inliers"], st[0]["hypotheses"])
        print("B=%d %s: %.3f ms per call (%.3f ms per sequence)%s" % (B, name, dt * 1e3, dt * 1e3 / B, extra))
