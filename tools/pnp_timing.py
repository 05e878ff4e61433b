"""Timing of the batched PnP-RANSAC call (host-synchronous API: includes the H2D of the correspondences and two syncs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))
import numpy as np
from vo_mi355x import VoContext, synthetic as syn
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 2000
rng = np.random.default_rng(0)
Ks, Xs, uvs = [], [], []
for b in range(B):
    s = syn.make_ba_scene(n_pts=n, n_slots=2, seed=b, obs_noise=0.3)
    X = s["points_gt"].astype(np.float32); uv = s["obs"][0].astype(np.float32)
    out = rng.choice(n, int(0.3 * n), replace=False)
    uv[out] += rng.uniform(-80, 80, (len(out), 2)).astype(np.float32)
    Ks.append(s["K"]); Xs.append(X); uvs.append(uv)
with VoContext(64, 64, max_pts=n, batch=B) as c:
    args = (np.stack(Ks), np.stack(Xs), np.stack(uvs))
    for _ in range(3):
        r = c.pnp_ransac(*args)
    t0 = time.perf_counter()
    for _ in range(20):
        r = c.pnp_ransac(*args)
    dt = (time.perf_counter() - t0) / 20
st = r[3] if isinstance(r[3], list) else [r[3]]
print("B=%d n=%d: %.3f ms per call (%.1f us per sequence); inliers %s hypotheses %s" % (B, n, dt * 1e3, dt * 1e6 / B, st[0]["n_inliers"], st[0]["hypotheses"]))
