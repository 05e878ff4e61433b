"""Per-call time of the drop-in classes (Python objects in / out, synchronous, host buffers) on one 1241x376 frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))
import numpy as np
from vo_mi355x import Extractor, Keypoint, VoContext, synthetic as syn

w, h = 1241, 376
frames, _ = syn.make_sequence(4, w=w, h=h, seed=3)
pts = syn.grid_points(2000, w, h, seed=1)


def kps(p):
    return [Keypoint(0, 1, q.reshape(2, 1).copy(), q.reshape(2, 1).copy(), np.zeros((1, 1)), [q.reshape(2, 1).copy()]) for q in p]


with VoContext(w, h, max_pts=4096) as c:
    ext = Extractor(min_kp_dist=7, ctx=c)
    ext._im_prev = frames[0]
    cand = kps(pts)
    for warm in range(2):
        ext.extend_tracks(frames[1], kps(pts), np.inf)
    t = {}
    for name, fn in (("extend_tracks (2000 keypoints: KLT x2 as the reference calls it + Python bookkeeping)", lambda: ext.extend_tracks(frames[1], cand, np.inf)),
                     ("extract shi-tomasi (2000 exclusion discs)", lambda: ext.extract(frames[1], 1, cand, 'shi-tomasi', 7, False))):
        t0 = time.perf_counter()
        for _ in range(5):
            out = fn()
        t[name] = (time.perf_counter() - t0) / 5
    # array API for comparison
    c.push_frame(frames[0]); c.push_frame(frames[1])
    t0 = time.perf_counter()
    for _ in range(20):
        c.klt_track(pts)
    t["VoContext.klt_track (array API, 2000 points, host in/out)"] = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        c.shi_tomasi(pts, 7)
    t["VoContext.shi_tomasi (array API)"] = (time.perf_counter() - t0) / 20
    s = syn.make_ba_scene(2000, 10, seed=0)
    c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"])
    t0 = time.perf_counter()
    for _ in range(10):
        c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"])
    t["VoContext.ba_adjust (2000 landmarks x 10 poses, host in/out)"] = (time.perf_counter() - t0) / 10
for k, v in t.items():
    print("%-95s %8.3f ms" % (k, v * 1e3))

# BundleAdjuster.adjust through the drop-in class: State with 2000 landmarks observed over a 10-frame window
from vo_mi355x import BundleAdjuster, Landmark, State, Trajectory
W, N = 10, 2000
s = syn.make_ba_scene(N, W, seed=0)
T = 12
def make_state():
    traj = Trajectory({})
    for t in range(T):
        slot = T - 1 - t
        H = np.eye(4)
        if slot < W:
            H[:3, :3] = syn.rodrigues(s["poses0"][slot, :3]); H[:3, 3] = s["poses0"][slot, 3:]
        traj.append(t, H)
    lms, ks = [], []
    for j in range(N):
        hist = [s["obs"][W - 1 - i, j].reshape(2, 1).copy() for i in range(W)]        # oldest first, newest = slot 0 last
        ks.append(Keypoint(T - W, W, hist[0].copy(), hist[-1].copy(), np.zeros((1, 1)), hist))
        lms.append(Landmark(T - 1, s["points0"][j].reshape(3, 1).copy(), np.zeros((1, 1))))
    return State(lms, ks, [], traj)
with VoContext(64, 64, max_pts=64) as c:
    ba = BundleAdjuster(verbosity=0, window_size=W, method='trf', xtol=1e-3, ftol=1e-3, ctx=c)
    ba.adjust(make_state(), [], [], s["K"], T - 1)
    ts = []
    for _ in range(5):
        st = make_state()
        t0 = time.perf_counter()
        ba.adjust(st, [], [], s["K"], T - 1)
        ts.append(time.perf_counter() - t0)
    print("%-95s %8.3f ms  (cost %.1f -> %.1f)" % ("BundleAdjuster.adjust (drop-in class, 2000 landmarks x 10 frames of Python objects)", min(ts) * 1e3,
                                                ba.last_stats["cost0"], ba.last_stats["cost"]))
