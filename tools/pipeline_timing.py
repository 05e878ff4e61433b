"""Wall time of the drop-in Pipeline (bootstrap + per-frame step, everything numerical on the GPU, Python bookkeeping as the
reference prescribes it) on a rendered 1241 x 376 two-plane sequence."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))
import numpy as np
from vo_mi355x import Pipeline, synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
frames, K, poses = syn.make_two_plane_sequence(n, w=1241, h=376, f=718.856, seed=7)


class MemLoader:                       # the Loader interface over frames in memory (already "filtered")
    _name = "synthetic"
    def __len__(self): return len(frames)
    def getCamera(self): return K
    def getInit(self): return (0, 4)
    def getImage(self, i): return frames[i]
    def getFrame(self, i): return frames[i], np.linalg.inv(poses[i])


t0 = time.perf_counter()
pipe = Pipeline(MemLoader(), headless=True)
t_boot = time.perf_counter() - t0
ts, sizes = [], []
for _ in range(n - 5):
    t0 = time.perf_counter()
    pipe.step()
    ts.append(time.perf_counter() - t0)
    sizes.append((len(pipe._state._landmarks), len(pipe._state._candidates_kp)))
ts = np.array(ts) * 1e3
unit = np.linalg.norm(poses[4][:3, 3])
err = max(np.linalg.norm(pipe._state._trajectory[k][:3, 3] - poses[4 + k - 1][:3, 3] / unit) for k in range(1, pipe._t_step + 1))
print("bootstrap %.1f ms; step: median %.2f ms, mean %.2f ms, max %.2f ms (%d steps; landmarks %d..%d, candidates %d..%d); "
      "max translation error %.3f baselines" % (t_boot * 1e3, np.median(ts), ts.mean(), ts.max(), len(ts), min(s[0] for s in sizes),
                                                max(s[0] for s in sizes), min(s[1] for s in sizes), max(s[1] for s in sizes), err))
