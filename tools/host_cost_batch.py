import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "visual-odom-pipeline_amd"))
import numpy as np
import bench
from vo_mi355x import synthetic as syn
fs = bench.render_sequences([1234 + k for k in range(4)], 100)
g = bench.Group(0, fs, seed0=0, batch=256, ba_iters=30)
g.c.set_side_stream("pipeline")
for _ in range(10): g.step()
g.drain(); g.c.sync()
# pure unpack: nothing in flight -> frame_fetch returns the last results again
t0 = time.perf_counter()
for _ in range(10): g.c.frame_fetch()
print("frame_fetch with nothing in flight (pure unpack into fresh numpy arrays): %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
te = tf = 0.0
n = 60
t00 = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter(); g.enqueue(); te += time.perf_counter() - t0
    if g.inflight == 2:
        t0 = time.perf_counter(); g.fetch(); tf += time.perf_counter() - t0
g.drain(); g.c.sync()
tot = time.perf_counter() - t00
print("per step: total %.3f ms, enqueue %.3f ms, fetch (wait + unpack + bookkeeping) %.3f ms" % (tot / n * 1e3, te / n * 1e3, tf / n * 1e3))
