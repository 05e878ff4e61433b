"""Where one sequence's closed-loop frame period goes on the HOST: time inside vo_pipe_step (enqueue) and inside vo_pipe_fetch (wait + copy)
per frame, with 3 steps in flight like bench.py --workload pipeline --seqs 1.   python tools/pipe_host_probe.py [seqs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "visual-odom-pipeline_amd"), ROOT]
import numpy as np
import bench

seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scenes = bench.pipe_scenes(max(1, min(seqs, 4)), 40, 4321)
from vo_mi355x import VoContext
boot = VoContext(bench.W_IMG, bench.H_IMG, max_pts=4096)
g = bench.PipeGroup(0, scenes, boot, 0, seqs, 10, 2048, True, 4, True)
for _ in range(30):
    g.step()
g.drain()
n = 300
t_enq = t_fet = 0.0
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); g.enqueue(); b = time.perf_counter(); t_enq += b - a
    if g.inflight == g.max_inflight:
        a = time.perf_counter(); g.fetch(); b = time.perf_counter(); t_fet += b - a
g.drain()
tot = time.perf_counter() - t0
print("%d sequence(s): %.1f us per frame; inside enqueue %.1f us, inside fetch %.1f us, rest (python loop) %.1f us"
      % (seqs, tot / n * 1e6, t_enq / n * 1e6, t_fet / n * 1e6, (tot - t_enq - t_fet) / n * 1e6))
