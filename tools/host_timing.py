import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["bench.py"]
import importlib.util as u
sp = u.spec_from_file_location("b", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")); b = u.module_from_spec(sp); sp.loader.exec_module(b)
import numpy as np
from vo_mi355x import synthetic as syn
frame_sets = [syn.make_sequence(16, b.W_IMG, b.H_IMG, seed=1234)[0]]
for mode in (1, 2):
    g = b.Group(0, frame_sets, seed0=7000, batch=1, ba_iters=10)
    g.c.set_side_stream(mode)
    for _ in range(30): g.step()
    g.drain(); g.c.sync()
    te = tf = 0.0; n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); g.enqueue(); bb = time.perf_counter()
        if g.inflight == g.max_inflight: g.fetch()
        c = time.perf_counter(); te += bb - a; tf += c - bb
    g.drain(); g.c.sync()
    tot = time.perf_counter() - t0
    print("mode", mode, "period us %.1f  enqueue us %.1f  fetch(wait) us %.1f" % (tot / n * 1e6, te / n * 1e6, tf / n * 1e6))
    g.c.close()
