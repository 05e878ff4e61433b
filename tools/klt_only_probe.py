"""probe: does an IDLE context whose front-end stream carries a CU mask (pipelined layout of a batch) slow another context's launches?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.argv = ["bench.py"]
import importlib.util as u
sp = u.spec_from_file_location("b", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")); b = u.module_from_spec(sp); sp.loader.exec_module(b)
from vo_mi355x import synthetic as syn, VoContext
frame_sets = [syn.make_sequence(16, b.W_IMG, b.H_IMG, seed=1234)[0]]
def run(g, n, warm=20):
    for _ in range(warm): g.step()
    g.drain(); g.c.sync()
    t = time.perf_counter()
    for _ in range(n): g.step()
    g.drain(); g.c.sync()
    return (time.perf_counter() - t) / n * 1e3
g = b.Group(0, frame_sets, seed0=7000, batch=1, ba_iters=10)
g.c.set_side_stream("pipeline")
print("one sequence, pipelined layout, alone on the device:      %.4f ms / frame" % run(g, 300))
g.stages = (False, False, False); print("  KLT only:                                            %.4f" % run(g, 300)); g.stages = (True, True, True)
other = VoContext(b.W_IMG, b.H_IMG, max_pts=2048, batch=8)
other.set_side_stream("pipeline")
print("idle second context:", other.step_layout())
print("one sequence beside an idle context with a CU-masked stream: %.4f ms / frame" % run(g, 300))
g.stages = (False, False, False); print("  KLT only:                                            %.4f" % run(g, 300)); g.stages = (True, True, True)
other.set_side_stream(True)
print("idle second context:", other.step_layout())
print("one sequence beside an idle context, mask released:        %.4f ms / frame" % run(g, 300))
g.stages = (False, False, False); print("  KLT only:                                            %.4f" % run(g, 300))
