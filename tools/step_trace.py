"""debug: GPU-side timeline of the bundle-adjustment stream of a pipelined single-sequence run (VO_STEP_TRACE=1)"""
import os, sys
os.environ["VO_STEP_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["bench.py"]
import importlib.util as u
sp = u.spec_from_file_location("b", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")); b = u.module_from_spec(sp); sp.loader.exec_module(b)
from vo_mi355x import synthetic as syn, _lib
frame_sets = [syn.make_sequence(16, b.W_IMG, b.H_IMG, seed=1234)[0]]
g = b.Group(0, frame_sets, seed0=7000, batch=1, ba_iters=10)
g.c.set_side_stream(2)
for _ in range(300): g.step()
g.drain(); g.c.sync()
import ctypes
L = ctypes.CDLL(_lib.LIB_PATH)
L.vo_debug_step_trace_dump(400, 12)
