"""debug: GPU-side timeline of the pipelined frame step of a BATCH (VO_STEP_TRACE=1): per step, the bundle adjustment's span on stream C and
the next frame's front end on stream A relative to it.  usage: python tools/step_trace_batch.py [sequences] [ba_iters]"""
import os, sys
os.environ["VO_STEP_TRACE"] = "1"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["bench.py"]
import importlib.util as u
sp = u.spec_from_file_location("b", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")); b = u.module_from_spec(sp); sp.loader.exec_module(b)
from vo_mi355x import synthetic as syn, _lib
frame_sets = [syn.make_sequence(16, b.W_IMG, b.H_IMG, seed=1234 + i)[0] for i in range(4)]
g = b.Group(0, frame_sets, seed0=7000, batch=B, ba_iters=IT)
g.c.set_side_stream(2)
print(g.c.step_layout(), file=sys.stderr)
for _ in range(60): g.step()
g.drain(); g.c.sync()
import ctypes
L = ctypes.CDLL(_lib.LIB_PATH)
L.vo_debug_step_trace_dump(40, 8)
