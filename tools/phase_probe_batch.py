"""Diagnostic: phase stamps of k_ba_build workgroup 0 / k_ba_solve while a 32-problem batch is in flight."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))
import numpy as np
from vo_mi355x import VoContext, synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = [syn.make_ba_scene(2000, 10, seed=b) for b in range(B)]
with VoContext(64, 64, max_pts=64, batch=B) as c:
    c.ba_upload(np.stack([s["K"] for s in sc]), np.stack([s["poses0"] for s in sc]), np.stack([s["points0"] for s in sc]),
                np.stack([s["obs"] for s in sc]))
    for _ in range(3):
        pr = c.ba_probe(1e-4)      # one iteration of all B problems, stamps from problem 0
    print("B =", B)
    print("ba_solve cycles [total, reduce, assemble, chol, backsub, publish]", c.debug_cycles(1)[:6])
    print("ba_build cycles [total, lin, sums+3x3, camsums, panel, mfma, (w: total, to 3rd chunk, pass1, pass2, gram, rest of walk, epilogue)]", c.debug_cycles(2)[:7])
    c.ba_solve_resident(c.ba_params(max_iters=10))
    po, pt, st = c.ba_fetch()
    st = st if isinstance(st, list) else [st]
    print("iters", [x["iters"] for x in st])
