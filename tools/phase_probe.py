"""Diagnostic: per-phase shader cycles of the single-workgroup kernels (vo_debug_cycles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))
import numpy as np
from vo_mi355x import VoContext, synthetic as syn
frames, _ = syn.make_sequence(2)
with VoContext(1241, 376, max_pts=2048) as c:
    c.push_frame(frames[0]); c.push_frame(frames[1])
    pts = syn.grid_points(2000, 1241, 376)
    for _ in range(3):
        p1, st, err, it = c.klt_track(pts, return_iters=True)
    print("klt one wave: iters", it[1000], "cycles [total, template, iter0, top level, levels..1, level0+err]", c.debug_cycles(3)[:6])
    for _ in range(3):
        cr = c.shi_tomasi(pts, 7)
    eig, mask, nc = c.shi_tomasi_read()
    print("st_select: corners", len(cr), "candidates", nc, "cycles [total, sort, grid, rounds, compact]", c.debug_cycles(0)[:5], "rounds", c.debug_cycles(0)[7], "round0 cycles", c.debug_cycles(0)[5])
    s = syn.make_ba_scene(2000, 10, seed=0)
    c.ba_upload(s["K"], s["poses0"], s["points0"], s["obs"])
    for _ in range(3):
        pr = c.ba_probe(1e-4)
    print("ba_solve cycles [total, reduce, assemble, chol, backsub, publish]", c.debug_cycles(1)[:6])
    print("ba_build cycles [total, lin, sums+3x3, camsums, panel, mfma]", c.debug_cycles(2)[:6])
