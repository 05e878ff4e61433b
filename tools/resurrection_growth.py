"""How the reference's dead-landmark resurrection (bundle_adjuster.py:142-150) fills the lists as a function of the BA window.
  python tools/resurrection_growth.py [frames] [max_pts]
ONE sequence of bench.py's closed-loop scene through the device tables (ResidentPipeline) at windows 4 and 10 with resurrection on, and at
window 10 with it off: per frame the list sizes, what the frame resurrected, and the frames in which the capacity policy cut something."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("visual-odom-pipeline_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np

import bench
from vo_mi355x import VoContext, synthetic as syn
from vo_mi355x.resident import ResidentPipeline

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
max_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
sc = bench.pipe_scenes(1, 40, 4321)[0]
nf = len(sc["frames"])
off = bench.pipe_phase_offsets(sc, 1)[-1]
roll = dict(frames=np.roll(sc["frames"], -off, axis=0), poses=np.roll(sc["poses"], -off, axis=0), K=sc["K"], f=sc["f"],
            surface=lambda t, xy: sc["surface"]((t + off) % nf, xy))
T1 = bench.PIPE_T1
for window, resurrect in ((4, True), (10, True), (10, False)):
    with VoContext(bench.W_IMG, bench.H_IMG, max_pts=max_pts) as boot, VoContext(bench.W_IMG, bench.H_IMG, max_pts=max_pts) as c:
        state, _ = syn.gt_bootstrap(boot, roll, 0, T1)
        rp = ResidentPipeline(c, sc["K"], ba_window=window, ba_max_iters=10, ba_budget=10, pnp_blind_batches=2, resurrect=resurrect)
        rp.seed(copy.deepcopy(state), None, None, t_step=1)
        c.push_frame(roll["frames"][T1])
        print("window %d, resurrection %s, %d-slot tables (landmark entries + candidates <= slots, dead entries <= slots)" % (window, "on" if resurrect else "off", max_pts))
        print("frame landmark_entries candidates dead_entries resurrected new detected ba_observations overflow_bits status")
        for k in range(frames):
            c.push_frame(roll["frames"][(T1 + 1 + k) % nf]); rp.step(); r = rp.fetch()
            print("%4d %9d %10d %10d %10d %6d %7d %10d %8d %6d" % (r["t"], r["n_landmarks"], r["n_candidates"], r["n_dead"], r["n_resurrected"], r["n_new"],
                                                                 r["n_detected"], r["ba_observations"], r["overflow"], r["status"]), flush=True)
            if r["status"]:
                print("  (the sequence stopped: status %d)" % r["status"])
                break
        print()
