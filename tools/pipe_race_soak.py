"""Soak for the closed loop's stream layout: the same 8-sequence run (full image size, 4 steps in flight, table filling up to its capacity)
repeated N times with the side stream and once on one stream -- every record and every table must be bit-identical every time.  A read of
something the other stream has not finished writing would show up as a run that differs.   python tools/pipe_race_soak.py [repeats]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "visual-odom-pipeline_amd")]
import numpy as np
from vo_mi355x import VoContext, synthetic as syn
from vo_mi355x.resident import ResidentPipeline

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
w, h, t1, n, B, depth = 1241, 376, 4, 24, 8, 4
scs = [syn.sway_scene(t1 + n + 1, w=w, h=h, f=718.856, seed=sd, pose_fn=lambda t: syn.sway_pose(t, period=40.0)) for sd in (99, 4321)]
boot = VoContext(w, h, max_pts=4096)
states = [syn.gt_bootstrap(boot, sc, 0, t1)[0] for sc in scs]
boot.close()


def run(side, window, resurrect):
    c = VoContext(w, h, max_pts=2048, batch=B)
    c.set_side_stream(side)
    c.upload_sequence(np.stack([scs[b % 2]["frames"] for b in range(B)]))
    rp = ResidentPipeline(c, np.stack([scs[b % 2]["K"] for b in range(B)]), ba_window=window, ba_max_iters=10, resurrect=resurrect)
    rp.seed([copy.deepcopy(states[b % 2]) for b in range(B)], None, None, 1)
    c.push_frame_resident(t1)
    recs = []
    for s0 in range(0, n, depth):
        for s in range(s0, s0 + depth):
            rp.step(t1 + 1 + s)
        for s in range(s0, s0 + depth):
            recs.append(rp.fetch())
    T = rp.read_tables()
    c.close()
    return recs, T


def same(a, b):
    (ra, Ta), (rb, Tb) = a, b
    for s in range(n):
        for q in range(B):
            for k in ra[s][q]:
                if not np.array_equal(np.asarray(ra[s][q][k]), np.asarray(rb[s][q][k])):
                    return "record step %d seq %d field %s: %r vs %r" % (s, q, k, ra[s][q][k], rb[s][q][k])
    for name in Ta:
        if not np.array_equal(Ta[name], Tb[name], equal_nan=Ta[name].dtype.kind == "f"):
            return "table " + name
    return None


bad = 0
for window, resurrect in ((4, True), (10, False)):
    ref = run(False, window, resurrect)
    overflow = sum(1 for r in ref[0] for q in r if q["overflow"])
    for i in range(reps):
        d = same(ref, run(True, window, resurrect))
        if d:
            bad += 1
            print("window %d run %d DIFFERS: %s" % (window, i, d))
    print("window %d: %d side-stream runs against the one-stream run, %d frames with the capacity policy active, statuses %s"
          % (window, reps, overflow, sorted({q["status"] for r in ref[0] for q in r})))
print("differences:", bad)
sys.exit(1 if bad else 0)
