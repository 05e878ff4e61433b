"""Soak of the host-frame paths against the resident ones: the closed loop (vo_pipe_step_host, three steps in flight) and the fused step (vo_frame_step_host,
two in flight, gated layout) over many frames of a looped scene, records / results bit-identical.   python tools/host_frames_soak.py [frames] [batch]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))
import numpy as np  # noqa: E402
from vo_mi355x import VoContext, synthetic as syn  # noqa: E402
from vo_mi355x.resident import ResidentPipeline  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h, t1, nf = 480, 200, 4, 40
scs = [syn.sway_scene(nf, w=w, h=h, f=360.0, seed=500 + k, pose_fn=lambda t: syn.sway_pose(t, period=float(nf))) for k in range(2)]
with VoContext(w, h, max_pts=2048) as boot:
    states = [syn.gt_bootstrap(boot, sc, 0, t1)[0] for sc in scs]
frames = np.stack([scs[b % 2]["frames"] for b in range(B)])
store = VoContext.host_alloc((2, nf, h, w))
for k in range(2):
    store[k] = scs[k]["frames"]


def closed_loop(host):
    with VoContext(w, h, max_pts=1024, batch=B) as c:
        rp = ResidentPipeline(c, np.stack([scs[b % 2]["K"] for b in range(B)]), ba_window=4, ba_max_iters=10, pnp_blind_batches=2)
        rp.seed([copy.deepcopy(states[b % 2]) for b in range(B)], None, None, 1)
        if host:
            c.push_frame(frames[:, t1])
        else:
            c.upload_sequence(frames); c.push_frame_resident(t1)
        out, f, inflight = [], t1 + 1, 0
        for _ in range(n_frames):
            if host:
                rp.step_host([store[b % 2, f % nf] for b in range(B)])
            else:
                rp.step(f % nf)
            f += 1; inflight += 1
            if inflight == 3:
                out.append(rp.fetch()); inflight -= 1
        while inflight:
            out.append(rp.fetch()); inflight -= 1
        return out


a, b = closed_loop(False), closed_loop(True)
bad = 0
for s, (ra, rb) in enumerate(zip(a, b)):
    for x, y in zip(ra, rb):
        for k, v in x.items():
            same = np.array_equal(v, y[k]) if isinstance(v, np.ndarray) else (v == y[k] or (isinstance(v, float) and v != v and y[k] != y[k]))
            if not same:
                bad += 1
                if bad < 5:
                    print("closed loop differs: step", s, k, v, y[k])
alive = sum(1 for r in a[-1] if r["status"] == 0)
print("closed loop: %d frames x %d sequences, host = resident: %s (%d sequences alive at the end)" % (n_frames, B, bad == 0, alive))

pts = np.stack([syn.grid_points(600, w, h, seed=b, margin=12) for b in range(B)])
scene = [syn.make_ba_scene(n_pts=400, n_slots=10, seed=b) for b in range(B)]


def fused(host):
    with VoContext(w, h, max_pts=1024, batch=B) as c:
        c.set_side_stream("pipeline")
        c.points_upload(pts)
        c.ba_upload(np.stack([s["K"] for s in scene]), np.stack([s["poses0"] for s in scene]), np.stack([s["points0"] for s in scene]), np.stack([s["obs"] for s in scene]))
        if host:
            c.push_frame(frames[:, 0])
        else:
            c.upload_sequence(frames); c.push_frame_resident(0)
        out, inflight = [], 0
        bap = c.ba_params(max_iters=12)
        for t in range(1, n_frames + 1):
            if host:
                c.frame_step_host([store[b % 2, t % nf] for b in range(B)], 600, do_dlt=False, ba=bap)
            else:
                c.frame_step_resident(t % nf, 600, do_dlt=False, ba=bap)
            inflight += 1
            if inflight == 2:
                r = c.frame_fetch(); inflight -= 1
                out.append((r["points2d"].copy(), r["status"].copy(), [x["cost"] for x in r["ba_stats"]], [len(x) for x in r["corners"]]))
        return out, c.step_layout() if False else None


fa, fb = fused(False)[0], fused(True)[0]
ok = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and x[3] == y[3] for x, y in zip(fa, fb))
print("fused step (gated layout): %d frames x %d sequences, host = resident: %s" % (len(fa), B, ok))
sys.exit(0 if (bad == 0 and ok) else 1)
