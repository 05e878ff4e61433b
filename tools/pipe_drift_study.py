"""Which sequence of the closed-loop bench drifts, and does the table model over the CPU oracle drift with it?  (round-4 review item 5:
`window4_tables_not_full_8192_slots` reports rotation_deg_max 6.3 deg.)
  python tools/pipe_drift_study.py [first_index] [frames] [max_pts] [window]
Runs sequence `first_index` of bench.py's closed-loop scenes (same scene, phase offset and ground-truth bootstrap as PipeGroup) through
(a) the device tables (ResidentPipeline, batch 1) and (b) oracle/pipe_oracle.PipeModel over the CPU oracle context, frame by frame, and prints
both pose errors against the rendered ground truth, the list sizes and whether the two trajectories agree."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("visual-odom-pipeline_amd", "oracle", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np

import bench
import pipe_oracle as po
from oracle_context_impl import OracleContext
from vo_mi355x import VoContext, synthetic as syn
from vo_mi355x.resident import ResidentPipeline

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
max_pts = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
window = int(sys.argv[4]) if len(sys.argv) > 4 else 4
with_model = os.environ.get("DRIFT_MODEL", "1") != "0"
scenes = bench.pipe_scenes(2, 40, 4321)
sc = scenes[first % len(scenes)]
nf = len(sc["frames"])
off = bench.pipe_phase_offsets(sc, first + 1)[-1]
roll = dict(frames=np.roll(sc["frames"], -off, axis=0), poses=np.roll(sc["poses"], -off, axis=0), K=sc["K"], f=sc["f"],
            surface=lambda t, xy: sc["surface"]((t + off) % nf, xy))
T1 = bench.PIPE_T1
G0 = roll["poses"][0]
unit = np.linalg.norm((roll["poses"][T1] @ np.linalg.inv(G0))[:3, 3])


def err(H, t):
    gt = roll["poses"][(T1 + t - 1) % nf] @ np.linalg.inv(G0)
    c = (np.trace(H[:3, :3] @ gt[:3, :3].T) - 1) / 2
    return float(np.degrees(np.arccos(np.clip(c, -1, 1)))), float(np.linalg.norm(H[:3, 3] - gt[:3, 3] / unit))


with VoContext(bench.W_IMG, bench.H_IMG, max_pts=max_pts) as boot, VoContext(bench.W_IMG, bench.H_IMG, max_pts=max_pts) as cb:
    state, _ = syn.gt_bootstrap(boot, roll, 0, T1)
    rp = ResidentPipeline(cb, sc["K"], ba_window=window, ba_max_iters=10, ba_budget=10, pnp_blind_batches=8)
    rp.seed(copy.deepcopy(state), None, None, t_step=1)
    cb.push_frame(roll["frames"][T1])
    model = None
    if with_model:
        oc = OracleContext(bench.W_IMG, bench.H_IMG, max_pts=max_pts)
        model = po.PipeModel(oc, sc["K"], bench.W_IMG, bench.H_IMG, cap=max_pts, params=po.Params(ba_window=window, ba_max_iters=10))
        model.seed(copy.deepcopy(state), [], [], 1)
        oc.push_frame(roll["frames"][T1])
    print("sequence %d: scene %d, phase offset %d, %d-slot tables, window %d" % (first, first % len(scenes), off, max_pts, window))
    print("frame  device: rot_deg  trans   landmarks cand  dead  ba_it |  model: rot_deg  trans  landmarks | max |H_dev - H_model|")
    for k in range(frames):
        im = roll["frames"][(T1 + 1 + k) % nf]
        cb.push_frame(im); rp.step(); rec = rp.fetch()
        e = err(np.vstack([np.array(rec["H"]).reshape(3, 4), [0, 0, 0, 1]]) if np.size(rec["H"]) == 12 else np.array(rec["H"]), rec["t"])
        line = "%4d   %7.3f %7.3f   %6d %5d %5d %5d" % (rec["t"], e[0], e[1], rec["n_landmarks"], rec["n_candidates"], rec["n_dead_total"], rec["ba_iters"])
        if model is not None:
            model.step(im)
            Hm = model.poses[model.t]
            em = err(Hm, model.t)
            Hd = np.vstack([np.array(rec["H"]).reshape(3, 4), [0, 0, 0, 1]]) if np.size(rec["H"]) == 12 else np.array(rec["H"])
            line += " |  %7.3f %7.3f  %6d | %.2e" % (em[0], em[1], len(model.lm_L), float(np.abs(Hd - Hm).max()))
        print(line, flush=True)
