#!/usr/bin/env python3
"""bench.py -- frames/sec of the VO inner loop on MI355X (BASELINE.json metric).

Workload (config.workload = "kitti_shaped_1241x376_2000pts_ba10", BASELINE.json configs[2], the configuration the
metric string names): per frame and per sequence
    pyramid + Scharr of the new frame -> KLT of 2000 keypoints (31x31, 4 levels, <= 30 its, eps 0.03)
    -> DLT of 1000 new tracks (+ filter statistics) -> sliding-window BA (N = 2000 landmarks, W = 10 poses,
    Huber, LM with ftol = xtol = 1e-3, at most --ba-iters iterations) -> Shi-Tomasi re-detection (<= 1000 corners,
    2000 exclusion discs).
One "step" = one such frame for each of the --seqs independent sequences a GPU carries (sequences are the unit
that shards: frames of ONE sequence are sequential).  The sequences are carried by --ctxs BATCHED contexts: a context
advances its sequences in lockstep, every kernel launch serves the whole batch (one sequence alone cannot fill 256 CUs).  All inputs (frames, keypoints, BA problem) are resident in HBM
before the timed region; per step only the results come back (points, corners, landmarks, poses).
Multi-GPU: one process per GPU (torch.distributed launch contract), independent sequences per rank, no data-path
collective -> "scaling": "weak".  torch.distributed (gloo) is used ONLY for the barrier / max-over-ranks timing.

`--gpus N` (N > 1) without a torch.distributed environment starts the N ranks itself (a `python -m torch.distributed.run`
child process, before this process touches the GPU) and relays rank 0's line.

The timed region (exactly --steps steps between barrier + sync on both sides) is repeated --regions times; `value` /
`ms_per_step` are the MEDIAN region, `regions` carries min / median / max.  After the timed regions rank 0 of a 1-GPU run adds
informational measurements of the other BASELINE configurations (`single_sequence`, `klt_only`, `pipeline_step`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))

import numpy as np  # noqa: E402

W_IMG, H_IMG, N_PTS, N_NEW, BA_N, BA_W = 1241, 376, 2000, 1000, 2000, 10       # --workload A (the metric's configuration)
WORKLOAD = "kitti_shaped_1241x376_2000pts_ba10"
K_CAM = None                                                                # None: synthetic.KITTI_K
# (observation noise px, point noise m, pose noise m, visibility) of the BA problems a sequence cycles through, one per frame
BA_VARIANTS = [(0.3, 0.3, 0.02, 1.0), (0.3, 0.6, 0.05, 1.0), (0.5, 0.3, 0.02, 0.9), (0.3, 1.0, 0.10, 1.0), (0.3, 0.3, 0.02, 0.8),
               (0.8, 0.5, 0.05, 1.0), (0.3, 0.8, 0.02, 0.95), (0.4, 0.4, 0.08, 0.85)]
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--regions", type=int, default=5, help="the K-step timed region is repeated this many times; the median is reported")
    ap.add_argument("--extras", action="store_true", help="after the headline: the informational single-sequence / KLT-only / pipeline / drop-in measurements "
                    "(minutes; they go to bench_extras.json, never into the printed line)")
    ap.add_argument("--no-extras", action="store_true", help="the headline only: also skip the two second figures of the default line (`host_frames`, "
                    "`closed_loop_w10_256`); what the profiling scripts under tools/ run")
    ap.add_argument("--full-line", action="store_true", help="print the full result object (tens of KB) as the last line instead of the compact one "
                    "(tools/ and the child runs of --extras parse it); the driver's contract is the compact line")
    ap.add_argument("--extras-file", default=None, help="where the full result object goes (default: bench_extras.json beside bench.py, and gpurun_out/ when it exists)")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)     # launcher plumbing test: no GPU work
    ap.add_argument("--cpu-pipe-worker", type=int, default=-1, help=argparse.SUPPRESS)   # CPU-baseline worker of the closed loop (table model over the oracle)
    ap.add_argument("--cpu-pipe-frames", type=int, default=6, help="frames each CPU worker of the closed-loop baseline steps (after 2 untimed ones)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--seqs", type=int, default=256, help="independent sequences per GPU (round 4: 256 in ONE batched context; rounds 1-3 ran "
                    "96 in three contexts: --seqs 96 --ctxs 3 --side-stream on, also measured by the default run as `layout_3_contexts_of_32`)")
    ap.add_argument("--ctxs", type=int, default=None, help="batched contexts the sequences are split over; default 1 (the stream layout of a "
                    "context overlaps its own stages: DESIGN.md section 6)")
    ap.add_argument("--ba-iters", type=int, default=None, help="LM iteration cap per adjust.  Default 30 for workload A: every solve of the workload stops by the LM's own "
                    "ftol / xtol tests (the slowest in its 21st iteration); rounds 1-4 ran 10, where 1 %% of the solves were cut (`lm_cap_10` of the default line).  "
                    "Default 10 for config5, whose launch budget is fixed at the cap (its solves take 5 iterations)")
    ap.add_argument("--pipe-ba-iters", type=int, default=10, help="--workload pipeline: LM iteration cap per adjust (= the blind launch groups per frame unless "
                    "--pipe-adaptive-budget)")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic image sequences behind a GPU's batch (default: 8 if 8 cores can render them "
                    "side by side, else 4)")
    ap.add_argument("--frames", type=int, default=100, help="distinct synthetic frames per sequence: a closed loop of smooth motion, played round and round")
    ap.add_argument("--pipe-window", type=int, default=4, help="--workload pipeline: BA window (the reference's own setting is 4, pipeline.py:19; BASELINE's 10)")
    ap.add_argument("--pipe-no-resurrect", action="store_true", help="--workload pipeline: dead landmarks stay dead (the reference appends the recently dead "
                                                                     "to the state's lists again in every adjust, bundle_adjuster.py:142-150)")
    ap.add_argument("--pipe-adaptive-budget", action="store_true",
                    help="--workload pipeline: enqueue only (newest fetched frame's maximum + 2) LM iterations per frame instead of --ba-iters "
                         "(+5 %%; a solve the budget cuts cannot be continued in a closed loop -- the next frame already depends on it -- so the default "
                         "enqueues the LM's full --ba-iters, whose surplus groups exit early)")
    ap.add_argument("--pipe-max-pts", type=int, default=2048, help="--workload pipeline: capacity of the tracked keypoint set per sequence (<= 8192)")
    ap.add_argument("--pipe-host-frames", action="store_true", help="--workload pipeline: every step's images are handed over by the host (page-locked numpy arrays, "
                    "one per sequence -> vo_pipe_step_host) instead of read from the sequence store in HBM")
    ap.add_argument("--pipe-second-max-pts", type=int, default=0, help="--workload pipeline: after the run, the same run once more with this table capacity (same scenes, "
                    "one process: what the default line's two closed-loop figures share); its result object goes under `second`")
    ap.add_argument("--pipe-second-warmup", type=int, default=60)
    ap.add_argument("--pipe-frames", type=int, default=40, help="--workload pipeline: rendered frames per scene (= the period of the camera's sway; played in a loop)")
    ap.add_argument("--graph", action="store_true", help="replay each frame from a captured hipGraph instead of plain launches")
    ap.add_argument("--host-threads", type=int, default=3, help="enqueue/fetch the contexts from this many host threads")
    ap.add_argument("--fixed-ba-budget", action="store_true",
                    help="always enqueue --ba-iters LM iterations (default: what the last fetched frame needed, + 2 after a frame that hit its budget)")
    ap.add_argument("--side-stream", choices=("on", "off", "pipeline"), default=None,
                    help="Shi-Tomasi + DLT of a step on a side stream beside the BA (on: +1-2 %% with three contexts, +10-20 %% with one); "
                         "pipeline: also the BA of frame t beside the front end of frame t + 1 (three streams).  Default: pipeline with one "
                         "context, on with several")
    ap.add_argument("--workload", choices=("A", "config5", "pipeline"), default="A",
                    help="A: BASELINE configs[2], the metric's configuration (default).  config5: ONE 1920x1080 sequence, 5000 "
                         "points, 20-frame BA whose landmarks are sharded over the ranks with an RCCL all-reduce per LM "
                         "iteration (front end replicated); strong scaling, not the headline metric.  pipeline: the whole Pipeline.step "
                         "resident on the device as a closed loop (tracks, landmarks, dead lists and the trajectory in device tables; one "
                         "enqueue per frame, every stage fed by the previous ones); informational")
    ap.add_argument("--tune", default="", help="vo_tuning fields for every context of the run, e.g. ba_kernels=1,gate_groups=5 (include/vo_mi355x.h; A/B scripts under tools/)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames each CPU-baseline worker process runs")
    ap.add_argument("--cpu-procs", type=int, default=16, help="CPU-baseline worker processes (one core each), capped by the host's cores")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)     # internal: run as CPU-baseline worker with this seed
    a = ap.parse_args()
    if a.ba_iters is None:
        a.ba_iters = 10 if a.workload == "config5" else 30
    return a


class Dist:
    """barrier + max over ranks; torch.distributed(gloo) only when launched with WORLD_SIZE > 1"""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if "VO_BENCH_FORCE_DEVICE" in os.environ:      # plumbing tests: several ranks on one GPU
            self.local_rank = int(os.environ["VO_BENCH_FORCE_DEVICE"])
        self.td = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch
            import torch.distributed as td
            td.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.td, self.torch = td, torch

    def barrier(self):
        if self.td:
            self.td.barrier()

    def max(self, v):
        if not self.td:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t[0])

    def sum(self, v):
        if not self.td:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.SUM)
        return float(t[0])

    def bcast_bytes(self, arr):
        """rank 0's uint8 array to everybody (the RCCL unique id of the sharded-BA communicator)"""
        if not self.td:
            return arr
        t = self.torch.from_numpy(np.ascontiguousarray(arr, np.uint8).copy())
        self.td.broadcast(t, 0)
        return t.numpy()

    def close(self):
        if self.td:
            self.td.destroy_process_group()


def pingpong(t, n):
    """frame index of step t for a sequence of n frames played 0..n-1..0.."""
    period = 2 * (n - 1)
    k = t % period
    return k if k < n else period - k


def ba_bank(seed0, b, **kw):
    """the BA problems sequence b of a context (seeded seed0) cycles through, one per frame in turn: a sliding window never shows the same problem
    twice, so the LM iteration count changes from frame to frame and the iteration budget has to cope with that"""
    from vo_mi355x import synthetic as syn
    return [syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=seed0 + b, **kw)] + \
           [syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=seed0 + b + 7919 * k, obs_noise=v[0], pt_noise=v[1], pose_noise=v[2], visibility=v[3], **kw)
            for k, v in enumerate(BA_VARIANTS) if k > 0]


def dlt_inputs(scene):
    """-> (P0, P1 float32 3x4, uv0, uv1 float32 [N_NEW, 2], K, H0, H1): 1000 new tracks between two window poses of a BA scene"""
    from vo_mi355x import synthetic as syn
    K = scene["K"]
    H0, H1 = np.eye(4), np.eye(4)
    H0[:3, :3], H0[:3, 3] = syn.rodrigues(scene["poses_gt"][3, :3]), scene["poses_gt"][3, 3:]
    H1[:3, :3], H1[:3, 3] = syn.rodrigues(scene["poses_gt"][0, :3]), scene["poses_gt"][0, 3:]
    return ((K @ H0[:3]).astype(np.float32), (K @ H1[:3]).astype(np.float32), scene["obs"][3, :N_NEW].astype(np.float32),
            scene["obs"][0, :N_NEW].astype(np.float32), K, H0, H1)


class Group:
    """`batch` independent VO sequences carried in lockstep by ONE batched context (one HIP stream): every launch of
    the hot path serves all of them.  Everything is resident in HBM."""

    def __init__(self, device, frame_sets, seed0, batch, ba_iters, shard=None):
        """shard = (rank, n_ranks, unique_id): config 5 -- this context holds landmark shard `rank` of ONE BA problem"""
        from vo_mi355x import VoContext, sharding, synthetic as syn
        self.B = batch
        self.c = VoContext(W_IMG, H_IMG, max_pts=max(N_PTS, N_NEW), device=device, batch=batch)
        c = self.c
        c.upload_sequence(np.stack([frame_sets[b % len(frame_sets)] for b in range(batch)]))
        self.nf = frame_sets[0].shape[0]
        pts0 = np.stack([syn.grid_points(N_PTS, W_IMG, H_IMG, seed=seed0 + b) for b in range(batch)])
        c.points_upload(pts0)
        # DLT: 1000 new tracks between two window poses of each BA scene; BA: N = 2000, W = 10 per sequence
        kw = {} if K_CAM is None else dict(K=K_CAM, width=W_IMG, height=H_IMG)
        self.n_ba = 1
        if shard is None:
            # a BANK of distinct problems per sequence, one per frame in turn (4 ... 21 LM iterations here)
            self.n_ba = len(BA_VARIANTS)
            bank = [ba_bank(seed0, b, **kw) for b in range(batch)]
            scenes = [bk[0] for bk in bank]
        else:
            scenes = [syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=seed0 + b, **kw) for b in range(batch)]
        d = [dlt_inputs(s) for s in scenes]
        Ks = [s["K"] for s in scenes]
        c.dlt_upload(*[np.stack([d[b][k] for b in range(batch)]) for k in range(7)])
        if shard is None:
            c.ba_upload_bank(np.stack(Ks), np.stack([[bank[b][k]["poses0"] for b in range(batch)] for k in range(self.n_ba)]),
                             np.stack([[bank[b][k]["points0"] for b in range(batch)] for k in range(self.n_ba)]),
                             np.stack([[bank[b][k]["obs"] for b in range(batch)] for k in range(self.n_ba)]))
        else:
            rank, n_ranks, uid = shard
            s = scenes[0]
            # RCCL writes a version banner to C stdout when the first communicator is made; stdout carries the ONE JSON
            # line, so the banner is sent to stderr
            import ctypes
            libc, saved = ctypes.CDLL(None), os.dup(1)
            sys.stdout.flush()
            os.dup2(2, 1)
            try:
                c.comm_init(n_ranks, rank, uid)
                c.sync()
            finally:
                libc.fflush(None)
                os.dup2(saved, 1)
                os.close(saved)
            c.ba_set_sharded(True)
            c.ba_upload(*sharding.shard_problem(s["K"], s["poses0"], s["points0"], s["obs"], n_ranks, first=rank, count=1))
        self.ba_prm = c.ba_params(max_iters=ba_iters, ftol=1e-3, xtol=1e-3)
        self.ba_iters_cap, self.adaptive = ba_iters, True
        self.klt_prm = c.klt_params()
        self.st_prm = c.st_params()
        c.push_frame_resident(0)
        self.truncated = 0                 # solves the iteration budget cut (each is completed at once, see fetch)
        self.at_cap = 0                    # solves the LM's own max_iters (--ba-iters) stopped
        self.iter_hist = {}                # LM iterations per solve over the timed regions
        self.groups_enq = self.groups_needed = self.n_steps = 0
        self.recent = []                   # iterations the last fetched frames needed (maximum over the batch)
        self.pending = []                  # (bank index, budget) of the steps in flight, oldest first
        self.stationary = False            # True: every frame solves problem 0 of the bank (the round-2 benchmark; informational)
        self.stages = (True, True, True)   # (DLT, BA, Shi-Tomasi) of the fused step
        self.t = 1
        self.inflight = 0
        self.max_inflight = 2          # plain launches and hipGraph replay alike (a graph per pinned mirror half)
        self.host = None               # use_host_frames(): per frame index the step's images as host arrays
        self.frame_sets = frame_sets

    def use_host_frames(self, on=True):
        """the frames of every following step are handed over by the host (vo_frame_step_host) instead of read from the sequence store in HBM:
        the distinct synthetic sequences live in page-locked host memory (what a loader decodes into, VoContext.host_alloc), sequence b's image
        of a step is a numpy view of it -- `batch` arrays and `batch` x width x height bytes over PCIe per step"""
        if not on:
            self.host = None
            return 0
        if getattr(self, "_host_store", None) is None:
            self._host_store = self.c.host_alloc((len(self.frame_sets), self.nf, H_IMG, W_IMG))
            for k, fs in enumerate(self.frame_sets):
                self._host_store[k] = fs
            ns = len(self.frame_sets)
            self._host_sets = [self.c.host_frames([self._host_store[b % ns, f] for b in range(self.B)]) for f in range(self.nf)]
        self.host = self._host_sets
        return self.B * W_IMG * H_IMG

    def enqueue(self):
        # one C call: pyramid + KLT + DLT + BA + Shi-Tomasi + result copies for the whole batch; the frames are a closed loop
        # (frame nf continues into frame 0), the BA problem of the step comes from the bank in turn
        k = 0 if self.stationary else self.t % self.n_ba
        if self.n_ba > 1 and self.stages[1]:
            self.c.ba_select(k)
        self.pending.append((k, self.ba_prm.max_iters))
        if self.host is not None:
            # this step's images come from the HOST (one numpy array per sequence, as Pipeline.step(img) receives them: pipeline.py:98,171-172)
            self.c.frame_step_host(self.host[self.t % self.nf], N_PTS, self.stages[0], self.stages[1], self.stages[2], 7, self.klt_prm,
                                   self.st_prm, self.ba_prm)
        else:
            self.c.frame_step_resident(self.t % self.nf, N_PTS, self.stages[0], self.stages[1], self.stages[2], 7, self.klt_prm,
                                       self.st_prm, self.ba_prm)
        self.t += 1
        self.inflight += 1

    def step(self):
        # software pipeline: frame t + 1 is enqueued before frame t's results are waited for, so the stream never drains
        # while the host unpacks (the library keeps two pinned result mirrors for exactly this)
        self.enqueue()
        if self.inflight == self.max_inflight:
            self.fetch()

    def drain(self):
        while self.inflight:
            self.fetch()

    def fetch(self):
        self.last = self.c.frame_fetch()
        self.inflight -= 1
        k, budget = self.pending.pop(0)
        if self.stages[1]:
            st = self.last["ba_stats"]
            st = st if isinstance(st, list) else [st]
            # LM status 0 = stopped by max_iters, which the fused step uses as its BLIND BUDGET (iterations enqueued without looking;
            # they exit early once the LM's own ftol / xtol tests have stopped it).  A solve the budget cut short of --ba-iters is
            # not a result: it is run again from its x0 with the full --ba-iters right here, inside the timed region (same
            # iterations, deterministic reductions: the result of an uncut solve), and replaces the cut one
            cut = [b for b, x in enumerate(st) if x["status"] == 0 and budget < self.ba_iters_cap]
            if cut:
                self.truncated += len(cut)
                self.c.sync()                                      # (the pipelined stream layout runs the next step's BA on another stream)
                if self.n_ba > 1:
                    self.c.ba_select(k)
                po, pt, st2 = self.c.ba_fetch_after(self.c.ba_solve_resident, self.c.ba_params(max_iters=self.ba_iters_cap, ftol=1e-3, xtol=1e-3))
                st2 = st2 if isinstance(st2, list) else [st2]
                for b in cut:
                    st[b] = st2[b]
                self.last["ba_stats"], self.last["ba_poses"], self.last["ba_points"] = (st if len(st) > 1 else st[0]), po, pt
            self.at_cap += sum(1 for x in st if x["status"] == 0)
            its = max(x["iters"] for x in st)
            for x in st:
                self.iter_hist[x["iters"]] = self.iter_hist.get(x["iters"], 0) + 1
            self.groups_enq += budget; self.groups_needed += its; self.n_steps += 1
            self.recent = (self.recent + [its])[-16:]
            if self.adaptive:
                # next budget: one more than the most any of the last 16 fetched frames needed (maximum over the batch), never more than
                # --ba-iters.  Every blind group beyond the need costs ~2 % of a step; a budget below the need costs a whole second solve.
                # Until one cycle of the bank has been seen the budget stays at the cap (a short warm-up must not start the timed region
                # on a budget learnt from two frames: the driver's `--warmup 5` cut and re-ran two solves in its first region).
                self.ba_prm.max_iters = self.ba_iters_cap if len(self.recent) < self.n_ba else max(3, min(self.ba_iters_cap, max(self.recent) + 1))
            else:
                self.ba_prm.max_iters = self.ba_iters_cap
        return self.last

    def ba_stats0(self):
        st = self.last["ba_stats"]
        return st[0] if isinstance(st, list) else st


# ---------------------------------------------------------------------------------------------------------------------------
# closed loop: Pipeline.step (reference pipeline.py:92-167) resident on the device
# ---------------------------------------------------------------------------------------------------------------------------
PIPE_T1 = 4          # the bootstrap pair is (frame 0, frame 4) of a sequence (the reference uses (0, 10): datasets.yml)


def pipe_scenes(n_scenes, n_frames, seed0):
    """rendered KITTI-shaped two-plane scenes, camera swaying with period n_frames (frame n_frames - 1 is followed seamlessly by frame 0)"""
    from vo_mi355x import synthetic as syn
    return [syn.sway_scene(n_frames, w=W_IMG, h=H_IMG, f=718.856, seed=seed0 + k, pose_fn=lambda t: syn.sway_pose(t, period=float(n_frames)))
            for k in range(n_scenes)]


def pipe_phase_offsets(scene, n):
    """start frames whose bootstrap pair (t, t + PIPE_T1) has a usable baseline (the sway stands still at its turning points)"""
    P = scene["poses"]
    nf = len(P)
    base = np.array([np.linalg.norm((P[(t + PIPE_T1) % nf] @ np.linalg.inv(P[t]))[:3, 3]) for t in range(nf)])
    good = [t for t in range(nf) if base[t] >= 0.5 * base.max()]
    return [good[(3 * i) % len(good)] for i in range(n)]


class PipeGroup:
    """`batch` sequences in ONE batched context through the device-resident Pipeline.step: per frame one enqueue (pyramid, KLT of the
    live landmark + candidate keypoints, list bookkeeping, RANSAC-P3P pose + pruning, triangulation of ripe candidates + promotion,
    resurrection of recently dead landmarks, 10-frame bundle adjustment + write-back, Shi-Tomasi re-detection + spawn); every stage
    reads what the previous stages and frames left in the device tables.  Up to 3 steps in flight; only the small records come back."""

    def __init__(self, device, scenes, boot_ctx, first, batch, ba_iters, max_pts, fixed_budget, ba_window, resurrect, host_frames=False):
        from vo_mi355x import VoContext, synthetic as syn
        from vo_mi355x.resident import ResidentPipeline
        self.B, self.nf = batch, len(scenes[0]["frames"])
        self.c = VoContext(W_IMG, H_IMG, max_pts=max_pts, device=device, batch=batch)
        frames, states, Ks, self.gt, where = [], [], [], [], []
        for b in range(batch):
            sc = scenes[(first + b) % len(scenes)]
            off = pipe_phase_offsets(sc, first + b + 1)[-1]
            where.append(((first + b) % len(scenes), off))
            roll = dict(frames=np.roll(sc["frames"], -off, axis=0), poses=np.roll(sc["poses"], -off, axis=0), K=sc["K"], f=sc["f"],
                        surface=lambda t, xy, sc=sc, off=off: sc["surface"]((t + off) % self.nf, xy))
            st, _ = syn.gt_bootstrap(boot_ctx, roll, 0, PIPE_T1)
            frames.append(roll["frames"]); states.append(st); Ks.append(sc["K"])
            G0 = roll["poses"][0]
            unit = np.linalg.norm((roll["poses"][PIPE_T1] @ np.linalg.inv(G0))[:3, 3])
            self.gt.append((roll["poses"], G0, unit))
        self.host = None
        if host_frames:
            # the scenes' frames in page-locked host memory (what a loader decodes into); sequence b's image of a step is a numpy view of it
            store = self.c.host_alloc((len(scenes), self.nf, H_IMG, W_IMG))
            for k, sc in enumerate(scenes):
                store[k] = sc["frames"]
            self._store = store
            self.host = [self.c.host_frames([store[k, (f + off) % self.nf] for k, off in where]) for f in range(self.nf)]
        else:
            self.c.upload_sequence(np.stack(frames))
        self.ba_cap, self.fixed = ba_iters, fixed_budget
        self.rp = ResidentPipeline(self.c, np.stack(Ks), ba_window=ba_window, ba_max_iters=ba_iters, ba_budget=ba_iters, pnp_blind_batches=2, resurrect=resurrect)
        self.rp.seed(states, None, None, t_step=1)
        if host_frames:
            self.c.push_frame(np.stack([fr[PIPE_T1] for fr in frames]) if batch > 1 else frames[0][PIPE_T1])
        else:
            self.c.push_frame_resident(PIPE_T1)
        self.frame, self.inflight, self.max_inflight = PIPE_T1 + 1, 0, 3
        self.recs = []                      # records of the timed region (kept for the statistics)
        self.keep = False
        self.budget = ba_iters
        self.last = None

    def enqueue(self):
        if self.host is not None:
            self.rp.step_host(self.host[self.frame % self.nf])
        else:
            self.rp.step(self.frame % self.nf)
        self.frame += 1
        self.inflight += 1

    def fetch(self):
        rec = self.rp.fetch()
        self.last = rec if isinstance(rec, list) else [rec]           # (a dict when the context carries ONE sequence)
        self.inflight -= 1
        if self.keep:
            self.recs.append(self.last)
        if not self.fixed:
            # the LM stops by its own tests; the budget only bounds the (early-exiting) launches enqueued blindly: what the newest
            # fetched frame needed over the batch + 2, never more than --ba-iters.  A solve the budget cut shows ba_done == 0.
            need = max(r["ba_iters"] for r in self.last) + 2
            b = max(3, min(self.ba_cap, need))
            if b != self.budget:
                self.rp.set_ba_budget(b)
                self.budget = b

    def step(self):
        self.enqueue()
        if self.inflight == self.max_inflight:
            self.fetch()

    def drain(self):
        while self.inflight:
            self.fetch()

    def pose_errors(self):
        """(rotation error in degrees, translation error in bootstrap baselines) of the newest fetched pose against the rendered ground truth"""
        out = []
        for b, r in enumerate(self.last):
            poses, G0, unit = self.gt[b]
            fidx = (PIPE_T1 + r["t"] - 1) % self.nf
            gt = poses[fidx] @ np.linalg.inv(G0)
            H = r["H"]
            cosang = (np.trace(H[:3, :3] @ gt[:3, :3].T) - 1) / 2
            out.append((float(np.degrees(np.arccos(np.clip(cosang, -1, 1)))), float(np.linalg.norm(H[:3, 3] - gt[:3, 3] / unit))))
        return out


def run_pipeline(device, a, dist, n_ctx, per_ctx, steps, warmup, regions, scenes=None):
    """the timed closed loop -> dict for the bench line (also used, smaller, for the informational `pipeline_step` key of the default run)"""
    from vo_mi355x import VoContext
    t0 = time.perf_counter()
    max_pts = a.pipe_max_pts
    if scenes is None:
        scenes = pipe_scenes(2, a.pipe_frames, 4321 + 16 * dist.rank)
    boot = VoContext(W_IMG, H_IMG, max_pts=4096, device=device)
    groups = [PipeGroup(device, scenes, boot, i * per_ctx, per_ctx, a.pipe_ba_iters, max_pts, not a.pipe_adaptive_budget, a.pipe_window, not a.pipe_no_resurrect, a.pipe_host_frames) for i in range(n_ctx)]
    boot.close()
    t_setup = time.perf_counter() - t0
    pool = None
    if n_ctx > 1 and a.host_threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(min(a.host_threads, n_ctx))

    def each(fn):
        if pool is not None:
            list(pool.map(fn, groups))
        else:
            for g in groups:
                fn(g)
    for _ in range(warmup):
        each(lambda g: g.step())
    each(lambda g: g.drain())
    for g in groups:
        g.keep = True
        g.c.profile_enable((g.c.PROF_KLT,))
        g.c.sync()
    region_dt = []
    for _ in range(max(1, regions)):
        for g in groups:
            g.c.sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            each(lambda g: g.step())
        each(lambda g: g.drain())
        for g in groups:
            g.c.sync()
        dist.barrier()
        region_dt.append(dist.max(time.perf_counter() - t0))
    dt = float(np.median(region_dt))
    klt_ms, klt_n = 0.0, 0
    for g in groups:
        ms, n = g.c.profile_read(g.c.PROF_KLT)
        klt_ms += ms; klt_n += n
        g.c.profile_enable(())
    # roofline of the closed loop's own KLT launch: algorithmic bytes = sum over the LIVE keypoints of the launch and the levels of
    # 5120 + 1024 * iterations, from the iteration counts the last launch of context 0 left behind, / the launch's duration from hipEvents
    g0 = groups[0]
    n_live = [r["n_tracked"] for r in g0.last]
    _, _, _, it = g0.c.points_download(max_pts, return_iters=True)
    it = np.maximum(it.reshape(g0.B, max_pts, -1), 0)
    live_bytes = sum(float((5120.0 + 1024.0 * it[b, :n_live[b]]).sum()) for b in range(g0.B))
    it_mean = (sum(it[b, :n_live[b]].sum(0) for b in range(g0.B)) / max(1, sum(n_live))).tolist()
    klt_avg_s = (klt_ms / max(klt_n, 1)) * 1e-3
    achieved = live_bytes / klt_avg_s / 1e9 if klt_avg_s > 0 else 0.0
    roof = {"bound": "hbm", "kernel": "k_klt_track", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": None, "avg_launch_us": round(klt_avg_s * 1e6, 3), "algorithmic_bytes_per_launch": int(live_bytes),
            "live_keypoints_per_launch": int(sum(n_live)), "klt_mean_iters_per_level": [round(x, 3) for x in it_mean],
            "valu": valu_roofline(klt_avg_s, int(sum(n_live))),
            "note": "the closed loop's KLT launch tracks what the device tables hold (landmark keypoints + candidates of every sequence of the context); it runs "
                    "on the side stream beside the bundle adjustment of the previous frame, so its duration includes what that overlap costs"}
    recs = [r for g in groups for step_recs in g.recs for r in step_recs]
    errs = np.array([e for g in groups for e in g.pose_errors()])
    alive = sum(1 for g in groups for r in g.last if r["status"] == 0)
    n_seq = n_ctx * per_ctx
    it = np.array([r["ba_iters"] for r in recs if r["status"] == 0])
    live = [r for r in recs if r["status"] == 0]

    def mean(k):
        return round(float(np.mean([r[k] for r in live])), 1) if live else 0.0
    # frames that count: only those of sequences still alive (a sequence whose status is set makes every list kernel return at once)
    frames_ok = len(live) / max(1, len(region_dt))
    out = {"frames_per_s": round(frames_ok / dt, 1), "frames_per_s_if_stopped_sequences_counted": round(n_seq * steps / dt, 1),
           "ms_per_step": round(dt / steps * 1e3, 4), "sequences": n_seq, "contexts": n_ctx,
           "steps": steps, "regions_ms_per_step": [round(x / steps * 1e3, 4) for x in region_dt],
           "sequences_alive_at_end": alive, "frames_in_sequence": scenes[0]["frames"].shape[0], "max_tracked_keypoints": max_pts,
           "frames_source": "host (page-locked arrays -> vo_pipe_step_host)" if a.pipe_host_frames else "resident in HBM (vo_seq_upload)",
           "ba_window": a.pipe_window, "resurrection_of_dead_landmarks": not a.pipe_no_resurrect, "ba_lm_iteration_cap": a.pipe_ba_iters, "ba_budget": "adaptive (newest fetched frame's maximum + 2)" if a.pipe_adaptive_budget else "the LM's full --ba-iters every frame (surplus groups exit early)",
           "mean_tracked_keypoints": mean("n_tracked"), "mean_landmark_entries": mean("n_landmarks"), "mean_candidates": mean("n_candidates"),
           "mean_pnp_inliers": mean("pnp_inliers"), "mean_new_landmarks": mean("n_new"), "mean_resurrected": mean("n_resurrected"),
           "mean_detected": mean("n_detected"), "mean_ba_observations": mean("ba_observations"),
           "ba_iterations_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique(it, return_counts=True))} if len(it) else {},
           "ba_solves_cut_by_the_budget": int(sum(1 for r in live if r["ba_done"] == 0)),
           "pnp_bound_not_reached": int(sum(1 for r in live if r["pnp_bound_reached"] == 0)),
           "capacity_policy_frames": {name: int(sum(1 for r in live if r["overflow"] & bit)) for name, bit in
                                      (("dead_list", 1), ("promotion", 2), ("resurrection", 4), ("detection", 8), ("st_candidates", 16))},
           "pose_error_vs_ground_truth": {"rotation_deg_median": round(float(np.median(errs[:, 0])), 4), "rotation_deg_max": round(float(errs[:, 0].max()), 4),
                                          "translation_baselines_median": round(float(np.median(errs[:, 1])), 4),
                                          # which sequences (index in the batch: scene = index % 2, start phase by index) are furthest off at the end
                                          "worst_sequences_deg": [[int(i), round(float(errs[i, 0]), 3)] for i in np.argsort(-errs[:, 0])[:3]]},
           "klt_avg_launch_us": round(klt_ms / max(klt_n, 1) * 1e3, 2), "roofline": roof, "setup_s": round(t_setup, 2)}
    for g in groups:
        g.c.close()
    if pool is not None:
        pool.shutdown()
    return out



def dropin_step_ms(device, scene, ba_window=4, n_warm=4, n_time=8):
    """What a user of the reference's own interface gets: Pipeline.step (pipeline.py:92-167) driven from Python over the drop-in
    Extractor / BundleAdjuster classes -- lists of Keypoint / Landmark objects in and out of every call, synchronous, host buffers --
    on ONE sequence of the closed-loop scene; beside it the same sequence through the device-resident tables (ResidentPipeline, one
    enqueue per frame).  -> dict"""
    import copy
    from vo_mi355x import BundleAdjuster, Extractor, VoContext, synthetic as syn
    from vo_mi355x.resident import ResidentPipeline
    frames, K = scene["frames"], scene["K"]
    out = {}

    def reference_loop(c, state, lazy, literal, n_warm, n_time):
        """pipeline.py:92-167 line for line over the drop-in classes; literal: `if i in inliers` on the list camera_pose returned (as the reference
        writes it; a plain list of ~3 000 indices makes that O(n^2), so the plain classes are timed with a set there, as in round 3)"""
        st = copy.deepcopy(state)
        ex = Extractor(min_kp_dist=7, ctx=c, lazy=lazy)
        ba = BundleAdjuster(verbosity=0, window_size=ba_window, method='trf', xtol=1e-3, ftol=1e-3, ctx=c, max_iters=10)
        ex._im_prev = frames[PIPE_T1]
        dead, dead_kp, t_step = [], [], 1
        times = []
        for s in range(n_warm + n_time):
            im = frames[(PIPE_T1 + 1 + s) % len(frames)]
            t0 = time.perf_counter()
            t_step += 1
            st._candidates_kp = ex.extend_tracks(im, st._candidates_kp, max_bidir_error=np.inf)
            st._landmarks, st._landmarks_kp, ld, lkd = ex.extend_landmarks(im, st._landmarks, st._landmarks_kp, max_bidir_error=np.inf)
            dead += copy.deepcopy(ld); dead_kp += copy.deepcopy(lkd)
            ex._im_prev = im.copy()
            inl, Hk = ex.camera_pose(K, st._landmarks, st._landmarks_kp, corr='3D-2D', max_err_reproj=2.0)
            keep = inl if literal else set(inl)
            lms, lkp = [], []
            for i in range(len(st._landmarks)):
                if i in keep:
                    lms.append(st._landmarks[i]); lkp.append(st._landmarks_kp[i])
                else:
                    dead.append(copy.deepcopy(st._landmarks[i])); dead_kp.append(copy.deepcopy(st._landmarks_kp[i]))
            st._landmarks, st._landmarks_kp = lms, lkp
            st._trajectory.append(t_step, Hk)
            l_new, lk_new, st._candidates_kp = ex.triangulate_tracks(K, st._candidates_kp, st._trajectory, t_curr=t_step, min_track_length=3,
                                                                     min_bearing_angle=0.5, max_err_reproj=2.0, refine=True)
            st._landmarks_kp += lk_new; st._landmarks += l_new
            st, dead, dead_kp = ba.adjust(st, dead, dead_kp, K, t_step)
            st._candidates_kp += ex.extract(im, t_step, st._landmarks_kp + st._candidates_kp, detector='shi-tomasi', mask_radius=7, describe=False)
            times.append(time.perf_counter() - t0)
        return times, st, ex

    with VoContext(W_IMG, H_IMG, max_pts=8192, device=device) as c:       # (the scene's steady state is ~4 400 keypoints)
        state, _ = syn.gt_bootstrap(c, scene, 0, PIPE_T1)
        times, st, ex = reference_loop(c, state, False, False, n_warm, n_time)
        out["python_objects_ms_per_step"] = round(float(np.median(times[n_warm:])) * 1e3, 3)
        out["python_objects_frames_per_s"] = round(1.0 / float(np.median(times[n_warm:])), 1)
        out["landmarks"], out["candidates"] = len(st._landmarks), len(st._candidates_kp)
        # the same caller, `if i in inliers` written as the reference writes it, over the lazy boundary (vo_mi355x/lazy.py: the lists are views of the
        # device tables; frame 1 takes the plain path and seeds them)
        try:
            times, st2, ex2 = reference_loop(c, state, True, True, n_warm, 4 * n_time)
            sess = ex2._lazy
            med = float(np.median(times[n_warm:]))
            out["lazy_views_ms_per_step"] = round(med * 1e3, 3)
            out["lazy_views_frames_per_s"] = round(1.0 / med, 1)
            out["lazy_views"] = {"session_alive": bool(sess is not None and sess.alive), "fast_calls": sess.stats["fast"] if sess else 0,
                                 "row_gathers": sess.stats["gathers"] if sess else 0, "frames": len(times),
                                 "first_frame_plain_path_and_seed_ms": round(times[0] * 1e3, 2),
                                 "landmarks": len(st2._landmarks), "candidates": len(st2._candidates_kp),
                                 "why_not": None if sess is not None else getattr(ex2, "_lazy_error", "the frame did not come through the reference's call order")}
            if sess is not None and sess.alive:
                sess.desync("bench done")
        except Exception as e:
            out["lazy_views"] = {"error": repr(e)}
        # the same sequence, state in device tables
        c.upload_sequence(frames)
        rp = ResidentPipeline(c, K, ba_window=ba_window, ba_max_iters=10, pnp_blind_batches=2)
        rp.seed(copy.deepcopy(state), [], [], 1)
        c.push_frame_resident(PIPE_T1)
        f = PIPE_T1 + 1
        for _ in range(6):
            rp.step(f % len(frames)); f += 1
            rp.fetch()
        n = 60
        t0 = time.perf_counter()
        for k in range(n):
            rp.step(f % len(frames)); f += 1
            if k >= 2:
                rp.fetch()
        rp.fetch(); rp.fetch()
        dt = time.perf_counter() - t0
        out["resident_tables_ms_per_step"] = round(dt / n * 1e3, 4)
        out["resident_tables_frames_per_s"] = round(n / dt, 1)
    out["what"] = ("ONE 1241x376 sequence, window %d: Pipeline.step over the drop-in classes -- python_objects: every call gathers / scatters Python "
                   "objects (host buffers, synchronous); lazy_views: the lists are views of the device tables (vo_mi355x/lazy.py), one vo_pipe_step stage "
                   "per call, the reference's own caller loop (`if i in inliers` for every landmark, two appends) included -- vs the same step resident "
                   "in device tables without any Python objects (vo_pipe_step, 3 steps in flight)" % ba_window)
    return out



def render_sequences(seeds, n_frames, parallel=True):
    """the distinct synthetic image sequences of a rank (`synthetic.make_sequence(n_frames, periodic=True)` of each seed, frame by frame), rendered by
    a pool of THREADS -- numpy drops the interpreter lock inside its array operations: ~4-6 x on 8-16 cores (8 x 100 frames of 1241 x 376 take
    ~40 s on one).  Not processes: forked workers hung the run under `rocprofv3 --pmc` once in ten (the profiler has initialised the GPU and
    installed its signal handlers before the fork; `Pool` terminates its workers with SIGTERM and waited for them for ever -- 40 GPU-minutes)."""
    from vo_mi355x import synthetic as syn
    margin = 96
    n_thr = max(1, min(usable_cores(), 16)) if parallel else 1
    if n_thr == 1:
        return [syn.make_sequence(n_frames, W_IMG, H_IMG, seed=sd, periodic=True)[0] for sd in seeds]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(n_thr) as ex:
        texs = list(ex.map(lambda sd: syn.make_texture(H_IMG + 2 * margin, W_IMG + 2 * margin, sd), seeds))
        out = [np.empty((n_frames, H_IMG, W_IMG), np.uint8) for _ in seeds]

        def one(job):
            k, t = job
            out[k][t] = syn.render_frame(texs[k], syn.frame_motion_periodic(t, W_IMG, H_IMG, n_frames), W_IMG, H_IMG, margin)
        list(ex.map(one, [(k, t) for k in range(len(seeds)) for t in range(n_frames)]))
    return out


def usable_cores():
    """host cores this process may actually use: os.cpu_count() capped by the cgroup CPU quota (the GPU box shows 256 cores and
    grants 16: 64 workers there were slower in total than 16)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:           # noqa: BLE001
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:           # noqa: BLE001
        pass
    return n


def cpu_baseline_parallel(n_procs, n_frames, ba_iters):
    """cpu_baseline on `n_procs` host cores: independent sequences, one single-threaded worker PROCESS per core (this
    file re-run with --cpu-worker), started before this process touches the GPU.  -> (frames/s, seconds, cores)"""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), "--cpu-frames", str(n_frames),
                               "--ba-iters", str(ba_iters)], stdout=subprocess.PIPE, env=env, text=True) for i in range(n_procs)]
    done, slowest = 0, 0.0
    for pr in procs:
        out, _ = pr.communicate()
        if pr.returncode == 0 and out.strip():
            r = json.loads(out.strip().splitlines()[-1])
            done += r["frames"]
            slowest = max(slowest, r["seconds"])
    wall = time.perf_counter() - t0
    extra = {}
    # the side job (the reference's solver recipe, live OpenCV timings) runs AFTER the workers: beside them it would take a core from the
    # timed baseline (17 busy processes on 16 granted cores)
    side = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "10000"], stdout=subprocess.PIPE, env=env, text=True)
    try:
        so, _ = side.communicate(timeout=600)
        if side.returncode == 0 and so.strip():
            extra = json.loads(so.strip().splitlines()[-1])
    except Exception:           # noqa: BLE001
        side.kill()
    if done == 0:
        return 0.0, wall, 0, extra
    return done / slowest, wall, n_procs, extra


def cpu_pipeline_worker(a):
    """one sequence of the closed loop on ONE host core: the table model of oracle/pipe_oracle.py (the algorithm csrc/vo_pipeline.hip implements, pinned
    to the reference's own Pipeline.step by tests/golden/pipe_*.npz) over the CPU oracle (C: pyramid, KLT, Shi-Tomasi, DLT; numpy: P3P-RANSAC, BA)"""
    for sub in ("oracle",):
        sys.path.insert(0, os.path.join(ROOT, sub))
    import pipe_oracle as po
    from oracle_context_impl import OracleContext
    from vo_mi355x import synthetic as syn
    sc = pipe_scenes(1, a.pipe_frames, 4321 + a.cpu_pipe_worker)[0]
    ctx = OracleContext(W_IMG, H_IMG)
    off = pipe_phase_offsets(sc, a.cpu_pipe_worker + 1)[-1]
    nf = len(sc["frames"])
    roll = dict(frames=np.roll(sc["frames"], -off, axis=0), poses=np.roll(sc["poses"], -off, axis=0), K=sc["K"], f=sc["f"],
                surface=lambda t, xy: sc["surface"]((t + off) % nf, xy))
    state, _ = syn.gt_bootstrap(ctx, roll, 0, PIPE_T1)
    m = po.PipeModel(ctx, sc["K"], W_IMG, H_IMG, cap=a.pipe_max_pts,
                     params=po.Params(ba_window=a.pipe_window, ba_max_iters=a.pipe_ba_iters, resurrect=not a.pipe_no_resurrect))
    m.seed(state, [], [], 1)
    ctx.push_frame(roll["frames"][PIPE_T1])
    f = PIPE_T1 + 1
    for _ in range(2):
        m.step(roll["frames"][f % nf]); f += 1
    t0 = time.perf_counter()
    for _ in range(a.cpu_pipe_frames):
        m.step(roll["frames"][f % nf]); f += 1
    secs = time.perf_counter() - t0
    print(json.dumps({"frames": a.cpu_pipe_frames, "seconds": secs, "status": m.status, "landmarks": len(m.lm_L), "candidates": len(m.cand)}))


def cpu_pipeline_baseline(a, n_procs):
    """-> the `cpu_baseline` object of the closed loop: n_procs single-threaded worker processes (this file re-run with --cpu-pipe-worker), one
    sequence each, started before this process touches the GPU"""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.perf_counter()
    argv = ["--cpu-pipe-frames", str(a.cpu_pipe_frames), "--pipe-ba-iters", str(a.pipe_ba_iters), "--pipe-frames", str(a.pipe_frames), "--pipe-window", str(a.pipe_window),
            "--pipe-max-pts", str(a.pipe_max_pts)] + (["--pipe-no-resurrect"] if a.pipe_no_resurrect else [])
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-pipe-worker", str(i)] + argv, stdout=subprocess.PIPE, env=env, text=True)
             for i in range(n_procs)]
    done, slowest, lm = 0, 0.0, []
    for pr in procs:
        out, _ = pr.communicate()
        if pr.returncode == 0 and out.strip():
            r = json.loads(out.strip().splitlines()[-1])
            if r["status"] == 0:
                done += r["frames"]; slowest = max(slowest, r["seconds"]); lm.append(r["landmarks"] + r["candidates"])
    wall = time.perf_counter() - t0
    v = done / slowest if done else 0.0
    return {"value": round(v, 3), "unit": "frames/s", "cores": n_procs, "cores_available": usable_cores(), "cores_visible": os.cpu_count() or 0, "kind": "port",
            "per_core": round(v / max(n_procs, 1), 3),
            "sample": "%d worker processes (1 core each, one sequence each) x %d frames of the closed loop (after 2 untimed) through the table model of "
                      "oracle/pipe_oracle.py over the CPU oracle (C: pyramid + KLT + Shi-Tomasi + DLT, numpy: P3P-RANSAC + BA), window %d, %d-slot tables, "
                      "~%d keypoints per sequence; %.1f s wall" % (n_procs, a.cpu_pipe_frames, a.pipe_window, a.pipe_max_pts, int(np.mean(lm)) if lm else 0, wall)}


def reference_recipe_ba_seconds():
    """ONE bundle adjustment of the workload's shape (2000 landmarks x 10 poses, every landmark seen in every frame) solved with
    the REFERENCE'S SOLVER RECIPE -- scipy least_squares(method='trf', 2-point finite differences with jac_sparsity, loss='huber',
    x_scale='jac', ftol = xtol = 1e-3: bundle_adjuster.py:189-194 as configured by pipeline.py:28-29) -- over the oracle's
    vectorised restatement of its objective (the reference's own objective is a Python double loop, 22 ms per evaluation;
    this one takes ~2 ms, so the figure is a LOWER bound of the reference's time: SURVEY.md section 6 measured 9.3 s per call).
    -> dict or None without scipy."""
    try:
        from scipy.optimize import least_squares
        from scipy.sparse import coo_matrix
    except Exception:           # noqa: BLE001
        return None
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ba_oracle as bo
    from vo_mi355x import synthetic as syn
    s = syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=0)
    K, obs = s["K"], s["obs"]
    x0 = bo.pack_x0(s["poses0"], s["points0"])
    rows, cols = bo.sparsity_coo(obs)
    m = int(bo.valid_mask(obs).sum())
    A = coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(m, len(x0))).tocsr()

    def fun(x):
        po, pt = bo.unpack_x(x, BA_N, BA_W)
        return bo.residual_norm(K, po, pt, obs)
    t0 = time.perf_counter()
    res = least_squares(fun, x0, jac_sparsity=A, method="trf", loss="huber", x_scale="jac", ftol=1e-3, xtol=1e-3, verbose=0)
    dt = time.perf_counter() - t0
    return {"seconds_per_adjust": round(dt, 3), "nfev": int(res.nfev), "njev": int(res.njev), "cost": round(float(res.cost), 3),
            "cost0": round(float(0.5 * bo.huber_rho(fun(x0) ** 2).sum()), 3), "cores": 1,
            "what": "scipy TRF, 2-point FD + jac_sparsity, huber, x_scale='jac', ftol=xtol=1e-3 over the oracle's vectorised objective"}


def opencv_baseline(frames):
    """When a real OpenCV is importable on this box: the reference's cv2 calls in its call pattern on one frame pair (4 x
    calcOpticalFlowPyrLK, mask loop + goodFeaturesToTrack, triangulatePoints; tests/live_cv2.py).  None otherwise."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import live_cv2 as lv
        cv2 = lv.find_real_cv2()
        if cv2 is None:
            return None
        from vo_mi355x import synthetic as syn
        p = syn.grid_points(N_PTS, W_IMG, H_IMG, seed=7).astype(np.float32)
        s = syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=0)
        K = s["K"]
        P0 = np.float32(K @ np.hstack([syn.rodrigues(s["poses_gt"][3, :3]), s["poses_gt"][3, 3:, None]]))
        P1 = np.float32(K @ np.hstack([syn.rodrigues(s["poses_gt"][0, :3]), s["poses_gt"][0, 3:, None]]))
        t = lv.time_reference_call_pattern(cv2, frames[0], frames[1], p[:N_PTS // 2], p[N_PTS // 2:], p, P0, P1,
                                           s["obs"][3, :N_NEW].astype(np.float32), s["obs"][0, :N_NEW].astype(np.float32))
        return dict({k: round(v, 6) if isinstance(v, float) else v for k, v in t.items()}, version=cv2.__version__,
                    frames_per_s_front_end=round(1.0 / t["frame_s"], 2))
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)}


def cpu_baseline(frames, n_frames, ba_iters):
    """The CPU oracle (the build's restatement of the reference's OpenCV / SciPy-side arithmetic) on the same
    workload, one host thread, a bounded sample of frames.  Reported, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ba_oracle as bo
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    p = syn.grid_points(N_PTS, W_IMG, H_IMG, seed=7)
    s = syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=0)
    bank = [s] + [syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=7919 * k, obs_noise=v[0], pt_noise=v[1], pose_noise=v[2], visibility=v[3])
                  for k, v in enumerate(BA_VARIANTS) if k > 0]                 # the same bank of problems the GPU sequences cycle through
    K = s["K"]
    P0 = (K @ np.hstack([syn.rodrigues(s["poses_gt"][3, :3]), s["poses_gt"][3, 3:, None]])).astype(np.float32)
    P1 = (K @ np.hstack([syn.rodrigues(s["poses_gt"][0, :3]), s["poses_gt"][0, 3:, None]])).astype(np.float32)
    t0 = time.perf_counter()
    for t in range(n_frames):
        a, b = frames[t % len(frames)], frames[(t + 1) % len(frames)]
        p1, st, err = o.klt(a, b, p)                                   # builds both pyramids, like one cv2 call
        o.triangulate(P0, P1, s["obs"][3, :N_NEW], s["obs"][0, :N_NEW])
        q = bank[t % len(bank)]
        bo.solve(K, q["poses0"], q["points0"], q["obs"], max_iters=ba_iters, ftol=1e-3, xtol=1e-3)
        mask = np.full((H_IMG, W_IMG), 255, np.uint8)
        for x, y in np.int32(p1):
            o.circle_mask(mask, (x, y), 7, 0)
        o.good_features(b, mask)
        p = p1
    dt = time.perf_counter() - t0
    return n_frames / dt, dt


def seeded_points_figure(device, seqs, frame_sets, a, step, drain):
    """the headline's step with the tracked points of every sequence = the strongest N_PTS Shi-Tomasi corners of its frame 0 (the build's own
    detector, the reference's quality level 0.03 and minimum distance 7; a sequence whose frame yields fewer is topped up from the grid)"""
    from vo_mi355x import VoContext, synthetic as syn
    corners = []
    with VoContext(W_IMG, H_IMG, max_pts=max(N_PTS, 4096), device=device) as c:
        for fs in frame_sets:
            c.push_frame(fs[0])
            corners.append(c.shi_tomasi(None, 7, params=c.st_params(max_corners=N_PTS)))
    n_found = [len(x) for x in corners]
    for g in seqs:
        pts = []
        for b in range(g.B):
            p = corners[b % len(corners)]
            if len(p) < N_PTS:
                p = np.concatenate([p, syn.grid_points(N_PTS, W_IMG, H_IMG, seed=b)[:N_PTS - len(p)]])
            pts.append(p[:N_PTS])
        g.c.points_upload(np.stack(pts) if g.B > 1 else pts[0])
        g.c.push_frame_resident(0)
        g.t = 1
    for _ in range(max(6, a.warmup)):
        step()
    drain()
    dts = []
    for _ in range(3):
        for g in seqs:
            g.c.sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        drain()
        for g in seqs:
            g.c.sync()
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    _, st, _, it = seqs[0].c.points_download(N_PTS, return_iters=True)
    it = np.maximum(it.reshape(-1, it.shape[-1]), 0)
    return {"value": round(sum(g.B for g in seqs) * a.steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / a.steps * 1e3, 4),
            "corners_found_per_distinct_sequence": n_found, "klt_mean_iters_per_level": [round(float(x), 3) for x in it.mean(0)],
            "tracked_with_status_1": round(float(st.mean()), 4),
            "what": "tracked points = the build's own Shi-Tomasi corners of frame 0 (max_corners %d, quality 0.03, min distance 7) instead of the jittered grid" % N_PTS}


CHILD_ENV_DROP = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")


def _closed_loop_short(d, cmd, seconds):
    """the short object of a closed-loop result the compact line carries"""
    pp, rf = d.get("pipeline") or {}, d.get("roofline") or {}
    cap = pp.get("capacity_policy_frames") or {}
    return {"value": d.get("value"), "unit": "frames/s", "ms_per_step": d.get("ms_per_step"), "sequences": pp.get("sequences"), "ba_window": pp.get("ba_window"),
            "table_slots": pp.get("max_tracked_keypoints"), "mean_tracked_keypoints": pp.get("mean_tracked_keypoints"),
            "capacity_policy_frames": int(sum(cap.values())) if cap else None, "frames_counted": int(pp.get("sequences", 0)) * int(pp.get("steps", 0)) * len(pp.get("regions_ms_per_step", [])),
            "sequences_alive_at_end": pp.get("sequences_alive_at_end"), "rotation_deg_median": (pp.get("pose_error_vs_ground_truth") or {}).get("rotation_deg_median"),
            "roofline": {"kernel": rf.get("kernel"), "frac": rf.get("frac"), "avg_launch_us": rf.get("avg_launch_us")},
            "cmd": cmd, "seconds": seconds}


def closed_loop_child(a, extra_args, steps=40, timeout=150, warmup=10):
    """`bench.py --workload pipeline ...` as a child process -> the short object the compact line carries; with `--pipe-second-max-pts` among the
    arguments a pair (first run, second run: same scenes, one process).  Never raises: an error or a timeout becomes {"error": ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "pipeline", "--ctxs", "1", "--steps", str(steps), "--warmup", str(warmup), "--regions", "3",
           "--no-cpu-baseline", "--full-line", "--extras-file", os.devnull, "--pipe-frames", str(a.pipe_frames), "--pipe-ba-iters", str(a.pipe_ba_iters)] + \
          (["--tune", a.tune] if a.tune else []) + list(extra_args)
    env = {k: v for k, v in os.environ.items() if k not in CHILD_ENV_DROP}
    pair = "--pipe-second-max-pts" in extra_args
    t0 = time.perf_counter()
    try:
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env)
        d = json.loads([ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")][-1])
    except Exception as e:          # noqa: BLE001
        err = {"error": repr(e)[:200], "seconds": round(time.perf_counter() - t0, 1)}
        return (err, err) if pair else err
    secs = round(time.perf_counter() - t0, 1)
    text = "bench.py --workload pipeline " + " ".join(extra_args)
    first = _closed_loop_short(d, text, secs)
    if not pair:
        return first
    d2 = d.get("second") or {}
    return first, (_closed_loop_short(d2, text + " (second run)", secs) if d2 else {"error": "no second run in the child's result"})


def measure_extras(device, frame_sets, a, dist, cpu_pipe=None):
    """Informational, after the timed regions, rank 0 of a 1-GPU run: the other BASELINE configurations on the same device.
    single_sequence: configs[2] / [3] literally -- ONE sequence in one context (launch-latency bound).
    klt_only: configs[1] -- pyramid + Scharr + KLT of 2000 points per frame, maxLevel 3 (4 levels) and 2, one sequence and a
    batch of 32.  pipeline_step: workload A plus the device-resident track table and the RANSAC-P3P pose (Pipeline.step)."""
    out = {}

    def run(g, n, warm=10):
        for _ in range(warm):
            g.step()
        g.drain(); g.c.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            g.step()
        g.drain(); g.c.sync()
        return time.perf_counter() - t0
    g = Group(device, frame_sets, seed0=7000, batch=1, ba_iters=a.ba_iters)
    dt = run(g, 300)
    side = {"frames_per_s": round(300 / dt, 1), "ms_per_frame": round(dt / 300 * 1e3, 4), "stream_layout": "side stream (re-detection + triangulation beside the BA)"}
    # the layout meant for ONE sequence: three streams, the BA of frame t beside the pyramid + KLT of frame t + 1 (vo_set_side_stream 2)
    g.c.set_side_stream("pipeline")
    dt = run(g, 300)
    out["single_sequence"] = {"frames_per_s": round(300 / dt, 1), "ms_per_frame": round(dt / 300 * 1e3, 4), "sequences": 1, "contexts": 1,
                              "workload": WORKLOAD, "stream_layout": "pipeline (three streams: BA of frame t beside the front end of frame t + 1)",
                              "side_stream_layout": side}
    g.c.set_side_stream(True)
    # the same sequence replayed from captured hipGraphs (one per frame parity, pinned mirror half and bank problem; fixed BA budget:
    # the budget is baked into a capture), two steps in flight
    try:
        g.adaptive = False
        cap0 = g.ba_iters_cap
        g.ba_iters_cap = min(10, cap0)        # (a capture bakes the budget in: the LM cap of rounds 1-4, so that no solve counts as cut and is run again)
        g.ba_prm.max_iters = g.ba_iters_cap
        g.c.set_graph_mode(True)
        dt = run(g, 300, warm=40)
        g.c.set_graph_mode(False)
        dtp = run(g, 300)
        out["single_sequence"]["graph_replay"] = {"frames_per_s": round(300 / dt, 1), "plain_launches_same_settings_frames_per_s": round(300 / dtp, 1),
                                                  "settings": "side-stream layout off under capture (one stream), fixed BA budget = LM cap %d" % g.ba_iters_cap}
        g.adaptive = True
        g.ba_iters_cap = cap0
    except Exception as e:      # noqa: BLE001
        out["single_sequence"]["graph_replay"] = {"error": str(e)}
    # the default layout of rounds 1-3 on this build and box -- 96 sequences in three batched contexts of 32, side stream on, three host threads --
    # as a child process running exactly that command (measured inside this process, beside the other informational contexts, it came out 20 % low)
    try:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--seqs", "96", "--ctxs", "3", "--side-stream", "on", "--host-threads", "3", "--steps", "60",
               "--warmup", "10", "--regions", "3", "--no-extras", "--no-cpu-baseline", "--ba-iters", "10", "--frames", str(a.frames), "--full-line", "--extras-file", os.devnull]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}      # (a 1-rank torchrun launch)
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=env)
        d3 = json.loads([l for l in pr.stdout.decode().splitlines() if l.startswith("{")][-1])
        out["layout_3_contexts_of_32"] = {"frames_per_s": d3["value"], "ms_per_step": d3["ms_per_step"], "sequences": 96, "contexts": 3,
                                          "klt_roofline_frac": d3["roofline"]["frac"],
                                          "what": "the default configuration of rounds 1-3 (BENCH_r01 .. r03) on this build and box: `python bench.py "
                                                  "--seqs 96 --ctxs 3 --side-stream on --host-threads 3 --ba-iters 10 --no-extras --no-cpu-baseline` run as a child process"}
    except Exception as e:      # noqa: BLE001
        out["layout_3_contexts_of_32"] = {"error": str(e)}
    # BASELINE configs[4] at N = 1 (ONE 1920x1080 sequence, 5000 points, 20-frame BA through a 1-rank communicator): the anchor a multi-GPU
    # run of `--workload config5` is compared with (child process: the workload rebinds the module's shape constants)
    try:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", "config5", "--steps", "60", "--warmup", "10", "--regions", "3", "--no-extras",
               "--no-cpu-baseline", "--full-line", "--extras-file", os.devnull]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=env)
        d5 = json.loads([l for l in pr.stdout.decode().splitlines() if l.startswith("{")][-1])
        out["config5_n1"] = {"frames_per_s": d5["value"], "ms_per_step": d5["ms_per_step"], "n_gpus": 1, "workload": d5["config"]["workload"],
                             "ba_lm_iterations_run": d5["config"].get("ba_lm_iterations_run"), "stage_ms": d5.get("stage_ms_per_batched_launch_group"),
                             "what": "`python bench.py --workload config5 --no-extras --no-cpu-baseline` on one GPU: the landmark-sharded solve with ONE shard "
                                     "(k_ba_xsum / k_ba_xstat + the 1-rank RCCL all-reduces in every LM iteration); N > 1 has never run on this pool"}
    except Exception as e:      # noqa: BLE001
        out["config5_n1"] = {"error": str(e)}
    kl = {}
    for batch in (1, 32):
        gk = g if batch == 1 else Group(device, frame_sets, seed0=7100, batch=32, ba_iters=a.ba_iters)
        for lvl in (3, 2):
            gk.klt_prm = gk.c.klt_params(max_level=lvl)
            gk.stages = (False, False, False)
            n = 300 if batch == 1 else 100
            dt = run(gk, n)
            kl["maxLevel%d_batch%d" % (lvl, batch)] = {"frames_per_s": round(n * batch / dt, 1), "ms_per_step": round(dt / n * 1e3, 4)}
        if gk is not g:
            gk.c.close()
    out["klt_only"] = dict(kl, workload="synthetic_1241x376_2000pts_klt_only (BASELINE configs[1]: pyramid + Scharr + KLT, no BA)")
    g.c.close()
    # Pipeline.step resident on the device, closed loop: 96 sequences like the headline.  Two configurations:
    # the reference's own (window 4, recently dead landmarks resurrected into every adjust) and BASELINE's 10-frame window with dead
    # landmarks left dead -- with the reference's resurrection a window of 10 fills the table with copies of young deaths
    import copy as _copy
    scenes = pipe_scenes(2, a.pipe_frames, 4321)
    a4 = _copy.copy(a); a4.pipe_window, a4.pipe_no_resurrect = 4, False
    a10 = _copy.copy(a); a10.pipe_window, a10.pipe_no_resurrect = 10, True
    try:
        # ONE context of 96 sequences: since the tracking of frame t + 1 and the spawn of frame t run beside the adjustment of frame t inside a
        # context (csrc/vo_pipeline.hip), one large batch beats three small ones (39.0 k against 33.1 k frames/s; 128 sequences: 40.9 k)
        # the same with tables the scene does not fill (8 192 slots; the scene's steady state is ~4 400 keypoints): the loop measured is then the
        # reference's unbounded lists, not the capacity policy's (capacity_policy_frames.detection = 0 after the fill-up)
        a4big = _copy.copy(a4); a4big.pipe_max_pts = 8192
        a4ad = _copy.copy(a4); a4ad.pipe_adaptive_budget = True
        out["pipeline_step"] = {"reference_configuration_window4": run_pipeline(device, a4, dist, 1, 96, 40, 10, 3, scenes),
                                # (96 sequences: the batch rounds 2-3 reported; 256: the batch of the headline since the end of round 4)
                                "reference_configuration_window4_256_sequences": run_pipeline(device, a4, dist, 1, 256, 40, 10, 3, scenes),
                                "window10_dead_stay_dead": run_pipeline(device, a10, dist, 1, 96, 40, 10, 3, scenes),
                                "closed_loop_window10_256_sequences": run_pipeline(device, a10, dist, 1, 256, 40, 10, 3, scenes),
                                "one_sequence_window4": run_pipeline(device, a4, dist, 1, 1, 100, 10, 3, scenes),
                                # the same with the LM launch groups bounded by what the newest fetched frame needed + 2 (3 .. --ba-iters)
                                # instead of --ba-iters blind groups every frame: `ba_solves_cut_by_the_budget` counts what that changed
                                "one_sequence_window4_adaptive_budget": run_pipeline(device, a4ad, dist, 1, 1, 100, 10, 3, scenes),
                                "window4_tables_not_full_8192_slots": run_pipeline(device, a4big, dist, 1, 32, 40, 60, 3, scenes)}
        if cpu_pipe is not None:
            out["pipeline_step"]["reference_configuration_window4"]["cpu_baseline"] = cpu_pipe
    except Exception as e:      # noqa: BLE001  (an informational key must not cost the others)
        out["pipeline_step"] = {"error": str(e)}
    try:
        out["dropin_step"] = dropin_step_ms(device, scenes[0])
    except Exception as e:      # noqa: BLE001
        out["dropin_step"] = {"error": str(e)}
    # LAST (a second process of this size beside this one leaves this process's small launches at twice their latency afterwards: the KLT-only
    # entry for one sequence read 0.150 instead of 0.066 ms per frame when this child ran before it).  The headline's command at the LM cap rounds 1-4
    # ran (--ba-iters 10: 1 % of the solves stop there, where the reference's least_squares would go on), for comparison with their numbers:
    # same batch, same layout, child process
    if a.ba_iters != 10:
        try:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(a.steps), "--warmup", "10", "--regions", "3", "--no-extras", "--no-cpu-baseline",
                   "--ba-iters", "10", "--frames", str(a.frames), "--seqs", str(a.seqs), "--full-line", "--extras-file", os.devnull]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                    "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=env)
            dc = json.loads([l for l in pr.stdout.decode().splitlines() if l.startswith("{")][-1])
            out["lm_cap_10"] = {"frames_per_s": dc["value"], "ms_per_step": dc["ms_per_step"], "sequences": a.seqs,
                                "ba_lm_iterations_histogram": dc["config"].get("ba_lm_iterations_histogram"),
                                "ba_solves_stopped_by_lm_max_iters": dc["config"].get("ba_solves_stopped_by_lm_max_iters"),
                                "ba_iteration_groups_enqueued_per_step": dc["config"].get("ba_iteration_groups_enqueued_per_step"),
                                "what": "the default command with `--ba-iters 10`, the cap of rounds 1-4"}
        except Exception as e:      # noqa: BLE001
            out["lm_cap_10"] = {"error": str(e)}
    return out


def _source_digest(names=None):
    """digest of kernel sources, comments and white space not counted (tools/source_digest.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import source_digest
    return source_digest.digest(names)


def _klt_source_digest(with_frame_store=False):
    """digest of the tracker's source; with_frame_store: also of vo_frame.hip, whose 4x derivative format the tracker consumes"""
    return _source_digest(["vo_klt.hip"] + (["vo_frame.hip"] if with_frame_store else []))


def profile_constant(fname, key):
    """a per-launch figure of k_klt_track that only rocprofv3 --pmc can measure (tools/profile_round.sh writes it under profiles/
    together with the digest of the kernel source it was measured on).  -> (value, provenance); the value is None when the
    kernel source has changed since, so a stale constant is never reported as current."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None, "profiles/%s missing" % fname
    try:
        d = json.load(open(path))
    except Exception as e:          # noqa: BLE001
        return None, "profiles/%s unreadable: %s" % (fname, e)
    if "klt_frame_source_sha256_16" in d and d["klt_frame_source_sha256_16"] != _klt_source_digest(True):
        return None, "profiles/%s was measured on another version of vo_klt.hip / vo_frame.hip (stale): re-run tools/profile_round.sh" % fname
    if d.get("klt_source_sha256_16") != _klt_source_digest():
        return None, "profiles/%s was measured on another version of vo_klt.hip (stale): re-run tools/profile_round.sh" % fname
    return d.get(key), "profiles/%s (rocprofv3 --pmc, %s)" % (fname, d.get("measured", "this round"))


VALU_ISSUE_PER_CLK_PER_SIMD = 0.5     # MI355X_MICROARCH.md: a wave64 VALU instruction passes a SIMD-32 in 2 cycles (full-rate ops)
N_SIMDS, CLK_HZ = 1024, 2.4e9         # 256 CUs x 4 SIMDs, 2.4 GHz peak clock
MIXED_CLK_PER_INST, MIXED_CLK_HZ = 4.1, 2.37e9    # tools/issue_probe.hip, "klt mix" rows (profiles/r02_issue_probe.txt)


def kernel_rooflines():
    """per-kernel figures of the default command from the committed rocprofv3 summaries (profiles/kernel_counters.json, written by
    tools/profile_round.sh + tools/summarize_profiles.py; every number is recomputable from the <tag>_kernel_stats_default.csv and
    <tag>_pmc_*_default.csv beside it): share of the kernel time, average launch, HBM bytes per launch against the 8 TB/s peak, vector
    wave-instructions per launch against the SIMDs' nominal issue rate, float64 MFMA flops and MfmaUtil.  Stale (measured on other kernel
    sources) -> reported as such, never as current."""
    path = os.path.join(ROOT, "profiles", "kernel_counters.json")
    if not os.path.exists(path):
        return {"source": "profiles/kernel_counters.json missing", "kernels": []}
    d = json.load(open(path))
    stale = d.get("csrc_sha256_16") != _source_digest()
    peak_issue = VALU_ISSUE_PER_CLK_PER_SIMD * N_SIMDS * CLK_HZ
    out = []
    for e in d.get("kernels", []):
        # rates are taken over the launch as it runs with the chip to itself (the one-stream trace) when that was collected: in the
        # three-stream default a narrow kernel's traced duration is mostly the wait for compute units behind the tracker's launch
        t = e.get("one_stream_avg_launch_us", e["avg_launch_us"]) * 1e-6
        r = {"kernel": e["kernel"], "pct_of_kernel_time": e["pct_of_kernel_time"], "avg_launch_us": round(e["avg_launch_us"], 2), "calls": e["calls"]}
        if "one_stream_avg_launch_us" in e:
            r["one_stream_avg_launch_us"] = round(e["one_stream_avg_launch_us"], 2); r["one_stream_pct_of_kernel_time"] = e["one_stream_pct_of_kernel_time"]
        if "hbm_bytes_per_launch" in e:
            r["hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
            r["hbm_gb_s"] = round(e["hbm_bytes_per_launch"] / t / 1e9, 1)
            r["hbm_frac"] = round(e["hbm_bytes_per_launch"] / t / 1e9 / HBM_PEAK_GBS, 4)
        if "valu_insts_per_launch" in e:
            r["valu_wave_insts_per_launch"] = int(e["valu_insts_per_launch"])
            r["valu_issue_frac"] = round(e["valu_insts_per_launch"] / t / peak_issue, 4)
            r["valu_issue_frac_of_quarter_rate_roof"] = round(e["valu_insts_per_launch"] / t / (N_SIMDS * MIXED_CLK_HZ / MIXED_CLK_PER_INST), 4)
        if "mfma_flops_f64_per_launch" in e and e["mfma_flops_f64_per_launch"] > 0:
            r["mfma_util_pct_mean"], r["mfma_util_pct_max"] = e["mfma_util_pct_mean"], e["mfma_util_pct_max"]
            r["mfma_f64_tflops"] = round(e["mfma_flops_f64_per_launch"] / t / 1e12, 2)
            r["mfma_f64_frac_of_78.6_tflops"] = round(e["mfma_flops_f64_per_launch"] / t / 78.6e12, 4)
        out.append(r)
    # the roof the WHOLE step runs against: every kernel of it is bound by vector-instruction issue (or, the BA build, by float64 MFMAs that
    # block the same issue slots for 64 cycles = 14 quarter-rate instructions: profiles/r05_mfma_overlap_probe.txt), so what a step costs
    # is the instructions it issues.  Per traced step of the default command: sum over the kernels of calls x (vector wave-instructions +
    # 14 x MFMAs) -- filled in by the caller with the measured step time
    steps = max(1, min([e["calls"] for e in d.get("kernels", []) if "k_klt_track" in e["kernel"]] or [1]))
    issue, hbm_step = 0.0, 0.0
    for e in d.get("kernels", []):
        per_launch = e.get("valu_insts_per_launch", 0.0) + 14.0 * e.get("mfma_flops_f64_per_launch", 0.0) / 2048.0
        issue += per_launch * e["calls"] / steps
        hbm_step += e.get("hbm_bytes_per_launch", 0) * e["calls"] / steps
    return {"source": "profiles/kernel_counters.json (%s)%s" % (d.get("measured"), "; STALE: the kernel sources changed since" if stale else ""),
            "issue_slots_per_step": int(issue), "hbm_bytes_per_step": int(hbm_step),
            "files": d.get("files"), "stale": stale,
            "how": "avg_launch_us, pct: rocprofv3 --kernel-trace --stats of the default command (launches averaged over full and tail LM groups); "
                   "hbm = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch / avg launch / 8 TB/s; valu_issue_frac = SQ_INSTS_VALU per launch / avg launch "
                   "/ (0.5 wave-instruction per clk per SIMD x 1024 x 2.4 GHz); quarter-rate roof = 1 / 4.1 clk at 2.37 GHz (float64, dot, DPP, perm: "
                   "profiles/r02_issue_probe.txt); a float64 MFMA occupies its SIMD like 14 such instructions and does not overlap with them "
                   "(profiles/r05_mfma_overlap_probe.txt)",
            "kernels": out}


def valu_roofline(launch_s, n_waves):
    """the roof that actually binds k_klt_track: vector-instruction issue.  SQ_INSTS_VALU per launch comes from the committed
    rocprofv3 --pmc summary (profiles/klt_valu.json); launch time is measured live in this run."""
    insts, src = profile_constant("klt_valu.json", "sq_insts_valu_per_launch")
    waves, _ = profile_constant("klt_valu.json", "sq_waves_per_launch")
    if insts is None or launch_s <= 0:
        return {"wave_insts_per_launch": None, "source": src}
    if waves and n_waves != int(waves):
        # measured for another launch shape (--seqs / --ctxs changed the batch): the count scales with the waves, every wave tracks one keypoint
        insts = insts * n_waves / waves
        src += "; scaled from %d to %d waves per launch" % (int(waves), n_waves)
    rate = insts / launch_s
    peak = VALU_ISSUE_PER_CLK_PER_SIMD * N_SIMDS * CLK_HZ
    return {"wave_insts_per_launch": int(insts), "wave_insts_per_wave": round(insts / max(n_waves, 1), 1),
            "achieved_ginst_s": round(rate / 1e9, 2), "peak_ginst_s": round(peak / 1e9, 2), "frac": round(rate / peak, 4),
            "cycles_per_inst_per_simd": round(N_SIMDS * CLK_HZ * launch_s / insts, 3),
            "peak_note": "1 wave-instruction / 2 clk / SIMD x 1024 SIMDs x 2.4 GHz (full-rate 32-bit ops; the chip clocks lower under load)",
            # a stream that MIXES full-rate ops with dot / perm / DPP / packed ops issues at the slow class's rate throughout
            # (profiles/r02_issue_probe.txt: "klt mix" and "add/dot2c alternating" 4.0-4.2 clk per instruction at 2.37 GHz)
            "mixed_stream_clk_per_inst": MIXED_CLK_PER_INST,
            "frac_of_mixed_stream_roof": round(MIXED_CLK_PER_INST / (N_SIMDS * MIXED_CLK_HZ * launch_s / insts), 4),
            "mixed_note": "measured issue rate of this instruction mix (tools/issue_probe.hip) at the clock the probe ran at (2.37 GHz)",
            "source": src}


LINE_MAX_BYTES = 4096       # the printed line stays below this (the round-5 line grew to 30 KB and the driver could not read it)


def _strict(v):
    """NaN / inf -> None, numpy scalars -> Python numbers, recursively: the line must load with a strict JSON parser"""
    if isinstance(v, dict):
        return {str(k): _strict(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_strict(x) for x in v]
    if isinstance(v, (np.floating, float)):
        v = float(v)
        return v if np.isfinite(v) else None
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.bool_,)):
        return bool(v)
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _short_kernel(name):
    """`void k_ba_build_w<4, 2, 5>(ba_args)` -> `k_ba_build_w<4,2,5>`"""
    n = str(name).replace("void ", "").split("(")[0].replace(", ", ",").strip()
    return n[:48]


def compact_line(full, extras_file=None):
    """The ONE line the driver reads: the contract's keys + `roofline` + `cpu_baseline` + the second figures, below LINE_MAX_BYTES,
    strict JSON.  Everything else of `full` lives in the side file named by `extras_file`.  Pure function (tests/test_bench_line.py
    builds the line from a committed full result)."""
    full = _strict(full)
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = full.get("vs_baseline")
    line.update(_pick(full, ("dtype", "data")))
    cfg = full.get("config") or {}
    c = _pick(cfg, ("workload", "sequences_per_gpu", "batched_contexts_per_gpu", "frames_per_step", "distinct_image_sequences", "frames_source", "tracked_points", "ba_lm_iteration_budget",
                    "ba_lm_iterations_run", "stream_layout", "launch"))
    if "ba_solves_stopped_by_lm_max_iters" in cfg:
        c["solves_stopped_by_cap"] = cfg["ba_solves_stopped_by_lm_max_iters"]
    if "ba_solves_cut_by_the_budget_and_rerun_in_the_timed_region" in cfg:
        c["solves_cut_and_rerun"] = cfg["ba_solves_cut_by_the_budget_and_rerun_in_the_timed_region"]
    if "parallelism" in cfg:
        c["parallelism"] = str(cfg["parallelism"])[:120]
    line["config"] = c
    reg = full.get("regions") or {}
    if reg:
        line["regions"] = _pick(reg, ("n", "frames_per_s_min", "frames_per_s_max"))
    roof = full.get("roofline")
    if isinstance(roof, dict):
        r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac"))
        r["traffic"] = roof.get("traffic")
        r.update(_pick(roof, ("avg_launch_us", "algorithmic_bytes_per_launch", "compute_units_of_the_launch")))
        if roof.get("traffic_source"):
            r["traffic_source"] = str(roof["traffic_source"])[:80]
        valu = roof.get("valu") or {}
        if valu.get("frac") is not None:
            r["valu"] = _pick(valu, ("frac", "frac_of_mixed_stream_roof", "wave_insts_per_wave"))
        issue = roof.get("issue") or {}
        if issue.get("frac_of_quarter_rate_capacity") is not None:
            r["issue"] = _pick(issue, ("frac_of_quarter_rate_capacity", "frac_at_2.07_ghz", "hbm_gb_s"))
        ks = (roof.get("kernels") or {})
        if isinstance(ks, dict) and ks.get("kernels"):
            top = sorted(ks["kernels"], key=lambda e: -float(e.get("pct_of_kernel_time") or 0))[:3]
            r["kernels"] = [dict(kernel=_short_kernel(e.get("kernel")), pct=e.get("pct_of_kernel_time"), us=e.get("one_stream_avg_launch_us", e.get("avg_launch_us")),
                                 hbm_frac=e.get("hbm_frac"), valu_issue_frac=e.get("valu_issue_frac"), mfma_util_pct=e.get("mfma_util_pct_mean")) for e in top]
            r["kernels_source"] = ("STALE " if ks.get("stale") else "") + str(ks.get("source", ""))[:60]
        line["roofline"] = r
    else:
        line["roofline"] = None
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        cb = _pick(cpu, ("value", "unit", "cores", "kind", "per_core"))
        if cpu.get("sample"):
            cb["sample"] = str(cpu["sample"])[:200]
        rr = cpu.get("reference_recipe_ba")
        if isinstance(rr, dict) and rr.get("seconds_per_adjust") is not None:
            cb["reference_recipe_ba_seconds_per_adjust"] = rr["seconds_per_adjust"]
        line["cpu_baseline"] = cb
    else:
        line["cpu_baseline"] = None
    for k in ("resident_frames", "host_frames", "shi_tomasi_seeded_points", "closed_loop_w10_256", "closed_loop_w10_uncapped_tables"):
        if isinstance(full.get(k), dict):
            line[k] = {kk: v for kk, v in full[k].items() if kk not in ("what", "cmd", "seconds", "frames_counted")}      # (prose stays in the side file)
    if extras_file:
        line["extras_file"] = extras_file
    # never above the limit: shed the optional parts, least important first
    for drop in (("shi_tomasi_seeded_points",), ("closed_loop_w10_uncapped_tables",), ("roofline", "kernels"), ("cpu_baseline", "sample"), ("closed_loop_w10_256",), ("host_frames",),
                 ("regions",), ("roofline", "traffic_source"), ("config", "parallelism")):
        if len(json.dumps(line, separators=(",", ":"), allow_nan=False)) < LINE_MAX_BYTES:
            break
        if len(drop) == 1:
            line.pop(drop[0], None)
        elif isinstance(line.get(drop[0]), dict):
            line[drop[0]].pop(drop[1], None)
    return json.dumps(line, separators=(",", ":"), allow_nan=False)


def emit(full, a):
    """full result object -> side file(s); the compact line (or, --full-line, the object itself) -> the LAST line of stdout"""
    full = _strict(full)
    targets = [a.extras_file] if a.extras_file else [os.path.join(ROOT, "bench_extras.json")] + \
        ([os.path.join(ROOT, "gpurun_out", "bench_extras.json")] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])
    written = None
    for t in targets:
        try:
            with open(t, "w") as f:
                json.dump(full, f, indent=1, allow_nan=False)
            written = written or os.path.relpath(t, ROOT)
        except Exception as e:          # noqa: BLE001  (a read-only tree must not cost the line)
            sys.stderr.write("bench: could not write %s: %s\n" % (t, e))
    sys.stdout.flush()
    print(json.dumps(full, allow_nan=False) if a.full_line else compact_line(full, written), flush=True)


def self_launch(a):
    """`bench.py --gpus N` outside a torch.distributed launch: start the N ranks as a CHILD process tree (one process per GPU,
    the command the driver itself uses) and relay rank 0's JSON line.  This process never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line)
    else:
        sys.stderr.write(r.stdout[-2000:])
    sys.exit(r.returncode if r.returncode else (0 if line is not None else 1))


def main():
    global W_IMG, H_IMG, N_PTS, N_NEW, BA_N, BA_W, WORKLOAD, K_CAM
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and a.cpu_worker < 0:
        self_launch(a)
    if a.dry_run:                  # launcher / rendezvous plumbing only (tests/test_dist_cpu.py): no GPU library, no kernels
        d = Dist()
        d.barrier()
        tot = d.sum(1.0)
        if d.rank == 0:
            print(json.dumps({"metric": "dry-run", "n_gpus": d.world, "ranks_seen": int(tot), "gpus_arg": a.gpus}))
        d.close()
        return
    if a.cpu_worker == 10_000:     # CPU-baseline side job: the reference's solver recipe + (if present) live OpenCV timings
        from vo_mi355x import synthetic as syn
        frames = syn.make_sequence(2, W_IMG, H_IMG, seed=1234)[0]
        print(json.dumps({"reference_recipe_ba": reference_recipe_ba_seconds(), "opencv": opencv_baseline(frames)}))
        return
    if a.cpu_pipe_worker >= 0:     # CPU-baseline worker of the closed loop: never touches the GPU library
        cpu_pipeline_worker(a)
        return
    if a.cpu_worker >= 0:          # CPU-baseline worker process: never touches the GPU library
        from vo_mi355x import synthetic as syn
        frames = syn.make_sequence(a.frames, W_IMG, H_IMG, seed=1234 + a.cpu_worker, periodic=True, n_render=a.cpu_frames + 2)[0]
        cpu_baseline(frames, 1, a.ba_iters)                    # page in the oracle, first-call costs
        v, secs = cpu_baseline(frames, a.cpu_frames, a.ba_iters)
        print(json.dumps({"frames": a.cpu_frames, "seconds": secs}))
        return
    dist = Dist()
    cpu = None
    if dist.rank == 0 and dist.world == 1 and not a.no_cpu_baseline and a.workload == "A" and a.cpu_worker < 0:
        # before anything initialises the GPU in this process (child processes are started here)
        n_procs = max(1, min(a.cpu_procs, usable_cores()))
        v, secs, cores, extra = cpu_baseline_parallel(n_procs, a.cpu_frames, a.ba_iters)
        cpu = {"value": round(v, 3), "unit": "frames/s", "cores": cores, "cores_available": usable_cores(), "cores_visible": os.cpu_count() or 0, "kind": "port",
               "per_core": round(v / max(cores, 1), 3),
               "reference_recipe_ba": extra.get("reference_recipe_ba"), "opencv": extra.get("opencv"),
               "sample": "%d worker processes (1 core each, independent sequences) x %d frames of the same workload on the CPU "
                         "oracle (C: pyramid+KLT+Shi-Tomasi+DLT, numpy: BA), %.1f s wall; %.2f frames/s per core; host shows %d cores, "
                         "its cgroup grants %d" % (cores, a.cpu_frames, secs, v / max(cores, 1), os.cpu_count() or 0, usable_cores())}
    cpu_pipe = None
    if dist.rank == 0 and dist.world == 1 and not a.no_cpu_baseline and (a.workload == "pipeline" or (a.workload == "A" and a.extras)):
        # (child processes: before anything initialises the GPU here.  In the default run it is the figure beside the informational
        #  `pipeline_step.reference_configuration_window4` entry: the reference's own window, 2 048-slot tables)
        ap = a
        if a.workload == "A":
            import copy as _c
            ap = _c.copy(a); ap.pipe_window, ap.pipe_no_resurrect = 4, False
        cpu_pipe = cpu_pipeline_baseline(ap, max(1, min(a.cpu_procs, usable_cores())))
    from vo_mi355x import VoContext, synthetic as syn
    if a.tune:
        VoContext.default_tuning = {k.strip(): int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv.strip())}
    # more stepping host threads on this node than cores it grants (8 ranks x 3 threads on a 16-core cgroup): wait for a step's event
    # in the driver instead of spinning on it (1 GPU: 34 650 vs 34 640 frames/s, two ranks on one GPU 31 580 vs 31 190 -- no loss)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if a.ctxs is None:
        a.ctxs = 1
    if a.side_stream is None:
        a.side_stream = "pipeline" if (a.workload == "A" and a.ctxs == 1) else "on"
    if "VO_BLOCKING_SYNC" not in os.environ and local_world * max(a.host_threads, 1) > usable_cores():
        os.environ["VO_BLOCKING_SYNC"] = "1"
    t_gen = time.perf_counter()
    c5 = a.workload == "config5"
    if a.workload == "pipeline":
        # the closed loop is its own harness (no resident BA problem, no uploaded point set: everything comes out of the device tables)
        a.ctxs = max(1, min(a.ctxs, a.seqs))
        scenes = pipe_scenes(2, a.pipe_frames, 4321 + 16 * dist.rank)
        r = run_pipeline(dist.local_rank, a, dist, a.ctxs, max(1, a.seqs // a.ctxs), a.steps, a.warmup, a.regions, scenes)
        tot = dist.sum(float(r["sequences"]))
        second = None
        if a.pipe_second_max_pts > 0:
            import copy as _c
            a2 = _c.copy(a); a2.pipe_max_pts = a.pipe_second_max_pts
            r2 = run_pipeline(dist.local_rank, a2, dist, a.ctxs, max(1, a.seqs // a.ctxs), a.steps, a.pipe_second_warmup, a.regions, scenes)
            tot2 = dist.sum(float(r2["sequences"]))
            second = {"value": round(tot2 * a.steps / (r2["ms_per_step"] * 1e-3 * a.steps), 2), "ms_per_step": r2["ms_per_step"], "roofline": r2.pop("roofline", None), "pipeline": r2}
        if dist.rank == 0:
            fps = tot * a.steps / (r["ms_per_step"] * 1e-3 * a.steps)
            emit({"metric": "frames/sec, Pipeline.step resident on the device @1241x376 (<= %d tracked keypoints, RANSAC-P3P pose, triangulation, "
                            "%d-frame BA, re-detection; closed loop)" % (a.pipe_max_pts, a.pipe_window), "value": round(fps, 2), "unit": "frames/s", "n_gpus": dist.world,
                  "steps": a.steps, "warmup": a.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                  "vs_baseline": None, "dtype": "u8/i32 (KLT, Shi-Tomasi) + f64 (P3P, DLT, BA)", "data": "synthetic (rendered two-plane scene, known trajectory)",
                  "config": {"workload": "pipeline_step_closed_loop_1241x376_ba%d" % a.pipe_window, "sequences_per_gpu": r["sequences"],
                             "batched_contexts_per_gpu": r["contexts"], "parallelism": "independent sequences, no collective"},
                  "pipeline": r, "roofline": r.pop("roofline", None), "cpu_baseline": cpu_pipe, "second": second}, a)
        dist.close()
        return
    if c5:
        # ONE sequence over all ranks: every rank runs the (launch-bound) front end on the whole frame redundantly and
        # owns 1/n_ranks of the landmarks of the 20-frame bundle adjustment
        W_IMG, H_IMG, N_PTS, N_NEW, BA_N, BA_W = 1920, 1080, 5000, 1000, 5000, 20
        WORKLOAD = "single_seq_1920x1080_5000pts_ba20_landmark_sharded"
        K_CAM = np.array([[1100.0, 0, 960.0], [0, 1100.0, 540.0], [0, 0, 1]])
        a.seqs, a.ctxs, a.host_threads, a.fixed_ba_budget, a.no_cpu_baseline = 1, 1, 1, True, True
        uid = dist.bcast_bytes(VoContext.comm_unique_id() if dist.rank == 0 else np.zeros(128, np.uint8))
        frame_sets = [syn.make_sequence(a.frames, W_IMG, H_IMG, seed=1234, margin=96, periodic=True)[0]]
        seqs = [Group(dist.local_rank, frame_sets, seed0=0, batch=1, ba_iters=a.ba_iters, shard=(dist.rank, dist.world, uid))]
    else:
        a.ctxs = max(1, min(a.ctxs, a.seqs))
        per = [a.seqs // a.ctxs + (1 if i < a.seqs % a.ctxs else 0) for i in range(a.ctxs)]
        # distinct image sequences behind the batch: 8 when they can be rendered side by side (one core each), 4 otherwise (rounds 1-5); sequence b
        # of a context tracks its own point set on image sequence b % n
        n_distinct = min(a.distinct, a.seqs) if a.distinct > 0 else min(8 if (dist.world == 1 and usable_cores() >= 8) else 4, a.seqs)
        frame_sets = render_sequences([1234 + 16 * dist.rank + k for k in range(n_distinct)], a.frames, parallel=dist.world == 1)
        seqs = [Group(dist.local_rank, frame_sets, seed0=1000 * dist.rank + 100 * i, batch=per[i], ba_iters=a.ba_iters)
                for i in range(a.ctxs)]
    t_setup = time.perf_counter() - t_gen
    side = {"on": 1, "off": 0, "pipeline": 2}[a.side_stream]
    for s in seqs:
        s.c.set_side_stream(side)
        s.c.set_graph_mode(bool(a.graph))
        s.max_inflight = 2
        s.adaptive = not a.fixed_ba_budget

    pool = None
    a.host_threads = min(a.host_threads, a.ctxs)
    if a.host_threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(a.host_threads)
        chunks = [seqs[i::a.host_threads] for i in range(a.host_threads)]

        def run_chunk(ch):
            for s in ch:
                s.step()

        def drain_chunk(ch):
            for s in ch:
                s.drain()

    def step():
        if pool is not None:
            list(pool.map(run_chunk, chunks))     # ctypes releases the GIL inside the C calls
            return
        for s in seqs:
            s.step()

    def drain():
        if pool is not None:
            list(pool.map(drain_chunk, chunks))
            return
        for s in seqs:
            s.drain()

    for _ in range(a.warmup):
        step()
    drain()
    # ---- timed regions: exactly K steps each, barrier + sync on both sides; KLT kernel bracketed by hipEvents on its stream ----
    for s in seqs:
        s.c.profile_enable((s.c.PROF_KLT,))
        s.c.sync()
        s.truncated = s.at_cap = 0
        s.iter_hist, s.groups_enq, s.groups_needed, s.n_steps = {}, 0, 0, 0
    region_dt = []
    for _ in range(max(1, a.regions)):
        for s in seqs:
            s.c.sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        drain()                                    # every one of the K frames is complete and fetched inside the timed region
        for s in seqs:
            s.c.sync()
        dist.barrier()
        region_dt.append(dist.max(time.perf_counter() - t0))
    dt = float(np.median(region_dt))               # an odd --regions makes this one actual region
    klt_ms, klt_n = 0.0, 0
    for s in seqs:
        ms, n = s.c.profile_read(s.c.PROF_KLT)
        klt_ms += ms
        klt_n += n
        s.c.profile_enable(())
    n_trunc = int(dist.sum(float(sum(s.truncated for s in seqs))))
    n_at_cap = int(dist.sum(float(sum(s.at_cap for s in seqs))))
    iter_hist = {}
    for s in seqs:
        for k, v in s.iter_hist.items():
            iter_hist[k] = iter_hist.get(k, 0) + v
    groups_enq = sum(s.groups_enq for s in seqs) / max(1, sum(s.n_steps for s in seqs))
    groups_need = sum(s.groups_needed for s in seqs) / max(1, sum(s.n_steps for s in seqs))
    # the same K steps once more with the other budget policy (one region, informational): what the adaptive budget is worth
    other_fps = stationary_fps = None
    if not c5:
        for s in seqs:
            s.adaptive = not s.adaptive
            s.ba_prm.max_iters = s.ba_iters_cap
        for _ in range(5):
            step()
        drain()
        for s in seqs:
            s.c.sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        drain()
        for s in seqs:
            s.c.sync()
        dist.barrier()
        other_fps = dist.sum(float(a.seqs)) * a.steps / dist.max(time.perf_counter() - t0)
        for s in seqs:
            s.adaptive = not s.adaptive
            s.stationary = True            # ... and with ONE problem per sequence solved every frame, as rounds 1-2 measured
            s.recent = []
        for _ in range(8):
            step()
        drain()
        for s in seqs:
            s.c.sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        drain()
        for s in seqs:
            s.c.sync()
        dist.barrier()
        stationary_fps = dist.sum(float(a.seqs)) * a.steps / dist.max(time.perf_counter() - t0)
        for s in seqs:
            s.stationary = False
            s.recent = []
    # ---- the same step with every frame handed over by the host (numpy arrays in page-locked memory -> vo_frame_step_host: upload on a copy stream
    # beside the previous frame's LM chain): 3 regions of exactly K steps, same barriers ----
    host_fig = None
    if not c5 and not a.no_extras and not a.graph:
        try:
            h2d_bytes = dist.sum(float(sum(s.use_host_frames(True) for s in seqs)))      # all ranks: every GPU has its own link to the host
            for _ in range(max(4, a.warmup)):
                step()
            drain()
            host_dt = []
            for _ in range(3):
                for s in seqs:
                    s.c.sync()
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    step()
                drain()
                for s in seqs:
                    s.c.sync()
                dist.barrier()
                host_dt.append(dist.max(time.perf_counter() - t0))
            hdt = float(np.median(host_dt))
            hfps = dist.sum(float(a.seqs)) * a.steps / hdt
            host_fig = {"value": round(hfps, 2), "unit": "frames/s", "ms_per_step": round(hdt / a.steps * 1e3, 4),
                        "h2d_mb_per_step": round(h2d_bytes / 1e6, 1), "h2d_gb_s": round(h2d_bytes * a.steps / hdt / 1e9, 2),
                        "vs_resident": round(hfps / (dist.sum(float(a.seqs)) * a.steps / dt), 4),
                        "what": "every step's images from page-locked host arrays through vo_frame_step_host (one array per sequence), copy stream"}
        except Exception as e:      # noqa: BLE001
            host_fig = {"error": repr(e)[:200]}
        for s in seqs:
            s.use_host_frames(False)
    frames_step = 1.0 if c5 else dist.sum(float(a.seqs))                             # config 5: ONE sequence on all ranks
    frames_total = frames_step * a.steps
    fps = frames_total / dt

    out = None
    if dist.rank == 0:
        # ---- untimed extras on rank 0: stage breakdown, iteration counts, parity spot check inputs ----
        s0 = seqs[0]
        s0.c.profile_enable((0, 1, 2, 3, 4))
        reps = 20
        for _ in range(reps):
            s0.enqueue()
            s0.fetch()
        stage = {}
        for name, r in (("pyramid_scharr", 0), ("klt", 1), ("shi_tomasi", 2), ("dlt", 3), ("ba", 4)):
            ms, n = s0.c.profile_read(r)
            stage[name] = round(ms / max(n, 1), 4)
        s0.c.profile_enable(())
        pts, st_, err_, it = s0.c.points_download(N_PTS, return_iters=True)
        ba_stats = s0.ba_stats0()
        it = it.reshape(-1, it.shape[-1])
        it_mean = [float(np.maximum(it[:, l], 0).mean()) for l in range(it.shape[1])]
        # ALGORITHMIC bytes of one KLT launch: sum over the tracked points and levels of 5120 + 1024 * it_l
        # (one launch tracks the whole batch; with the track table only the live slots count)
        klt_bytes = N_PTS * s0.B * sum(5120.0 + 1024.0 * x for x in it_mean)
        klt_avg_s = (klt_ms / max(klt_n, 1)) * 1e-3
        achieved = klt_bytes / klt_avg_s / 1e9 if klt_avg_s > 0 else 0.0
        traffic, traffic_src = profile_constant("klt_traffic.json", "hbm_bytes_per_launch")
        # (the counters were averaged over launches of `sq_waves_per_launch` keypoints -- the default command's; another --seqs scales the figure)
        pw, _ = profile_constant("klt_valu.json", "sq_waves_per_launch")
        if traffic is not None and pw and int(pw) != s0.B * N_PTS:
            traffic = int(traffic * (s0.B * N_PTS) / float(pw))
            traffic_src += "; scaled from %d to %d keypoints per launch" % (int(pw), s0.B * N_PTS)
        valu = valu_roofline(klt_avg_s, s0.B * N_PTS)
        roof = {"bound": "hbm", "kernel": "k_klt_track", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src, "valu": valu,
                "avg_launch_us": round(klt_avg_s * 1e6, 3), "algorithmic_bytes_per_launch": int(klt_bytes),
                "klt_mean_iters_per_level": [round(x, 3) for x in it_mean],
                "kernels": kernel_rooflines(),
                "issue": None,      # filled below: the whole step against the chip's vector-issue capacity
                # the same launch when context 0 runs its steps alone after the timed region (no other context's kernels beside it)
                "alone_avg_launch_us": round(stage["klt"] * 1e3, 3),
                "alone_frac": round(klt_bytes / max(stage["klt"] * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 5),
                "compute_units_of_the_launch": 256 - s0.c.step_layout()["reserved_cus"],
                "note": "HBM figure = algorithmic bytes / launch time as the contract defines it; the kernel itself is vector-ALU bound "
                        "(see `valu`: instructions issued per launch against the SIMDs' issue rate) and moves `traffic` bytes per launch through HBM. "
                        "`frac` is the launch as timed INSIDE the run: with several batched contexts it shares the vector ALUs with the "
                        "Shi-Tomasi / BA kernels of the other contexts and takes longer than on its own (`alone_*`).  In the pipelined layout of a batch "
                        "the launch is confined to `compute_units_of_the_launch` of the chip's 256 CUs (the others carry the previous frame's LM tail "
                        "groups beside it, include/vo_mi355x.h: vo_set_side_stream): `frac` and `valu.frac` are taken against the WHOLE chip's peaks, "
                        "so 224 CUs at the rate 256 reached before read 0.875 x as much"}
        if roof["kernels"].get("issue_slots_per_step") and not c5 and s0.B * len(seqs) == 256:
            slots = roof["kernels"]["issue_slots_per_step"]
            step_s = dt / a.steps
            roof["issue"] = {"issue_slots_per_step": slots, "step_ms": round(step_s * 1e3, 4),
                             "frac_of_quarter_rate_capacity": round(slots * MIXED_CLK_PER_INST / (N_SIMDS * MIXED_CLK_HZ * step_s), 4),
                             "what": "vector wave-instructions + 14 per float64 MFMA of ALL kernels of a 256-sequence step (rocprofv3 counters of the default command, "
                                     "profiles/kernel_counters.json) x 4.1 clk (the rate a stream of quarter-rate instructions issues at, profiles/r02_issue_probe.txt) "
                                     "/ (1024 SIMDs x 2.37 GHz x the step time measured in this run): how much of the chip's issue capacity the step uses -- the "
                                     "bound of the whole workload (at the ~2.07 GHz the chip sustains under it, profiles/r05_mfma_overlap_probe.txt, the same count "
                                     "is 1.14 x this fraction); HBM: `hbm_gb_s` over the step",
                             "frac_at_2.07_ghz": round(slots * MIXED_CLK_PER_INST / (N_SIMDS * 2.07e9 * step_s), 4),
                             "hbm_gb_s": round(roof["kernels"].get("hbm_bytes_per_step", 0) / step_s / 1e9, 1)}
        out = {"metric": ("frames/sec @1920x1080, 5000 KLT pts, 20-frame sharded BA (config 5)" if c5 else
                          "frames/sec @1241x376, 2000 KLT pts, 10-frame BA window"), "value": round(fps, 2),
               "unit": "frames/s", "n_gpus": dist.world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if c5 else "weak",
               "vs_baseline": None, "dtype": "u8/i32 (KLT, Shi-Tomasi) + f64 (DLT, BA)", "data": "synthetic",
               "config": {"workload": WORKLOAD, "width": W_IMG, "height": H_IMG,
                          "klt_points": N_PTS, "klt_win": 31, "klt_levels": 4, "dlt_points": N_NEW,
                          "ba_landmarks": BA_N, "ba_window": BA_W, "ba_observations": BA_N * BA_W,
                          "ba_lm_iteration_budget": a.ba_iters, "ba_lm_iteration_budget_note": "LM cap per adjust; rounds 1-4 ran 10 (1 % of the solves cut there): `lm_cap_10` of this line is that configuration on this build",
                          "ba_lm_iterations_run": ba_stats["iters"],
                          "ba_final_cost": round(ba_stats["cost"], 4), "ba_initial_cost": round(ba_stats["cost0"], 2),
                          "launch": "hipGraph replay" if a.graph else "plain",
                          "ba_problems_per_sequence": seqs[0].n_ba, "frames_per_sequence": a.frames,
                          "ba_budget": "fixed (--ba-iters groups every frame)" if a.fixed_ba_budget else
                                       "adaptive (1 + the most any of the last 16 fetched frames needed, over the batch)",
                          "ba_lm_iterations_histogram": {str(k): iter_hist[k] for k in sorted(iter_hist)},
                          "ba_iteration_groups_enqueued_per_step": round(groups_enq, 3), "ba_iteration_groups_needed_per_step": round(groups_need, 3),
                          "ba_solves_cut_by_the_budget_and_rerun_in_the_timed_region": n_trunc, "ba_solves_stopped_by_lm_max_iters": n_at_cap,
                          ("adaptive_budget_frames_per_s" if a.fixed_ba_budget else "fixed_budget_frames_per_s"): None if other_fps is None else round(other_fps, 1),
                          "one_stationary_ba_problem_frames_per_s": None if stationary_fps is None else round(stationary_fps, 1), "side_stream": a.side_stream, "stream_layout": seqs[0].c.step_layout(), "host_threads": max(a.host_threads, 1),
                          "host_wait": "blocking" if os.environ.get("VO_BLOCKING_SYNC", "0") not in ("", "0") else "spin",
                          "sequences_per_gpu": a.seqs, "batched_contexts_per_gpu": a.ctxs,
                          "frames_per_step": 1 if c5 else a.seqs * dist.world,
                          "parallelism": ("one sequence, BA landmarks sharded over %d GPU(s), RCCL all-reduce of the reduced camera "
                                          "packet + 4 statistics per LM iteration, front end replicated" % dist.world) if c5 else
                                         ("independent sequences, %d per GPU in %d batched context(s) x %d GPU(s), no collective"
                                          % (a.seqs, a.ctxs, dist.world))},
               "regions": {"n": len(region_dt), "frames_per_s_min": round(frames_total / max(region_dt), 2),
                           "frames_per_s_median": round(fps, 2), "frames_per_s_max": round(frames_total / min(region_dt), 2),
                           "ms_per_step_each": [round(x / a.steps * 1e3, 4) for x in region_dt]},
               "stage_ms_per_batched_launch_group": stage, "roofline": roof, "cpu_baseline": cpu,
               "setup_s": round(t_setup, 2)}
        out["config"]["frames_source"] = "resident in HBM (vo_seq_upload); `host_frames`: the same with every frame handed over by the host"
        out["config"]["distinct_image_sequences"] = len(frame_sets)
        out["config"]["tracked_points"] = "jittered grid (SURVEY 8d's fallback); `shi_tomasi_seeded_points`: the build's own corners of frame 0"
        if host_fig is not None:
            out["host_frames"] = host_fig
    # ---- the tracked points seeded by the build's OWN Shi-Tomasi on frame 0 (SURVEY 8d's first choice; the headline tracks its fallback, a jittered
    # grid): 2 000 corners per distinct sequence at the reference's quality level and minimum distance, 3 regions of K steps ----
    if out is not None and dist.world == 1 and a.workload == "A" and not a.no_extras and not c5:
        try:
            out["shi_tomasi_seeded_points"] = seeded_points_figure(dist.local_rank, seqs, frame_sets, a, step, drain)
        except Exception as e:      # noqa: BLE001
            out["shi_tomasi_seeded_points"] = {"error": repr(e)[:200]}
    dist.barrier()
    for s in seqs:
        s.c.close()
    try:
        if out is not None and dist.world == 1 and a.workload == "A" and not a.no_extras:
            # second figure of the line: the coupled loop (Pipeline.step resident on the device) at BASELINE's window and the headline's batch, as a
            # CHILD with a timeout -- whatever it does, the headline built above is printed
            # ... and, in the same child (same scenes), with tables the scene does not fill (8 192 slots per sequence; ~4 200 keypoints tracked): no
            # frame of it is shaped by the capacity policy (`capacity_policy_frames` 0) -- the reference's unbounded lists at the headline's batch
            out["closed_loop_w10_256"], out["closed_loop_w10_uncapped_tables"] = closed_loop_child(
                a, ["--pipe-window", "10", "--pipe-no-resurrect", "--seqs", str(a.seqs), "--pipe-second-max-pts", "8192"], timeout=240)
        if out is not None and dist.world == 1 and a.workload == "A" and a.extras:
            out.update(measure_extras(dist.local_rank, frame_sets, a, dist, cpu_pipe))
    except Exception as e:          # noqa: BLE001  (informational keys must never cost the bench line)
        out["extras_error"] = repr(e)
    finally:
        dist.close()
        if out is not None:
            emit(out, a)


if __name__ == "__main__":
    main()
