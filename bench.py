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
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--seqs", type=int, default=96, help="independent sequences per GPU")
    ap.add_argument("--ctxs", type=int, default=3, help="batched contexts (HIP streams) the sequences are split over")
    ap.add_argument("--ba-iters", type=int, default=10, help="LM iteration budget per adjust")
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames per sequence (played ping-pong)")
    ap.add_argument("--graph", action="store_true", help="replay each frame from a captured hipGraph instead of plain launches")
    ap.add_argument("--host-threads", type=int, default=3, help="enqueue/fetch the contexts from this many host threads")
    ap.add_argument("--fixed-ba-budget", action="store_true",
                    help="always enqueue --ba-iters LM iterations (default: last frame's iteration count + 2, capped)")
    ap.add_argument("--workload", choices=("A", "config5", "pipeline"), default="A",
                    help="A: BASELINE configs[2], the metric's configuration (default).  config5: ONE 1920x1080 sequence, 5000 "
                         "points, 20-frame BA whose landmarks are sharded over the ranks with an RCCL all-reduce per LM "
                         "iteration (front end replicated); strong scaling, not the headline metric.  pipeline: workload A plus the "
                         "steps of Pipeline.step around it on the device -- track table (KLT + pruning + history + re-detection "
                         "spawn) and RANSAC-P3P pose -- issued as separate calls with one sync per frame; informational")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames each CPU-baseline worker process runs")
    ap.add_argument("--cpu-procs", type=int, default=16, help="CPU-baseline worker processes (one core each), capped by the host's cores")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)     # internal: run as CPU-baseline worker with this seed
    return ap.parse_args()


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


class Group:
    """`batch` independent VO sequences carried in lockstep by ONE batched context (one HIP stream): every launch of
    the hot path serves all of them.  Everything is resident in HBM."""

    def __init__(self, device, frame_sets, seed0, batch, ba_iters, shard=None, pipeline=False):
        """shard = (rank, n_ranks, unique_id): config 5 -- this context holds landmark shard `rank` of ONE BA problem"""
        from vo_mi355x import VoContext, sharding, synthetic as syn
        self.B = batch
        self.c = VoContext(W_IMG, H_IMG, max_pts=max(N_PTS, N_NEW), device=device, batch=batch)
        c = self.c
        c.upload_sequence(np.stack([frame_sets[b % len(frame_sets)] for b in range(batch)]))
        self.nf = frame_sets[0].shape[0]
        self.pipeline = pipeline
        pts0 = np.stack([syn.grid_points(N_PTS, W_IMG, H_IMG, seed=seed0 + b) for b in range(batch)])
        if not pipeline:
            c.points_upload(pts0)
        # DLT: 1000 new tracks between two window poses of each BA scene; BA: N = 2000, W = 10 per sequence
        kw = {} if K_CAM is None else dict(K=K_CAM, width=W_IMG, height=H_IMG)
        scenes = [syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=seed0 + b, **kw) for b in range(batch)]
        P0s, P1s, u0, u1, Ks, H0s, H1s = [], [], [], [], [], [], []
        for s in scenes:
            K = s["K"]
            H0, H1 = np.eye(4), np.eye(4)
            H0[:3, :3], H0[:3, 3] = syn.rodrigues(s["poses_gt"][3, :3]), s["poses_gt"][3, 3:]
            H1[:3, :3], H1[:3, 3] = syn.rodrigues(s["poses_gt"][0, :3]), s["poses_gt"][0, 3:]
            P0s.append((K @ H0[:3]).astype(np.float32)); P1s.append((K @ H1[:3]).astype(np.float32))
            u0.append(s["obs"][3, :N_NEW].astype(np.float32)); u1.append(s["obs"][0, :N_NEW].astype(np.float32))
            Ks.append(K); H0s.append(H0); H1s.append(H1)
        c.dlt_upload(np.stack(P0s), np.stack(P1s), np.stack(u0), np.stack(u1), np.stack(Ks), np.stack(H0s), np.stack(H1s))
        if shard is None:
            c.ba_upload(np.stack(Ks), np.stack([s["poses0"] for s in scenes]), np.stack([s["points0"] for s in scenes]),
                        np.stack([s["obs"] for s in scenes]))
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
        if pipeline:
            # track table seeded with the keypoints; 3D-2D correspondences of the newest BA frame with 30 % gross outliers
            c.tracks_seed(pts0, t=0)
            rng = np.random.default_rng(seed0)
            X = np.stack([s["points_gt"] for s in scenes]).astype(np.float32)
            uv = np.stack([s["obs"][0] for s in scenes]).astype(np.float32)
            for b in range(batch):
                out = rng.choice(BA_N, int(0.3 * BA_N), replace=False)
                uv[b, out] += rng.uniform(-80, 80, (len(out), 2)).astype(np.float32) + np.float32(15)
            c.pnp_upload(np.stack(Ks), X, uv)
            self.pnp_prm = c.pnp_params(reproj_err=2.0, confidence=0.9999, max_iters=1000000, seed=seed0)
        self.t = 1
        self.inflight = 0
        self.max_inflight = 2          # 1 when the step is replayed from a hipGraph (its host destinations are baked in)

    def enqueue(self):
        if self.pipeline:
            # Pipeline.step (reference pipeline.py:92-167) as separate device calls: frame -> extend tracks -> 3D-2D pose ->
            # triangulate -> bundle adjust -> re-detect
            c, t = self.c, self.t
            c.push_frame_resident(pingpong(t, self.nf))
            c.tracks_track(t, self.klt_prm)
            c.pnp_solve_resident(self.pnp_prm, 2)
            c.dlt_resident()
            c.ba_solve_resident(self.ba_prm)
            c.tracks_detect(t, 7, self.st_prm, max_new=1000)
            self.t += 1
            self.inflight += 1
            return
        # one C call: pyramid + KLT + DLT + BA + Shi-Tomasi + result copies for the whole batch
        self.c.frame_step_resident(pingpong(self.t, self.nf), N_PTS, True, True, True, 7, self.klt_prm, self.st_prm,
                                   self.ba_prm)
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
        if self.pipeline:
            rv, tv, inl, pst = self.c.pnp_fetch()                  # waits for the frame
            po, pt, bst = self.c.ba_fetch()
            tr = self.c.tracks_read()
            tr = tr if isinstance(tr, list) else [tr]
            self.last = {"ba_stats": bst, "pnp_stats": pst, "n_tracks": [len(x["tag"]) for x in tr]}
        else:
            self.last = self.c.frame_fetch()
        self.inflight -= 1
        if self.adaptive:
            # the LM stops by its own ftol/xtol tests; the budget only bounds how many (early-exiting) launches are
            # enqueued blindly.  Next frame: what this frame needed (max over the batch) + 2, never more than --ba-iters.
            st = self.last["ba_stats"]
            its = max(x["iters"] for x in st) if isinstance(st, list) else st["iters"]
            self.ba_prm.max_iters = max(3, min(self.ba_iters_cap, its + 2))
        return self.last

    def ba_stats0(self):
        st = self.last["ba_stats"]
        return st[0] if isinstance(st, list) else st


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
    if done == 0:
        return 0.0, wall, 0
    return done / slowest, wall, n_procs


def cpu_baseline(frames, n_frames, ba_iters):
    """The CPU oracle (the build's restatement of the reference's OpenCV / SciPy-side arithmetic) on the same
    workload, one host thread, a bounded sample of frames.  Reported, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ba_oracle as bo
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    p = syn.grid_points(N_PTS, W_IMG, H_IMG, seed=7)
    s = syn.make_ba_scene(n_pts=BA_N, n_slots=BA_W, seed=0)
    K = s["K"]
    P0 = (K @ np.hstack([syn.rodrigues(s["poses_gt"][3, :3]), s["poses_gt"][3, 3:, None]])).astype(np.float32)
    P1 = (K @ np.hstack([syn.rodrigues(s["poses_gt"][0, :3]), s["poses_gt"][0, 3:, None]])).astype(np.float32)
    t0 = time.perf_counter()
    for t in range(n_frames):
        a, b = frames[pingpong(t, len(frames))], frames[pingpong(t + 1, len(frames))]
        p1, st, err = o.klt(a, b, p)                                   # builds both pyramids, like one cv2 call
        o.triangulate(P0, P1, s["obs"][3, :N_NEW], s["obs"][0, :N_NEW])
        bo.solve(K, s["poses0"], s["points0"], s["obs"], max_iters=ba_iters, ftol=1e-3, xtol=1e-3)
        mask = np.full((H_IMG, W_IMG), 255, np.uint8)
        for x, y in np.int32(p1):
            o.circle_mask(mask, (x, y), 7, 0)
        o.good_features(b, mask)
        p = p1
    dt = time.perf_counter() - t0
    return n_frames / dt, dt


def main():
    global W_IMG, H_IMG, N_PTS, N_NEW, BA_N, BA_W, WORKLOAD, K_CAM
    a = parse()
    if a.cpu_worker >= 0:          # CPU-baseline worker process: never touches the GPU library
        from vo_mi355x import synthetic as syn
        frames = syn.make_sequence(a.frames, W_IMG, H_IMG, seed=1234 + a.cpu_worker)[0]
        cpu_baseline(frames, 1, a.ba_iters)                    # page in the oracle, first-call costs
        v, secs = cpu_baseline(frames, a.cpu_frames, a.ba_iters)
        print(json.dumps({"frames": a.cpu_frames, "seconds": secs}))
        return
    dist = Dist()
    cpu = None
    if dist.rank == 0 and dist.world == 1 and not a.no_cpu_baseline and a.workload == "A" and a.cpu_worker < 0:
        # before anything initialises the GPU in this process (child processes are started here)
        n_procs = max(1, min(a.cpu_procs, os.cpu_count() or 1))
        v, secs, cores = cpu_baseline_parallel(n_procs, a.cpu_frames, a.ba_iters)
        cpu = {"value": round(v, 3), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": "%d worker processes (1 core each, independent sequences) x %d frames of the same workload on the CPU "
                         "oracle (C: pyramid+KLT+Shi-Tomasi+DLT, numpy: BA), %.1f s wall; %.2f frames/s per core; host has %d cores"
                         % (cores, a.cpu_frames, secs, v / max(cores, 1), os.cpu_count() or 0)}
    from vo_mi355x import VoContext, synthetic as syn
    t_gen = time.perf_counter()
    c5 = a.workload == "config5"
    pl = a.workload == "pipeline"
    if pl:
        a.no_cpu_baseline = True
    if c5:
        # ONE sequence over all ranks: every rank runs the (launch-bound) front end on the whole frame redundantly and
        # owns 1/n_ranks of the landmarks of the 20-frame bundle adjustment
        W_IMG, H_IMG, N_PTS, N_NEW, BA_N, BA_W = 1920, 1080, 5000, 1000, 5000, 20
        WORKLOAD = "single_seq_1920x1080_5000pts_ba20_landmark_sharded"
        K_CAM = np.array([[1100.0, 0, 960.0], [0, 1100.0, 540.0], [0, 0, 1]])
        a.seqs, a.ctxs, a.host_threads, a.fixed_ba_budget, a.no_cpu_baseline = 1, 1, 1, True, True
        uid = dist.bcast_bytes(VoContext.comm_unique_id() if dist.rank == 0 else np.zeros(128, np.uint8))
        frame_sets = [syn.make_sequence(a.frames, W_IMG, H_IMG, seed=1234, margin=96)[0]]
        seqs = [Group(dist.local_rank, frame_sets, seed0=0, batch=1, ba_iters=a.ba_iters, shard=(dist.rank, dist.world, uid))]
    else:
        a.ctxs = max(1, min(a.ctxs, a.seqs))
        per = [a.seqs // a.ctxs + (1 if i < a.seqs % a.ctxs else 0) for i in range(a.ctxs)]
        frame_sets = [syn.make_sequence(a.frames, W_IMG, H_IMG, seed=1234 + 16 * dist.rank + k)[0] for k in range(min(4, a.seqs))]
        seqs = [Group(dist.local_rank, frame_sets, seed0=1000 * dist.rank + 100 * i, batch=per[i], ba_iters=a.ba_iters, pipeline=pl)
                for i in range(a.ctxs)]
    t_setup = time.perf_counter() - t_gen
    for s in seqs:
        s.c.set_graph_mode(bool(a.graph))
        s.max_inflight = 1 if (a.graph or pl) else 2
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
    # ---- timed region: exactly K steps; KLT kernel bracketed by hipEvents on its own stream ----
    for s in seqs:
        s.c.profile_enable((s.c.PROF_KLT,))
        s.c.sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()                                        # every one of the K frames is complete and fetched inside the timed region
    for s in seqs:
        s.c.sync()
    dist.barrier()
    dt = dist.max(time.perf_counter() - t0)
    klt_ms, klt_n = 0.0, 0
    for s in seqs:
        ms, n = s.c.profile_read(s.c.PROF_KLT)
        klt_ms += ms
        klt_n += n
        s.c.profile_enable(())
    frames_total = float(a.steps) if c5 else dist.sum(float(a.steps * a.seqs))     # config 5: ONE sequence on all ranks
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
        if s0.pipeline:
            live = (it >= 0).any(axis=1)
            klt_bytes = float(((5120.0 + 1024.0 * np.maximum(it[live], 0))).sum())
            it_mean = [float(np.maximum(it[live][:, l], 0).mean()) for l in range(it.shape[1])]
        else:
            klt_bytes = N_PTS * s0.B * sum(5120.0 + 1024.0 * x for x in it_mean)
        klt_avg_s = (klt_ms / max(klt_n, 1)) * 1e-3
        achieved = klt_bytes / klt_avg_s / 1e9 if klt_avg_s > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "klt_traffic.json")     # HBM bytes per k_klt_track launch from rocprofv3 --pmc
        if os.path.exists(tf):                                      # passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), see DESIGN.md
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": "k_klt_track", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_us": round(klt_avg_s * 1e6, 3), "algorithmic_bytes_per_launch": int(klt_bytes),
                "klt_mean_iters_per_level": [round(x, 3) for x in it_mean],
                # the same launch when context 0 runs its steps alone after the timed region (no other context's kernels beside it)
                "alone_avg_launch_us": round(stage["klt"] * 1e3, 3),
                "alone_frac": round(klt_bytes / max(stage["klt"] * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "HBM figure = algorithmic bytes / launch time as the contract defines it; the kernel itself is vector-ALU bound "
                        "(rocprofv3 VALUBusy 96 %, profiles/r01_pmc_valu_batch32.csv) and moves 0.13 GB per launch through HBM. "
                        "`frac` is the launch as timed INSIDE the run: with several batched contexts it shares the vector ALUs with the "
                        "Shi-Tomasi / BA kernels of the other contexts and takes longer than on its own (`alone_*`)"}
        out = {"metric": ("frames/sec @1920x1080, 5000 KLT pts, 20-frame sharded BA (config 5)" if c5 else
                          "frames/sec, Pipeline.step on the device @1241x376 (track table <= 2000 pts, PnP, DLT, 10-frame BA, re-detection)" if pl else
                          "frames/sec @1241x376, 2000 KLT pts, 10-frame BA window"), "value": round(fps, 2),
               "unit": "frames/s", "n_gpus": dist.world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if c5 else "weak",
               "vs_baseline": None, "dtype": "u8/i32 (KLT, Shi-Tomasi) + f64 (DLT, BA)", "data": "synthetic",
               "config": {"workload": ("pipeline_step_" + WORKLOAD) if pl else WORKLOAD, "width": W_IMG, "height": H_IMG,
                          "klt_points": N_PTS, "klt_win": 31, "klt_levels": 4, "dlt_points": N_NEW,
                          "ba_landmarks": BA_N, "ba_window": BA_W, "ba_observations": BA_N * BA_W,
                          "ba_lm_iteration_budget": a.ba_iters, "ba_lm_iterations_run": ba_stats["iters"],
                          "ba_final_cost": round(ba_stats["cost"], 4), "ba_initial_cost": round(ba_stats["cost0"], 2),
                          "launch": "hipGraph replay" if a.graph else "plain", "ba_budget": "fixed" if a.fixed_ba_budget else "adaptive (last + 2)", "host_threads": max(a.host_threads, 1),
                          "sequences_per_gpu": a.seqs, "batched_contexts_per_gpu": a.ctxs,
                          "frames_per_step": 1 if c5 else a.seqs * dist.world,
                          "parallelism": ("one sequence, BA landmarks sharded over %d GPU(s), RCCL all-reduce of the reduced camera "
                                          "packet + 4 statistics per LM iteration, front end replicated" % dist.world) if c5 else
                                         ("independent sequences, %d per GPU in %d batched context(s) x %d GPU(s), no collective"
                                          % (a.seqs, a.ctxs, dist.world))},
               "pipeline": ({"mean_live_tracks": round(float(np.mean(s0.last["n_tracks"])), 1),
                             "pnp_inliers": (s0.last["pnp_stats"][0] if isinstance(s0.last["pnp_stats"], list) else s0.last["pnp_stats"])["n_inliers"],
                             "pnp_status_ok": all(x["status"] == 0 for x in (s0.last["pnp_stats"] if isinstance(s0.last["pnp_stats"], list) else [s0.last["pnp_stats"]]))}
                            if pl else None),
               "stage_ms_per_batched_launch_group": stage, "roofline": roof, "cpu_baseline": cpu,
               "setup_s": round(t_setup, 2)}
    dist.barrier()
    for s in seqs:
        s.c.close()
    dist.close()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
