"""GPU: the fused per-frame step (plain launches and hipGraph replay) equals the individual resident calls, frame by frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(c, frames, pts, scene, n_new):
    from vo_mi355x import synthetic as syn
    c.upload_sequence(frames)
    c.points_upload(pts)
    K = scene["K"]
    H0, H1 = np.eye(4), np.eye(4)
    H0[:3, :3], H0[:3, 3] = syn.rodrigues(scene["poses_gt"][3, :3]), scene["poses_gt"][3, 3:]
    H1[:3, :3], H1[:3, 3] = syn.rodrigues(scene["poses_gt"][0, :3]), scene["poses_gt"][0, 3:]
    c.dlt_upload((K @ H0[:3]).astype(np.float32), (K @ H1[:3]).astype(np.float32), scene["obs"][3, :n_new],
                 scene["obs"][0, :n_new], K, H0, H1)
    c.ba_upload(K, scene["poses0"], scene["points0"], scene["obs"])
    c.push_frame_resident(0)


@pytest.mark.parametrize("graph,block_size", [(False, 31), (True, 31), (True, 15), (False, 7)])
def test_frame_step_matches_individual_calls(graph, block_size):
    """block size 31 takes the fused Shi-Tomasi eigenvalue kernel, the others the two-kernel form whose row-sum planes are
    allocated on demand (before a graph capture starts)"""
    from vo_mi355x import VoContext, synthetic as syn
    w, h, n, n_new = 640, 240, 600, 200
    frames, _ = syn.make_sequence(5, w=w, h=h, seed=21, margin=64)
    pts = syn.grid_points(n, w, h, seed=4)
    scene = syn.make_ba_scene(n_pts=300, n_slots=6, seed=2, visibility=0.9)
    order = [1, 2, 3, 4, 3, 2, 1]
    ref = []
    with VoContext(w, h, max_pts=1024) as c:
        _setup(c, frames, pts, scene, n_new)
        bap = c.ba_params(max_iters=6)
        for f in order:
            c.push_frame_resident(f)
            c.klt_track_resident(n)
            c.dlt_resident()
            c.ba_solve_resident(bap)
            c.shi_tomasi_resident(n, 7, params=c.st_params(block_size=block_size))
            ref.append((c.points_download(n), c.dlt_fetch(), c.ba_fetch(), c.shi_tomasi_fetch()))
    with VoContext(w, h, max_pts=1024) as c:
        c.set_graph_mode(graph)
        _setup(c, frames, pts, scene, n_new)
        bap = c.ba_params(max_iters=6)
        for k, f in enumerate(order):
            c.frame_step_resident(f, n, ba=bap, st=c.st_params(block_size=block_size))
            got = c.frame_fetch()
            (p, st, err), (X4, depth, reproj), (po, pt, bst), corners = ref[k]
            assert np.array_equal(got["points2d"], p) and np.array_equal(got["status"], st) and np.array_equal(got["err"], err), k
            assert np.array_equal(got["X4"], X4, equal_nan=True) and np.array_equal(got["depth1"], depth, equal_nan=True)
            assert np.array_equal(got["reproj"], reproj, equal_nan=True)   # unobserved pairs (NaN uv) stay NaN
            assert np.array_equal(got["poses"], po) and np.array_equal(got["landmarks"], pt)
            assert got["ba_stats"]["cost"] == bst["cost"] and got["ba_stats"]["iters"] == bst["iters"]
            assert np.array_equal(got["corners"], corners), k


@pytest.mark.parametrize("layout", [1, "pipeline", "pipeline_gated"])
def test_two_steps_in_flight_match_one_at_a_time(layout, monkeypatch):
    """step t + 1 may be enqueued before step t is fetched (alternating pinned mirrors); results are those of the
    one-at-a-time loop on ONE stream, fetched oldest first; a third step without a fetch is refused.  layout 1: re-detection and
    triangulation on a side stream; "pipeline": three streams, the bundle adjustment of frame t runs beside the pyramid and the
    KLT of frame t + 1 (vo_set_side_stream(ctx, 2)); "pipeline_gated": the form a batch of >= 8 sequences gets by default, forced here on one
    sequence -- the tracker launch of frame t + 1 waits for the first 2 LM groups of frame t, stream A leaves 32 compute units free"""
    gated = layout == "pipeline_gated"
    from vo_mi355x import VoContext, VoError, synthetic as syn
    if gated:
        monkeypatch.setattr(VoContext, "default_tuning", {"gate_groups": 2, "reserve_cus": 32})
        layout = "pipeline"
    w, h, n, n_new = 640, 240, 600, 200
    frames, _ = syn.make_sequence(5, w=w, h=h, seed=23, margin=64)
    pts = syn.grid_points(n, w, h, seed=5)
    scene = syn.make_ba_scene(n_pts=300, n_slots=6, seed=3, visibility=0.9)
    order = [1, 2, 3, 4, 3, 2, 1, 0] * 3
    keys = ("points2d", "status", "err", "X4", "depth1", "reproj", "poses", "landmarks", "corners")
    with VoContext(w, h, max_pts=1024) as c:
        c.set_side_stream(False)
        _setup(c, frames, pts, scene, n_new)
        bap = c.ba_params(max_iters=6)
        ref = []
        for f in order:
            c.frame_step_resident(f, n, ba=bap)
            ref.append(c.frame_fetch())
    with VoContext(w, h, max_pts=1024) as c:
        c.set_side_stream(layout)
        if gated:
            assert c.step_layout() == {"layout": 2, "gate_groups": 2, "reserved_cus": 32}
        _setup(c, frames, pts, scene, n_new)
        bap = c.ba_params(max_iters=6)
        got = []
        c.frame_step_resident(order[0], n, ba=bap)
        for f in order[1:]:
            c.frame_step_resident(f, n, ba=bap)       # two in flight
            got.append(c.frame_fetch())               # the older one
        with pytest.raises(VoError):
            c.frame_step_resident(1, n, ba=bap)       # 1 in flight is fine ...
            c.frame_step_resident(2, n, ba=bap)       # ... a third is refused
        got.append(c.frame_fetch())
        got.append(c.frame_fetch())                   # the accepted extra step
        again = c.frame_fetch()                       # nothing in flight: the last step again
    for k in range(len(order)):
        for key in keys:
            assert np.array_equal(got[k][key], ref[k][key], equal_nan=True), (k, key)
        assert got[k]["ba_stats"] == ref[k]["ba_stats"]
    for key in keys:
        assert np.array_equal(again[key], got[-1][key], equal_nan=True)


def test_resident_shi_tomasi_keeps_no_map_and_hands_the_mask_back_clean(seq3):
    """The resident path (fused frame step, track table) forms the eigenvalue map and the 3 x 3 suppression in ONE kernel: the map
    is never stored and the exclusion mask is restored while it is read.  A synchronous call afterwards (which re-initialises
    and keeps both) still equals the oracle, and so does the resident result before it."""
    import vo_oracle as o
    from vo_mi355x import VoContext, VoError, synthetic as syn
    frames, _ = seq3
    p0 = syn.grid_points(500, frames[0].shape[1], frames[0].shape[0], seed=3)
    with VoContext(frames[0].shape[1], frames[0].shape[0], max_pts=1024) as c:
        c.push_frame(frames[0]); c.push_frame(frames[1])
        c.points_upload(p0)
        for rep in range(2):                              # the second launch runs WITHOUT k_st_mask_init on the restored mask
            c.shi_tomasi_resident(len(p0), 7)
            got = c.shi_tomasi_fetch()
            mask = np.full(frames[1].shape, 255, np.uint8)
            for x, y in np.int32(p0):
                o.circle_mask(mask, (x, y), 7, 0)
            ref, _, nc = o.good_features(frames[1], mask, return_aux=True)
            assert np.array_equal(got, ref), rep
            with pytest.raises(VoError):
                c.shi_tomasi_read()                       # neither map nor mask kept
        got = c.shi_tomasi(p0, 7)
        eig, m, n = c.shi_tomasi_read()
        assert np.array_equal(got, ref) and np.array_equal(m, mask) and n == nc
        assert np.array_equal(eig, o.min_eig(frames[1]))


def test_graph_replay_two_steps_in_flight_with_a_ba_bank():
    """graph mode: one captured graph per (frame parity, pinned mirror half, selected bank problem), TWO steps in flight like the plain
    path (round 2 kept one graph per parity and one step in flight -- that, not the replay, made graph mode slower); results bit for bit
    those of plain launches one at a time"""
    from vo_mi355x import VoContext, synthetic as syn
    w, h, n, n_new, nb = 640, 240, 600, 200, 3
    frames, _ = syn.make_sequence(5, w=w, h=h, seed=23, margin=64)
    pts = syn.grid_points(n, w, h, seed=5)
    scenes = [syn.make_ba_scene(n_pts=300, n_slots=6, seed=3 + k, visibility=0.9, pt_noise=0.3 + 0.2 * k) for k in range(nb)]
    order = [1, 2, 3, 4, 3, 2, 1, 0] * 3
    keys = ("points2d", "status", "err", "X4", "depth1", "reproj", "poses", "landmarks", "corners")

    def setup(c):
        _setup(c, frames, pts, scenes[0], n_new)
        c.ba_upload_bank(scenes[0]["K"], np.stack([s["poses0"][None] for s in scenes]), np.stack([s["points0"][None] for s in scenes]),
                         np.stack([s["obs"][None] for s in scenes]))
    with VoContext(w, h, max_pts=1024) as c:
        c.set_side_stream(False)
        setup(c)
        bap = c.ba_params(max_iters=8)
        ref = []
        for t, f in enumerate(order):
            c.ba_select(t % nb)
            c.frame_step_resident(f, n, ba=bap)
            ref.append(c.frame_fetch())
    with VoContext(w, h, max_pts=1024) as c:
        c.set_graph_mode(True)
        setup(c)
        bap = c.ba_params(max_iters=8)
        got = []
        c.ba_select(0)
        c.frame_step_resident(order[0], n, ba=bap)
        for t, f in enumerate(order[1:], start=1):
            c.ba_select(t % nb)
            c.frame_step_resident(f, n, ba=bap)       # two in flight
            got.append(c.frame_fetch())
        got.append(c.frame_fetch())
    assert len({tuple(np.round(r["ba_stats"]["cost"], 6) for r in ref[k::nb]) for k in range(nb)}) == nb     # the bank's problems differ
    for k in range(len(order)):
        for key in keys:
            assert np.array_equal(got[k][key], ref[k][key], equal_nan=True), (k, key)
        assert got[k]["ba_stats"] == ref[k]["ba_stats"]


def test_step_layout_defaults():
    """what vo_set_side_stream puts into effect (vo_step_layout): one sequence keeps the plain pipelined layout, a batch of >= 8 gets the gated
    tracker launch and the compute-unit reserve; graph replay and layouts 0 / 1 run on the unmasked stream"""
    from vo_mi355x import VoContext
    with VoContext(160, 120, max_pts=64) as c:
        c.set_side_stream("pipeline")
        assert c.step_layout() == {"layout": 2, "gate_groups": 0, "reserved_cus": 0}
    with VoContext(160, 120, max_pts=64, batch=8) as c:
        assert c.step_layout() == {"layout": 1, "gate_groups": 0, "reserved_cus": 0}
        c.set_side_stream("pipeline")
        assert c.step_layout() == {"layout": 2, "gate_groups": 5, "reserved_cus": 32}
        c.set_graph_mode(True)
        assert c.step_layout() == {"layout": 2, "gate_groups": 0, "reserved_cus": 0}
        c.set_graph_mode(False)
        assert c.step_layout() == {"layout": 2, "gate_groups": 5, "reserved_cus": 32}
        c.set_side_stream(True)
        assert c.step_layout() == {"layout": 1, "gate_groups": 0, "reserved_cus": 0}


def test_gated_layout_with_steps_that_skip_the_adjustment(monkeypatch):
    """the gated pipelined layout (the tracker launch of frame t + 1 waits for the wide LM groups of frame t) when some steps carry no bundle
    adjustment, a zero-iteration budget or a budget shorter than the gate: the wait then refers to an older, finished record; results are those
    of the one-stream loop"""
    from vo_mi355x import VoContext, synthetic as syn
    w, h, n, n_new = 640, 240, 600, 200
    frames, _ = syn.make_sequence(5, w=w, h=h, seed=29, margin=64)
    pts = syn.grid_points(n, w, h, seed=6)
    scene = syn.make_ba_scene(n_pts=300, n_slots=6, seed=5, visibility=0.9)
    plan = [(1, True, 6), (2, False, 6), (3, True, 0), (4, True, 2), (3, True, 6), (2, False, 6), (1, False, 6), (0, True, 1), (1, True, 6), (2, True, 6)]
    keys = ("points2d", "status", "err", "X4", "poses", "landmarks", "corners")

    def run(c, in_flight):
        _setup(c, frames, pts, scene, n_new)
        out = []
        for k, (f, do_ba, iters) in enumerate(plan):
            c.frame_step_resident(f, n, do_ba=do_ba, ba=c.ba_params(max_iters=iters))
            if k >= in_flight - 1:
                out.append(c.frame_fetch())
        while len(out) < len(plan):
            out.append(c.frame_fetch())
        return out

    with VoContext(w, h, max_pts=1024) as c:
        c.set_side_stream(False)
        ref = run(c, 1)
    with VoContext(w, h, max_pts=1024) as c:
        c.set_tuning(gate_groups=3, reserve_cus=32)
        c.set_side_stream("pipeline")
        assert c.step_layout()["gate_groups"] == 3
        got = run(c, 2)
    for k in range(len(plan)):
        assert set(got[k]) == set(ref[k]), (k, set(got[k]) ^ set(ref[k]))
        for key in keys:
            if key in got[k]:                      # (a step without an adjustment returns no poses / landmarks)
                assert np.array_equal(got[k][key], ref[k][key], equal_nan=True), (k, key)
        assert {"points2d", "status", "corners"} <= set(got[k])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_pipelined_batch_random_stage_patterns(seed):
    """a batched context of 8 sequences in the pipelined layout AS vo_set_side_stream puts it into effect (gate of 5 LM groups, 32 compute units
    set aside -- no environment override): random stage sets (with / without triangulation, adjustment, re-detection), random LM budgets (0 ... 9:
    shorter and longer than the gate) and two steps in flight, against the same steps one at a time on ONE stream"""
    from vo_mi355x import VoContext, synthetic as syn
    rng = np.random.default_rng(1000 + seed)
    B, w, h, n, n_new = 8, 320, 160, 250, 100
    seqs = [syn.make_sequence(5, w=w, h=h, seed=40 + 3 * seed + b, margin=48)[0] for b in range(B)]
    pts = np.stack([syn.grid_points(n, w, h, seed=9 + b) for b in range(B)])
    scenes = [syn.make_ba_scene(n_pts=200, n_slots=6, seed=70 + 5 * seed + b, visibility=0.9, pt_noise=0.2 + 0.1 * b) for b in range(B)]
    plan = [(int(rng.integers(0, 5)), bool(rng.integers(0, 2)), bool(rng.integers(0, 4) > 0), bool(rng.integers(0, 2)), int(rng.integers(0, 10))) for _ in range(14)]

    def setup(c):
        c.upload_sequence(np.stack(seqs)); c.points_upload(pts)
        d = []
        for s in scenes:
            K = s["K"]
            H0, H1 = np.eye(4), np.eye(4)
            H0[:3, :3], H0[:3, 3] = syn.rodrigues(s["poses_gt"][3, :3]), s["poses_gt"][3, 3:]
            H1[:3, :3], H1[:3, 3] = syn.rodrigues(s["poses_gt"][0, :3]), s["poses_gt"][0, 3:]
            d.append(((K @ H0[:3]).astype(np.float32), (K @ H1[:3]).astype(np.float32), s["obs"][3, :n_new], s["obs"][0, :n_new], K, H0, H1))
        c.dlt_upload(*[np.stack([x[k] for x in d]) for k in range(7)])
        c.ba_upload(np.stack([s["K"] for s in scenes]), np.stack([s["poses0"] for s in scenes]), np.stack([s["points0"] for s in scenes]),
                    np.stack([s["obs"] for s in scenes]))
        c.push_frame_resident(0)

    def run(c, in_flight):
        out = []
        for k, (f, do_dlt, do_ba, do_st, iters) in enumerate(plan):
            c.frame_step_resident(f, n, do_dlt=do_dlt, do_ba=do_ba, do_st=do_st, ba=c.ba_params(max_iters=iters))
            if k >= in_flight - 1:
                out.append(c.frame_fetch())
        while len(out) < len(plan):
            out.append(c.frame_fetch())
        return out

    with VoContext(w, h, max_pts=512, batch=B) as c:
        c.set_side_stream(False)
        setup(c)
        ref = run(c, 1)
    with VoContext(w, h, max_pts=512, batch=B) as c:
        c.set_side_stream("pipeline")
        assert c.step_layout() == {"layout": 2, "gate_groups": 5, "reserved_cus": 32}
        setup(c)
        got = run(c, 2)
    for k in range(len(plan)):
        assert set(got[k]) == set(ref[k]), (k, plan[k])
        for key in got[k]:
            if key == "ba_stats":
                assert got[k][key] == ref[k][key], (k, plan[k])
            elif isinstance(got[k][key], list):            # (corners: one array per sequence, their lengths differ)
                assert len(got[k][key]) == len(ref[k][key]) and all(np.array_equal(x, y) for x, y in zip(got[k][key], ref[k][key])), (k, key, plan[k])
            else:
                assert np.array_equal(got[k][key], ref[k][key], equal_nan=True), (k, key, plan[k])


@pytest.mark.parametrize("batch,layout,source", [(1, 1, "pinned"), (8, "pipeline", "pinned"), (8, "pipeline", "pageable"), (3, 0, "strided")])
def test_host_frame_step_equals_resident_step(batch, layout, source):
    """vo_frame_step_host (this step's images handed over by the host, uploaded on the copy stream: the reference's Pipeline.step(img),
    pipeline.py:98,171-172) = vo_frame_step_resident on the same frames, bit for bit, two steps in flight; pinned arrays (DMA), pageable arrays
    (staged by the runtime), one [batch, h, w] array (one copy) and per-sequence arrays with a row stride > width"""
    from vo_mi355x import VoContext, synthetic as syn
    w, h, n, n_new = 480, 200, 400, 150
    seqs = [syn.make_sequence(6, w=w, h=h, seed=31 + b, margin=64)[0] for b in range(batch)]
    frames = np.stack(seqs)                                     # [batch, 6, h, w]
    pts = np.stack([syn.grid_points(n, w, h, seed=9 + b) for b in range(batch)])
    scenes = [syn.make_ba_scene(n_pts=200, n_slots=5, seed=4 + b, visibility=0.9) for b in range(batch)]
    order = [1, 2, 3, 4, 5, 4, 3, 2, 1, 0, 1, 2]
    keys = ("points2d", "status", "err", "X4", "depth1", "reproj", "poses", "landmarks", "corners")

    def setup(c):
        c.set_side_stream(layout)
        K = np.stack([s["K"] for s in scenes])
        P0, P1, u0, u1, H0s, H1s = [], [], [], [], [], []
        for s in scenes:
            H0, H1 = np.eye(4), np.eye(4)
            H0[:3, :3], H0[:3, 3] = syn.rodrigues(s["poses_gt"][3, :3]), s["poses_gt"][3, 3:]
            H1[:3, :3], H1[:3, 3] = syn.rodrigues(s["poses_gt"][0, :3]), s["poses_gt"][0, 3:]
            P0.append((s["K"] @ H0[:3]).astype(np.float32)); P1.append((s["K"] @ H1[:3]).astype(np.float32))
            u0.append(s["obs"][3, :n_new].astype(np.float32)); u1.append(s["obs"][0, :n_new].astype(np.float32)); H0s.append(H0); H1s.append(H1)
        sq = (lambda x: np.stack(x)) if batch > 1 else (lambda x: x[0])
        c.points_upload(pts if batch > 1 else pts[0])
        c.dlt_upload(sq(P0), sq(P1), sq(u0), sq(u1), K if batch > 1 else K[0], sq(H0s), sq(H1s))
        c.ba_upload(K if batch > 1 else K[0], sq([s["poses0"] for s in scenes]), sq([s["points0"] for s in scenes]), sq([s["obs"] for s in scenes]))

    def run(c, enqueue):
        out = []
        bap = c.ba_params(max_iters=6)
        for k, f in enumerate(order):
            enqueue(c, f, bap)
            if k >= 1:
                out.append(c.frame_fetch())
        out.append(c.frame_fetch())
        return out

    with VoContext(w, h, max_pts=512, batch=batch) as c:
        setup(c)
        c.upload_sequence(frames if batch > 1 else frames[0])
        c.push_frame_resident(0)
        ref = run(c, lambda c, f, bap: c.frame_step_resident(f, n, ba=bap))
    with VoContext(w, h, max_pts=512, batch=batch) as c:
        setup(c)
        if source == "pinned":
            host = VoContext.host_alloc((6, batch, h, w))       # a [batch, h, w] block per frame: ONE copy per step
            host[:] = frames.transpose(1, 0, 2, 3)
            give = lambda f: host[f]
        elif source == "pageable":
            give = lambda f: [seqs[b][f] for b in range(batch)]  # views into the sequences' own arrays: `batch` copies per step
        else:
            wide = np.zeros((batch, 6, h, w + 37), np.uint8)
            wide[..., :w] = frames
            give = lambda f: [wide[b, f, :, :w] for b in range(batch)]
        c.push_frame(frames[:, 0] if batch > 1 else frames[0, 0])
        got = run(c, lambda c, f, bap: c.frame_step_host(give(f), n, ba=bap))
    assert len(got) == len(ref) == len(order)
    for k, (g, r) in enumerate(zip(got, ref)):
        for key in keys:
            if key == "corners" and batch > 1:
                assert all(np.array_equal(g[key][b], r[key][b]) for b in range(batch)), (k, key)
            else:
                assert np.array_equal(g[key], r[key], equal_nan=True), (k, key)
        gs = g["ba_stats"] if isinstance(g["ba_stats"], list) else [g["ba_stats"]]
        rs = r["ba_stats"] if isinstance(r["ba_stats"], list) else [r["ba_stats"]]
        assert [(x["cost"], x["iters"], x["status"]) for x in gs] == [(x["cost"], x["iters"], x["status"]) for x in rs], k


def test_host_alloc_is_a_plain_numpy_array_and_frees_itself():
    import gc
    from vo_mi355x import VoContext
    a = VoContext.host_alloc((4, 16, 32))
    assert a.shape == (4, 16, 32) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"]
    a[:] = 7
    v = a[1]
    del a
    gc.collect()
    assert int(v.sum()) == 7 * 16 * 32           # a view keeps the allocation alive
    del v
    gc.collect()


def test_gated_layout_is_suspended_by_closed_loop_steps_and_comes_back():
    """The gate + CU mask of a batch's pipelined layout are a property only vo_set_side_stream / vo_set_graph_mode / vo_set_tuning change: a
    closed-loop step (vo_pipe_step: its chain needs the whole chip) SUSPENDS them -- vo_step_layout says so -- and the next frame step puts them
    back; the layout cannot be switched while closed-loop steps are in flight; tracker results around it are those of a context that never ran the loop"""
    import copy
    from vo_mi355x import VoContext, VoError, synthetic as syn
    from vo_mi355x.resident import ResidentPipeline
    w, h, B, n, t1 = 256, 160, 8, 200, 3
    sc = syn.sway_scene(10, w=w, h=h, f=260.0, seed=2024, pose_fn=lambda t: syn.sway_pose(t, period=24.0))
    frames = np.stack([sc["frames"]] * B)
    pts = np.stack([syn.grid_points(n, w, h, seed=b, margin=12) for b in range(B)])

    def klt_steps(c, order):
        out = []
        for f in order:
            c.frame_step_resident(f, n, do_dlt=False, do_ba=False, do_st=False)
            r = c.frame_fetch()
            out.append((r["points2d"].copy(), r["status"].copy(), r["err"].copy()))
        return out

    with VoContext(w, h, max_pts=1024, batch=B) as c:
        c.upload_sequence(frames); c.points_upload(pts); c.push_frame_resident(0)
        ref = klt_steps(c, [1, 2]) + klt_steps(c, [5, 6])
    with VoContext(w, h, max_pts=1024) as boot:
        state, _ = syn.gt_bootstrap(boot, sc, 0, t1)
    with VoContext(w, h, max_pts=1024, batch=B) as c:
        c.upload_sequence(frames); c.points_upload(pts); c.push_frame_resident(0)
        c.set_side_stream("pipeline")
        gated = {"layout": 2, "gate_groups": 5, "reserved_cus": 32}
        assert c.step_layout() == gated
        got = klt_steps(c, [1, 2])
        assert c.step_layout() == gated
        rp = ResidentPipeline(c, np.stack([sc["K"]] * B), ba_max_iters=6, pnp_blind_batches=4)
        rp.seed([copy.deepcopy(state) for _ in range(B)], None, None, 1)
        c.push_frame_resident(t1)
        rp.step(t1 + 1)
        assert c.step_layout() == {"layout": 2, "gate_groups": 0, "reserved_cus": 0}          # suspended, not silently gone: layout is still 2
        with pytest.raises(VoError):
            c.set_side_stream("pipeline")                                                     # a closed-loop step is in flight
        rec = rp.fetch()
        assert all(r["status"] == 0 for r in rec)
        rp.step(t1 + 2); rp.fetch()
        # back to the frame steps: the points of the first run, the frame store re-primed
        c.points_upload(got[-1][0]); c.push_frame_resident(2); c.push_frame_resident(4)
        got += klt_steps(c, [5, 6])
        assert c.step_layout() == gated
    with VoContext(w, h, max_pts=1024, batch=B) as c:     # the reference for the second half: frames 4 -> 5 -> 6 from the same points
        c.upload_sequence(frames); c.points_upload(ref[1][0]); c.push_frame_resident(4)
        ref = ref[:2] + klt_steps(c, [5, 6])
    for k, (a, b) in enumerate(zip(got, ref)):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), k
