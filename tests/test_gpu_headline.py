"""GPU: the bench's own headline configuration against the oracle -- ONE batched context of 256 sequences at 1241x376 with 2000 tracked points,
the pipelined stream layout as `vo_set_side_stream(ctx, 2)` applies it to such a batch (the tracker launch behind 4 LM launch groups, 32 compute units
reserved), the bench's bank of eight 2000 x 10 bundle adjustments per sequence, LM cap 30, two frame steps in flight.  Every other test reaches these
kernels at smaller shapes; the rules that only fire here (gate_groups = 4 at a batch >= 256, two workgroups per BA problem + 16-way compaction, the band
count of the fused eigenvalue kernel, the interior / border index split of the pyramid kernels, the XCD remap over 256 sequences) are compared HERE.

Follows the reference's frame loop (src/extractor/extractor.py:38-132: KLT + re-detection; :255-277: DLT; src/bundle_adjuster/bundle_adjuster.py:127-215)
through the fused step the bench times (bench.Group = what `python bench.py` runs)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKED = (0, 1, 127, 255)


@pytest.fixture(scope="module")
def headline_run():
    """bench.Group in the bench's default configuration, three steps with two in flight, then one step with the frames handed over by the host;
    everything the tests below look at is read back here, the context is closed before they run"""
    sys.path.insert(0, ROOT)
    import bench
    from vo_mi355x import synthetic as syn
    W, H, N, B = bench.W_IMG, bench.H_IMG, bench.N_PTS, 256
    assert (W, H, N, bench.N_NEW, bench.BA_N, bench.BA_W) == (1241, 376, 2000, 1000, 2000, 10)
    n_steps = 3
    # the bench's frames: the first frames of its 100-frame periodic sequences, 8 distinct sequences serving the batch (bench.py main())
    frame_sets = [syn.make_sequence(100, W, H, seed=1234 + k, periodic=True, n_render=n_steps + 2)[0] for k in range(8)]
    g = bench.Group(0, frame_sets, seed0=0, batch=B, ba_iters=30)
    try:
        g.c.set_side_stream("pipeline")
        assert g.c.step_layout() == {"layout": 2, "gate_groups": 4, "reserved_cus": 32}
        assert g.n_ba == 8 and g.ba_prm.max_iters == 30 and g.max_inflight == 2 and g.adaptive
        results = []
        for _ in range(n_steps):
            g.enqueue()
            if g.inflight == 2:
                results.append(g.fetch())
        while g.inflight:
            results.append(g.fetch())
        assert len(results) == n_steps and g.at_cap == 0
        _, _, _, it_dev = g.c.points_download(N, return_iters=True)          # iteration counts of the LAST tracker launch
        pyr = {(b, which, lvl): g.c.pyramid_read(which, lvl, seq=b) for b in CHECKED for which in (0, 1) for lvl in range(4)}
        # one more step with the frames handed over by the host (page-locked arrays, one per sequence: vo_frame_step_host, what the bench's
        # `host_frames` figure times)
        assert g.use_host_frames(True) == B * W * H
        g.enqueue()
        r_host = g.fetch()
        lvl0_host = {b: g.c.pyramid_read(1, 0, seq=b)[0] for b in CHECKED}
        # ... and a step whose frame IS the previous one (the host's frame 4 again, this time from the resident store): the tracker must not move a point
        g.use_host_frames(False)
        g.c.frame_step_resident(n_steps + 1, N, False, False, False, 7, g.klt_prm, g.st_prm, g.ba_prm)
        r_same = g.c.frame_fetch()
        _, _, _, it_same = g.c.points_download(N, return_iters=True)
    finally:
        g.c.close()
    return dict(bench=bench, frame_sets=frame_sets, results=results, it_dev=it_dev.reshape(B, N, -1), pyr=pyr, r_host=r_host, lvl0_host=lvl0_host,
                r_same=r_same, it_same=it_same.reshape(B, N, -1),
                n_steps=n_steps, W=W, H=H, N=N, B=B)


def test_headline_shape_256_sequences_against_the_oracle(headline_run):
    import ba_oracle as bo
    import vo_oracle as o
    from vo_mi355x import synthetic as syn
    R = headline_run
    bench, frame_sets, results, n_steps, W, H, N = R["bench"], R["frame_sets"], R["results"], R["n_steps"], R["W"], R["H"], R["N"]
    for b in CHECKED:
        fs = frame_sets[b % 8]
        p = syn.grid_points(N, W, H, seed=b)
        bank = bench.ba_bank(0, b)
        P0, P1, u0, u1 = bench.dlt_inputs(bank[0])[:4]
        Xr = o.triangulate(P0, P1, u0, u1)
        Xr = (Xr[:3] / Xr[3]).T
        for t in range(1, n_steps + 1):
            r = results[t - 1]
            # ---- KLT (extractor.py:44-45,65-66): positions, status, err bit-exact; iteration counts of the last launch ----
            p1, st, err, its = o.klt(fs[t - 1], fs[t], p, return_iters=True)
            assert np.array_equal(r["points2d"][b], p1), (b, t)
            assert np.array_equal(r["status"][b], st) and np.array_equal(r["err"][b], err), (b, t)
            if t == n_steps:
                assert np.array_equal(R["it_dev"][b][:, :its.shape[1]], its), b
            # ---- re-detection (extractor.py:104-112): discs at the tracked points, ordered corner list bit-exact ----
            mask = np.full((H, W), 255, np.uint8)
            for x, y in np.int32(p1):
                o.circle_mask(mask, (x, y), 7, 0)
            assert np.array_equal(r["corners"][b], o.good_features(fs[t], mask)), (b, t)
            # ---- DLT (extractor.py:255-277): 1e-4 relative ----
            X = (r["X4"][b][:3] / r["X4"][b][3]).T
            assert (np.linalg.norm(X - Xr, axis=1) / np.linalg.norm(Xr, axis=1)).max() <= 1e-4, (b, t)
            # ---- bundle adjustment (bundle_adjuster.py:127-215 through the LM of oracle/ba_oracle.py): problem t % 8 of the bank ----
            q = bank[t % 8]
            ref = bo.solve(q["K"], q["poses0"], q["points0"], q["obs"], max_iters=30, ftol=1e-3, xtol=1e-3)
            s = r["ba_stats"][b]
            assert (s["iters"], s["accepted"], s["status"]) == (ref["iters"], ref["accepted"], ref["status"]), (b, t, s, ref["iters"])
            assert abs(s["cost"] - ref["cost"]) <= 1e-7 * ref["cost"] and abs(s["cost0"] - ref["cost0"]) <= 1e-9 * ref["cost0"], (b, t)
            assert np.abs(r["poses"][b] - ref["poses"]).max() <= 1e-6 and np.abs(r["landmarks"][b] - ref["points"]).max() <= 1e-5, (b, t)
            p = p1
        # ---- frame store after the last step: pyramid levels and Scharr derivatives of cur (frame 3) and prev (frame 2), bit-exact ----
        for which, f in ((1, n_steps), (0, n_steps - 1)):
            for lvl, im in enumerate(o.build_pyramid(fs[f])):
                img, der = R["pyr"][(b, which, lvl)]
                assert np.array_equal(img, im), (b, which, lvl)
                assert np.array_equal(der, o.scharr(im)), (b, which, lvl)
        # ---- the step with the frames from the host: the same tracker results as the oracle on the next frame ----
        p1, st, err = o.klt(fs[n_steps], fs[n_steps + 1], results[-1]["points2d"][b])
        rh = R["r_host"]
        assert np.array_equal(rh["points2d"][b], p1) and np.array_equal(rh["status"][b], st) and np.array_equal(rh["err"][b], err), b
        assert np.array_equal(R["lvl0_host"][b], fs[n_steps + 1]), b
    # the iteration counts of the batch are what the bench reports its budget on: not all problems take the same number
    its_all = np.array([[x["iters"] for x in r["ba_stats"]] for r in results])
    assert its_all.min() >= 3 and its_all.max() <= 30 and len(np.unique(its_all)) >= 3


def test_headline_properties_over_all_256_sequences(headline_run):
    """what the oracle comparison shows for four sequences, as size-independent properties of ALL 256 (and of the host-frame step): the re-detection
    honours its own rules (<= 1 000 corners, pairwise >= 7 px apart, none on an exclusion disc of a tracked point, integer pixels inside the image);
    tracked points stay finite, lost ones are flagged; every adjustment ends by the LM's own tests with a cost not above its start and finite poses /
    landmarks; sequences that share an image sequence AND a bank problem index but not the point set differ (no sequence is served another's data),
    while the triangulation -- the same scene for every sequence b -- is per-sequence deterministic across steps"""
    from scipy.spatial import cKDTree
    R = headline_run
    results, W, H, B = R["results"] + [R["r_host"]], R["W"], R["H"], R["B"]
    for t, r in enumerate(results):
        p, st = r["points2d"], r["status"]
        assert p.shape == (B, 2000, 2) and np.isfinite(p).all() and set(np.unique(st)) <= {0, 1}
        assert st.mean() > 0.98                                                     # the synthetic motion loses (almost) nothing
        inside = (p[..., 0] >= -31) & (p[..., 0] <= W + 31) & (p[..., 1] >= -31) & (p[..., 1] <= H + 31)
        assert inside[st == 1].all()
        for b in range(B):
            c = r["corners"][b]
            assert len(c) <= 1000 and (c == np.rint(c)).all()
            assert (c[:, 0] >= 0).all() and (c[:, 0] < W).all() and (c[:, 1] >= 0).all() and (c[:, 1] < H).all()
            if len(c) > 1:
                assert len(cKDTree(c).query_pairs(6.999)) == 0, (t, b)              # minDistance 7 (featureselect.cpp keeps dx^2 + dy^2 >= 49)
            # no corner on a disc of radius 7 drawn at int32(tracked point) (extractor.py:104-107): the disc covers dx^2 + dy^2 <= 49 at least up to r - 1
            d, _ = cKDTree(np.int32(p[b]).astype(np.float64)).query(c)
            assert (d > 6.0).all(), (t, b, d.min())
        for b, s in enumerate(r["ba_stats"]):
            assert s["status"] in (1, 2, 3) and 3 <= s["iters"] <= 30 and s["cost"] <= s["cost0"] and np.isfinite(s["cost"]), (t, b, s)
        assert np.isfinite(r["poses"]).all() and np.isfinite(r["landmarks"]).all()
        assert np.isfinite(r["X4"][:, :, :]).all()
    # idempotence: tracking a frame against itself leaves every point where it was (up to the float rounding of (p - 15) + 15 at each of the four
    # levels: OpenCV's own arithmetic -- the oracle gives the same bits), with a residual of (almost) nothing (the patch is re-sampled at a position an ulp away:
    # err <= 0.01 grey levels), in ONE iteration per level
    rs, ok = R["r_same"], R["r_host"]["status"] == 1
    assert np.abs(rs["points2d"][ok] - R["r_host"]["points2d"][ok]).max() <= 1e-3 and (rs["status"][ok] == 1).all() and (rs["err"][ok] <= 0.01).all()
    assert (R["it_same"][ok] == 1).all()
    # own data per sequence: sequences 0 and 8 share image sequence 0 but not the points, the scene or the bank
    a, b = results[0], results[0]
    assert not np.array_equal(a["points2d"][0], b["points2d"][8]) and not np.array_equal(a["poses"][0], b["poses"][8])
    # the triangulation inputs of a sequence do not change from step to step (an uploaded pair set): identical output every step
    for r in results[1:]:
        assert np.array_equal(r["X4"], results[0]["X4"])


def test_closed_loop_figure_256_sequences_equals_its_sequences_alone():
    """`closed_loop_w10_256` of the bench line (bench.PipeGroup: Pipeline.step resident on the device, BASELINE's window 10, dead landmarks stay dead,
    2 048-slot tables, 256 sequences in ONE context, three steps in flight): sequences 0, 1 and 255 of the batch against the same sequences stepped
    ALONE in a context of their own -- every integer of the record (list lengths, inliers, new / detected / resurrected counts, LM iterations, status,
    capacity flags) equal, poses and costs to 1e-9 (a batch folds a problem's partial sums in another order than a single problem: ~1e-12)."""
    sys.path.insert(0, ROOT)
    import bench
    from vo_mi355x import VoContext
    n_steps = 5
    scenes = bench.pipe_scenes(2, 12, 4321)
    boot = VoContext(bench.W_IMG, bench.H_IMG, max_pts=4096, device=0)
    try:
        big = bench.PipeGroup(0, scenes, boot, 0, 256, 10, 2048, True, 10, False)
        try:
            recs = []
            for _ in range(n_steps):
                big.enqueue()
                if big.inflight == 3:
                    big.fetch(); recs.append(big.last)
            while big.inflight:
                big.fetch(); recs.append(big.last)
        finally:
            big.c.close()
        assert len(recs) == n_steps and all(r["status"] == 0 for step in recs for r in step)
        for b in (0, 1, 255):
            one = bench.PipeGroup(0, scenes, boot, b, 1, 10, 2048, True, 10, False)
            try:
                for s in range(n_steps):
                    one.enqueue(); one.fetch()
                    ref, got = one.last[0], recs[s][b]
                    for k, v in ref.items():
                        if isinstance(v, np.ndarray):
                            assert np.abs(got[k] - v).max() <= 1e-9, (b, s, k)
                        elif isinstance(v, float):
                            assert abs(got[k] - v) <= 1e-9 * max(1.0, abs(v)), (b, s, k, got[k], v)
                        else:
                            assert got[k] == v, (b, s, k, got[k], v)
            finally:
                one.c.close()
    finally:
        boot.close()


def test_headline_step_is_bitwise_reproducible(headline_run):
    """every reduction of the path has a fixed order (integer sums in the tracker and the eigenvalue pass, partial sets folded in workgroup order, the LM's
    decision derived by every workgroup from the same statistics): a second run of the same three steps in a fresh context -- other launch timing, other
    placement of the 512 000 tracker waves, the tail groups compacted as the running problems of THAT run allow -- gives the same bits for all 256 sequences"""
    R = headline_run
    bench = R["bench"]
    g = bench.Group(0, R["frame_sets"], seed0=0, batch=R["B"], ba_iters=30)
    try:
        g.c.set_side_stream("pipeline")
        again = []
        for _ in range(R["n_steps"]):
            g.enqueue()
            if g.inflight == 2:
                again.append(g.fetch())
        while g.inflight:
            again.append(g.fetch())
    finally:
        g.c.close()
    for t, (a, b) in enumerate(zip(R["results"], again)):
        for k in ("points2d", "status", "err", "X4", "depth1", "reproj", "poses", "landmarks"):
            assert np.array_equal(a[k], b[k], equal_nan=True), (t, k)
        assert all(np.array_equal(x, y) for x, y in zip(a["corners"], b["corners"])), t
        assert [(s["iters"], s["accepted"], s["status"], s["cost"]) for s in a["ba_stats"]] == [(s["iters"], s["accepted"], s["status"], s["cost"]) for s in b["ba_stats"]], t
