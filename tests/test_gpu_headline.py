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


def test_headline_shape_256_sequences_against_the_oracle():
    sys.path.insert(0, ROOT)
    import ba_oracle as bo
    import bench
    import vo_oracle as o
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
        it_dev = it_dev.reshape(B, N, -1)
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
                    assert np.array_equal(it_dev[b][:, :its.shape[1]], its), b
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
                pyr = o.build_pyramid(fs[f])
                for lvl, im in enumerate(pyr):
                    img, der = g.c.pyramid_read(which, lvl, seq=b)
                    assert np.array_equal(img, im), (b, which, lvl)
                    assert np.array_equal(der, o.scharr(im)), (b, which, lvl)
        # ---- one more step with the frames handed over by the host (page-locked arrays, one per sequence: vo_frame_step_host, what the bench's
        # `host_frames` figure times): the same tracker results as the oracle on the next frame ----
        assert g.use_host_frames(True) == B * W * H
        g.enqueue()
        r = g.fetch()
        for b in CHECKED:
            fs = frame_sets[b % 8]
            p1, st, err = o.klt(fs[n_steps], fs[n_steps + 1], results[-1]["points2d"][b])
            assert np.array_equal(r["points2d"][b], p1) and np.array_equal(r["status"][b], st) and np.array_equal(r["err"][b], err), b
            img, der = g.c.pyramid_read(1, 0, seq=b)
            assert np.array_equal(img, fs[n_steps + 1]), b
        # the iteration counts of the batch are what the bench reports its budget on: not all problems take the same number
        its_all = np.array([[x["iters"] for x in r["ba_stats"]] for r in results])
        assert its_all.min() >= 3 and its_all.max() <= 30 and len(np.unique(its_all)) >= 3
    finally:
        g.c.close()


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
