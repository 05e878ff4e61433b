"""GPU: a batched context (B sequences in lockstep, one launch for all) gives bit-identical results to B single contexts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [3, 8])     # 8: the XCD-aware KLT block mapping (batch % 8 == 0) is in effect
def test_batched_context_equals_single_contexts(B):
    from vo_mi355x import VoContext, synthetic as syn
    w, h, n, n_new = 480, 200, 300, 120
    seqs = [syn.make_sequence(4, w=w, h=h, seed=60 + b, margin=64)[0] for b in range(B)]
    pts = [syn.grid_points(n, w, h, seed=10 + b, margin=8) for b in range(B)]
    scenes = [syn.make_ba_scene(n_pts=150 + 0 * b, n_slots=5, seed=30 + b, visibility=0.9) for b in range(B)]

    def dlt_in(s):
        K = s["K"]
        H0, H1 = np.eye(4), np.eye(4)
        H0[:3, :3], H0[:3, 3] = syn.rodrigues(s["poses_gt"][3, :3]), s["poses_gt"][3, 3:]
        H1[:3, :3], H1[:3, 3] = syn.rodrigues(s["poses_gt"][0, :3]), s["poses_gt"][0, 3:]
        return ((K @ H0[:3]).astype(np.float32), (K @ H1[:3]).astype(np.float32), s["obs"][3, :n_new].astype(np.float32),
                s["obs"][0, :n_new].astype(np.float32), K, H0, H1)

    # ---- reference: one context per sequence ----
    ref = []
    for b in range(B):
        with VoContext(w, h, max_pts=512) as c:
            c.push_frame(seqs[b][0]); c.push_frame(seqs[b][1])
            klt = c.klt_track(pts[b], return_iters=True)
            st = c.shi_tomasi(klt[0], 7)
            eig, mask, nc = c.shi_tomasi_read()
            tri = c.triangulate(*dlt_in(scenes[b]))
            s = scenes[b]
            ba = c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c.ba_params(max_iters=8))
            # resident stepping
            c.upload_sequence(seqs[b]); c.points_upload(pts[b]); c.dlt_upload(*dlt_in(s)); c.ba_upload(s["K"], s["poses0"], s["points0"], s["obs"])
            c.push_frame_resident(0)
            steps = []
            for f in (1, 2, 3, 2):
                c.frame_step_resident(f, n, ba=c.ba_params(max_iters=5))
                steps.append(c.frame_fetch())
            ref.append((klt, st, (eig, mask, nc), tri, ba, steps))
    # ---- batched ----
    with VoContext(w, h, max_pts=512, batch=B) as c:
        c.push_frame(np.stack([seqs[b][0] for b in range(B)])); c.push_frame(np.stack([seqs[b][1] for b in range(B)]))
        for b in range(B):
            for lvl in range(3):
                assert np.array_equal(c.pyramid_read(1, lvl, seq=b)[0].shape, c.pyramid_read(1, lvl, seq=0)[0].shape)
        p1, stt, err, it = c.klt_track(np.stack(pts), return_iters=True)
        corners = c.shi_tomasi(p1, 7)
        eig, mask, nc = c.shi_tomasi_read()
        d = [dlt_in(s) for s in scenes]
        X4, depth, reproj = c.triangulate(*[np.stack([d[b][k] for b in range(B)]) for k in range(7)])
        po, pt, bst = c.ba_adjust(np.stack([s["K"] for s in scenes]), np.stack([s["poses0"] for s in scenes]),
                                  np.stack([s["points0"] for s in scenes]), np.stack([s["obs"] for s in scenes]),
                                  c.ba_params(max_iters=8))
        for b in range(B):
            klt, st, (reig, rmask, rnc), tri, ba, _ = ref[b]
            assert np.array_equal(p1[b], klt[0]) and np.array_equal(stt[b], klt[1]) and np.array_equal(err[b], klt[2]) and np.array_equal(it[b], klt[3])
            assert np.array_equal(corners[b], st) and np.array_equal(eig[b], reig) and np.array_equal(mask[b], rmask) and nc[b] == rnc
            assert np.array_equal(X4[b], tri[0], equal_nan=True) and np.array_equal(depth[b], tri[1], equal_nan=True)
            assert np.array_equal(reproj[b], tri[2], equal_nan=True)
            assert np.array_equal(po[b], ba[0]) and np.array_equal(pt[b], ba[1])
            assert bst[b]["cost"] == ba[2]["cost"] and bst[b]["iters"] == ba[2]["iters"] and bst[b]["n_obs"] == ba[2]["n_obs"]
        # resident stepping of the whole batch
        c.upload_sequence(np.stack(seqs)); c.points_upload(np.stack(pts))
        c.dlt_upload(*[np.stack([d[b][k] for b in range(B)]) for k in range(7)])
        c.ba_upload(np.stack([s["K"] for s in scenes]), np.stack([s["poses0"] for s in scenes]),
                    np.stack([s["points0"] for s in scenes]), np.stack([s["obs"] for s in scenes]))
        c.push_frame_resident(0)
        for k, f in enumerate((1, 2, 3, 2)):
            c.frame_step_resident(f, n, ba=c.ba_params(max_iters=5))
            got = c.frame_fetch()
            for b in range(B):
                r = ref[b][5][k]
                assert np.array_equal(got["points2d"][b], r["points2d"]) and np.array_equal(got["status"][b], r["status"])
                assert np.array_equal(got["err"][b], r["err"])
                assert np.array_equal(got["X4"][b], r["X4"], equal_nan=True) and np.array_equal(got["reproj"][b], r["reproj"], equal_nan=True)
                assert np.array_equal(got["poses"][b], r["poses"]) and np.array_equal(got["landmarks"][b], r["landmarks"])
                assert got["ba_stats"][b]["cost"] == r["ba_stats"]["cost"]
                assert np.array_equal(got["corners"][b], r["corners"])
        # the same steps again in the pipelined stream layout with two steps in flight (strided result copies of a batch)
        c.set_tuning(gate_groups=3, reserve_cus=32)       # (the gate + CU mask a batch of >= 8 gets by default)
        c.set_side_stream("pipeline")
        assert c.step_layout() == {"layout": 2, "gate_groups": 3, "reserved_cus": 32}
        c.points_upload(np.stack(pts))
        c.push_frame_resident(0)
        order = (1, 2, 3, 2)
        got_all = []
        c.frame_step_resident(order[0], n, ba=c.ba_params(max_iters=5))
        for f in order[1:]:
            c.frame_step_resident(f, n, ba=c.ba_params(max_iters=5))
            got_all.append(c.frame_fetch())
        got_all.append(c.frame_fetch())
        for k, got in enumerate(got_all):
            for b in range(B):
                r = ref[b][5][k]
                assert np.array_equal(got["points2d"][b], r["points2d"]) and np.array_equal(got["status"][b], r["status"]), (k, b)
                assert np.array_equal(got["err"][b], r["err"]) and np.array_equal(got["X4"][b], r["X4"], equal_nan=True)
                assert np.array_equal(got["poses"][b], r["poses"]) and np.array_equal(got["landmarks"][b], r["landmarks"])
                assert np.array_equal(got["corners"][b], r["corners"]), (k, b)
