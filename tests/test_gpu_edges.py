"""GPU: error behaviour of the C ABI (codes + messages, no crash, context stays usable) and degenerate problem shapes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_error_codes_and_context_survives():
    from vo_mi355x import VoContext, VoError, synthetic as syn
    frames, _ = syn.make_sequence(2, w=320, h=240, seed=1, margin=48)
    pts = syn.grid_points(100, 320, 240, seed=1)
    with VoContext(320, 240, max_pts=128) as c:
        with pytest.raises(VoError) as e:
            c.klt_track(pts)                                   # no frames yet
        assert e.value.code == -4 and "two pushed frames" in str(e.value)       # VO_E_STATE
        c.push_frame(frames[0])
        with pytest.raises(VoError) as e:
            c.klt_track(pts)                                   # one frame only
        assert e.value.code == -4
        c.push_frame(frames[1])
        with pytest.raises(VoError) as e:
            c.klt_track(syn.grid_points(200, 320, 240))        # more points than max_pts
        assert e.value.code == -5                              # VO_E_CAPACITY
        with pytest.raises(VoError) as e:
            c.klt_track(pts, c.klt_params(win=32))             # even / too large window
        assert e.value.code == -1                              # VO_E_INVALID
        with pytest.raises(VoError) as e:
            c.frame_step_resident(0, 100)                      # no resident sequence
        assert e.value.code == -4
        with pytest.raises(VoError) as e:
            c.ba_solve_resident()                              # nothing uploaded
        assert e.value.code == -4
        s = syn.make_ba_scene(n_pts=50, n_slots=21, seed=0)
        with pytest.raises(VoError) as e:
            c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"])             # window > 20 slots
        assert e.value.code == -5
        with pytest.raises(VoError) as e:
            c.set_prefilter(9, 1.5, 1.5)                       # diameter > 7
        assert e.value.code == -5
        # the context is still good
        p1, st, err = c.klt_track(pts)
        assert st.sum() > 90
        p_empty, st_empty, err_empty = c.klt_track(np.zeros((0, 2), np.float32))
        assert p_empty.shape == (0, 2) and st_empty.shape == (0,)
    with pytest.raises(VoError):
        VoContext(320, 240, device=99)                         # no such device: fails loudly, no fallback


def test_ba_degenerate_shapes():
    import ba_oracle as bo
    from vo_mi355x import VoContext, synthetic as syn
    with VoContext(64, 64, max_pts=64) as c:
        # one landmark, two frames; a landmark seen once; a frame that sees nothing; zero iterations
        s = syn.make_ba_scene(n_pts=1, n_slots=2, seed=3)
        po, pt, st = c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c.ba_params(max_iters=8))
        ref = bo.solve(s["K"], s["poses0"], s["points0"], s["obs"], max_iters=8)
        assert st["iters"] == ref["iters"] and abs(st["cost"] - ref["cost"]) <= 1e-9 * max(ref["cost"], 1e-12) + 1e-15
        s = syn.make_ba_scene(n_pts=40, n_slots=5, seed=4)
        obs = s["obs"].copy()
        obs[1:, 7] = np.nan            # landmark 7 observed once
        obs[:, 9] = np.nan             # landmark 9 never observed
        obs[3] = np.nan                # slot 3 observes nothing
        po, pt, st = c.ba_adjust(s["K"], s["poses0"], s["points0"], obs, c.ba_params(max_iters=10))
        ref = bo.solve(s["K"], s["poses0"], s["points0"], obs, max_iters=10)
        assert st["n_obs"] == int((~np.isnan(obs[..., 0])).sum())
        assert (st["iters"], st["status"]) == (ref["iters"], ref["status"]) and abs(st["cost"] - ref["cost"]) <= 1e-7 * ref["cost"]
        assert np.array_equal(pt[9], s["points0"][9]) and np.array_equal(po[3], s["poses0"][3])   # untouched: no information
        assert np.isfinite(po).all() and np.isfinite(pt).all()
        po0, pt0, st0 = c.ba_adjust(s["K"], s["poses0"], s["points0"], obs, c.ba_params(max_iters=0))
        assert np.array_equal(po0, s["poses0"]) and np.array_equal(pt0, s["points0"]) and st0["iters"] == 0


def test_klt_points_outside_and_on_the_border():
    import vo_oracle as o
    from vo_mi355x import VoContext, synthetic as syn
    w, h = 320, 240
    frames, _ = syn.make_sequence(2, w=w, h=h, seed=8, margin=48)
    pts = np.array([[-40.0, 10.0], [0.0, 0.0], [w - 1.0, h - 1.0], [w + 50.0, h / 2], [w / 2, -33.0], [5.5, h - 0.25],
                    [1e6, 1e6], [-1e6, 12.0], [w / 2, h / 2]], np.float32)
    with VoContext(w, h, max_pts=64) as c:
        c.push_frame(frames[0]); c.push_frame(frames[1])
        p1, st, err, it = c.klt_track(pts, return_iters=True)
        q1, qs, qe, qi = o.klt(frames[0], frames[1], pts, return_iters=True)
        assert np.array_equal(p1, q1) and np.array_equal(st, qs) and np.array_equal(err, qe) and np.array_equal(it, qi)
        assert st[0] == 0 and st[6] == 0 and st[-1] == 1


def test_shi_tomasi_plateaus_every_pixel_a_local_maximum():
    """block images give large exactly-flat regions of the eigenvalue map; under OpenCV's `value == dilated` rule every pixel of
    a plateau is a candidate, so a 256 x 8 NMS tile can yield 2048 of them (regression: its LDS list held half of that)"""
    import vo_oracle as o
    from vo_mi355x import VoContext
    rng = np.random.default_rng(1)
    w, h = 500, 400
    img = np.kron(rng.integers(0, 2, (h // 16 + 1, w // 16 + 1)) * 200.0 + 20, np.ones((16, 16)))[:h, :w].astype(np.uint8)
    for B in (1, 3):
        with VoContext(w, h, max_pts=128, batch=B) as c:
            c.push_frame(np.stack([img] * B) if B > 1 else img)
            corners = c.shi_tomasi(None, 7, params=c.st_params(max_corners=300, quality_level=0.03, min_distance=7.0, block_size=31))
            eig, mask, nc = c.shi_tomasi_read()
        ref, reig, rnc = o.good_features(img, np.full((h, w), 255, np.uint8), maxCorners=300, qualityLevel=0.03, minDistance=7.0,
                                         blockSize=31, return_aux=True)
        assert rnc > 16384
        for b in range(B):
            assert np.array_equal(eig[b] if B > 1 else eig, reig) and int(nc[b] if B > 1 else nc) == rnc
            assert np.array_equal(np.asarray(corners[b] if B > 1 else corners).reshape(-1, 2), ref.reshape(-1, 2))


def test_unpaired_landmark_half_and_batched_row_reads_are_refused():
    """VO_PIPE_TRACK_LANDMARKS alone is the second half of a VO_PIPE_TRACK | VO_PIPE_TRACK_CANDIDATES call (extractor.py:38-59 then :61-88): without
    that call before it the point buffer holds stale positions -> VO_E_STATE; the row / inlier read-backs are single-sequence calls"""
    from vo_mi355x import VoContext, VoError, State, Trajectory, synthetic as syn
    from vo_mi355x.resident import ResidentPipeline, TRACK, TRACK_CANDIDATES, TRACK_LANDMARKS
    frames, _ = syn.make_sequence(3, w=320, h=240, seed=2, margin=48)
    K = np.array([[300.0, 0, 160], [0, 300.0, 120], [0, 0, 1]])
    with VoContext(320, 240, max_pts=256) as c:
        rp = ResidentPipeline(c, K)
        rp.seed(State([], [], [], Trajectory({0: np.eye(4)})), [], [], t_step=0)
        c.push_frame(frames[0]); c.push_frame(frames[1])
        with pytest.raises(VoError) as e:
            rp.step(-1, TRACK_LANDMARKS)                       # nothing tracked yet
        assert e.value.code == -4
        with pytest.raises(VoError) as e:
            rp.step(-1, TRACK_CANDIDATES)                      # the candidates' half never comes without the tracking
        assert e.value.code == -4
        rp.step(-1, TRACK | TRACK_CANDIDATES); rp.fetch()
        rp.step(-1, TRACK_LANDMARKS); rp.fetch()               # the pair, as the reference calls it
        with pytest.raises(VoError) as e:
            rp.step(-1, TRACK_LANDMARKS)                       # ... and only once per tracked set
        assert e.value.code == -4
        c.push_frame(frames[2])
        rp.step(-1, TRACK); rp.fetch()                         # the context is still good
    with VoContext(320, 240, max_pts=256, batch=2) as c2:
        rp2 = ResidentPipeline(c2, np.stack([K, K]))
        with pytest.raises(ValueError):
            rp2.read_rows("K", [0, 1])
        with pytest.raises(ValueError):
            rp2.read_inliers(4)


def test_host_frame_and_tuning_error_paths():
    """round-6 entry points: bad arguments are refused with a code and a message, nothing is enqueued, the context stays usable"""
    import ctypes as C
    from vo_mi355x import VoContext, VoError, synthetic as syn
    from vo_mi355x.resident import ResidentPipeline
    w, h = 320, 240
    frames, _ = syn.make_sequence(3, w=w, h=h, seed=2, margin=48)
    pts = syn.grid_points(100, w, h, seed=1)
    with VoContext(w, h, max_pts=128, batch=2) as c:
        c.points_upload(np.stack([pts, pts]))
        with pytest.raises(VoError) as e:
            c.frame_step_host([frames[1], frames[1]], 100, do_dlt=False, do_ba=False, do_st=False)       # no frame pushed yet
        assert e.value.code == -4
        c.push_frame(np.stack([frames[0], frames[0]]))
        with pytest.raises(ValueError):
            c.frame_step_host([frames[1]], 100)                                  # one image for a batch of two
        with pytest.raises(ValueError):
            c.frame_step_host([frames[1], frames[1][:, ::2]], 100)               # wrong shape
        with pytest.raises(ValueError):
            c.frame_step_host([frames[1], np.asfortranarray(frames[1])], 100)    # rows not contiguous
        ptrs = (C.c_void_p * 2)(frames[1].ctypes.data, None)
        with pytest.raises(VoError) as e:
            c.frame_step_host((ptrs, w, [frames[1]]), 100, do_dlt=False, do_ba=False, do_st=False)      # a null image pointer
        assert e.value.code == -1 and "null frame pointer" in str(e.value)
        ptrs = (C.c_void_p * 2)(frames[1].ctypes.data, frames[1].ctypes.data)
        with pytest.raises(VoError) as e:
            c.frame_step_host((ptrs, w - 1, [frames[1]]), 100, do_dlt=False, do_ba=False, do_st=False)  # stride < width
        assert e.value.code == -1
        with pytest.raises(VoError) as e:
            c.frame_step_host([frames[1], frames[1]], 100, do_dlt=True, do_ba=False, do_st=False)       # DLT without an upload
        assert e.value.code == -4
        # tuning: unknown field, out of range, experiment-only field
        with pytest.raises(ValueError):
            c.set_tuning(no_such_field=1)
        for bad in (dict(ba_kernels=3), dict(ba_lanes=5), dict(ba_threads=128), dict(klt_waves=7), dict(reserve_cus=256), dict(gate_groups=-2), dict(gather_workgroups=-1)):
            with pytest.raises(VoError) as e:
                c.set_tuning(**bad)
            assert e.value.code == -1, bad
        assert all(v == 0 for v in c.tuning().values())                         # a refused call changes nothing
        # still good: two host steps, results of the second = the resident call on the same frame
        c.frame_step_host([frames[1], frames[1]], 100, do_dlt=False, do_ba=False, do_st=False)
        c.frame_step_host([frames[2], frames[2]], 100, do_dlt=False, do_ba=False, do_st=False)
        with pytest.raises(VoError) as e:
            c.frame_step_host([frames[1], frames[1]], 100, do_dlt=False, do_ba=False, do_st=False)      # a third step without a fetch
        assert e.value.code == -4
        with pytest.raises(VoError) as e:
            c.set_tuning(klt_waves=5)                                            # steps in flight
        assert e.value.code == -4
        c.frame_fetch(); got = c.frame_fetch()
    with VoContext(w, h, max_pts=128, batch=2) as c:
        c.push_frame(np.stack([frames[0], frames[0]])); c.push_frame(np.stack([frames[1], frames[1]]))
        p1, _, _ = c.klt_track(np.stack([pts, pts]))
        c.push_frame(np.stack([frames[2], frames[2]]))
        p2, st2, err2 = c.klt_track(p1)
    assert np.array_equal(got["points2d"], p2) and np.array_equal(got["status"], st2) and np.array_equal(got["err"], err2)
    # the closed loop's host step needs TRACK among its stages
    with VoContext(w, h, max_pts=256) as c:
        rp = ResidentPipeline(c, syn.KITTI_K, ba_max_iters=4)
        with pytest.raises(VoError) as e:
            rp.step_host([frames[1]], stages=0)
        assert e.value.code == -1
