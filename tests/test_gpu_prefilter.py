"""GPU: the loader's bilateral pre-filter fused into the level-0 kernel is bit-identical to the
oracle's cv2.bilateralFilter restatement, for pushed and resident frames, and the rest of the path sees the filtered image."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _images():
    from vo_mi355x import synthetic as syn
    rng = np.random.default_rng(3)
    yield syn.make_sequence(1, w=321, h=123, seed=5, margin=32)[0][0]        # smooth texture: most pixels change
    yield rng.integers(0, 256, (97, 203)).astype(np.uint8)                   # noise: colour weights cut everything off
    g = np.tile(np.arange(160, dtype=np.float32), (64, 1))
    yield np.clip(g * 1.5 + rng.normal(0, 2.0, g.shape), 0, 255).astype(np.uint8)   # noisy ramp: ties at .5 exercised
    yield np.full((64, 80), 200, np.uint8)


@pytest.mark.parametrize("d,sc,ss", [(5, 1.5, 1.5), (3, 10.0, 2.0), (7, 25.0, 3.0), (-1, 4.0, 1.2)])
def test_prefilter_level0_matches_oracle(d, sc, ss):
    import vo_oracle as o
    from vo_mi355x import VoContext
    for img in _images():
        h, w = img.shape
        with VoContext(w, h, max_pts=64) as c:
            c.set_prefilter(d, sc, ss)
            c.push_frame(img)
            got, _ = c.pyramid_read(1, 0)
            ref = o.bilateral(img, d, sc, ss)
            assert np.array_equal(got, ref), (d, sc, ss, img.shape, int((got != ref).sum()))
            c.set_prefilter(0)
            c.push_frame(img)
            assert np.array_equal(c.pyramid_read(1, 0)[0], img)


def test_prefilter_feeds_pyramid_klt_and_resident_batch():
    import vo_oracle as o
    from vo_mi355x import VoContext, synthetic as syn
    w, h, n = 640, 240, 400
    fa, _ = syn.make_sequence(3, w=w, h=h, seed=31, margin=64)
    fb, _ = syn.make_sequence(3, w=w, h=h, seed=32, margin=64)
    rng = np.random.default_rng(0)
    fa = np.clip(fa.astype(np.int32) + rng.integers(-6, 7, fa.shape), 0, 255).astype(np.uint8)   # sensor-like noise
    fb = np.clip(fb.astype(np.int32) + rng.integers(-6, 7, fb.shape), 0, 255).astype(np.uint8)
    pts = syn.grid_points(n, w, h, seed=2)
    with VoContext(w, h, max_pts=512, batch=2) as c:
        c.set_prefilter()                                      # the reference's d = 5, sigmas = 1.5
        c.upload_sequence(np.stack([fa, fb]))
        c.points_upload(np.stack([pts, pts]))
        c.push_frame_resident(0)
        c.push_frame_resident(1)
        c.klt_track_resident(n)
        p1, st, err = c.points_download(n)
        for b, fr in enumerate((fa, fb)):
            f0, f1 = o.bilateral(fr[0]), o.bilateral(fr[1])
            assert (f0 != fr[0]).mean() > 0.05                 # the filter does something on this input
            lv = o.build_pyramid(f1)
            for l in range(len(lv)):
                img_l, der_l = c.pyramid_read(1, l, seq=b)
                assert np.array_equal(img_l, lv[l]) and np.array_equal(der_l, o.scharr(lv[l]))
            q1, qs, qe = o.klt(f0, f1, pts)
            assert np.array_equal(p1[b], q1) and np.array_equal(st[b], qs) and np.array_equal(err[b], qe)
