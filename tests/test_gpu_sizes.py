"""GPU: odd image sizes through the vectorised kernels (4 pixels per thread pyramid, 5-wide sliding row sums, 10x258 NMS
tiles, 1280-column strips): pyramid / Scharr / KLT / eigen map / corner list bit-exact against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(97, 65), (130, 81), (257, 129), (1283, 67), (1281, 40), (255, 255), (66, 300), (1026, 131)])
def test_front_end_bit_exact_on_odd_sizes(w, h):
    import vo_oracle as o
    from vo_mi355x import VoContext, synthetic as syn
    frames, _ = syn.make_sequence(2, w=w, h=h, seed=w * 7 + h, margin=40)
    n = 150
    pts = syn.grid_points(n, w, h, seed=3, margin=4)
    with VoContext(w, h, max_pts=256) as c:
        c.push_frame(frames[0]); c.push_frame(frames[1])
        lv = o.build_pyramid(frames[1])
        for l in range(len(lv)):
            img_l, der_l = c.pyramid_read(1, l)
            assert np.array_equal(img_l, lv[l]) and np.array_equal(der_l, o.scharr(lv[l])), (w, h, l)
        p1, st, err, it = c.klt_track(pts, return_iters=True)
        q1, qs, qe, qi = o.klt(frames[0], frames[1], pts, return_iters=True)
        assert np.array_equal(p1, q1) and np.array_equal(st, qs) and np.array_equal(err, qe) and np.array_equal(it, qi)
        for bs, md in ((31, 7.0), (15, 4.0), (3, 1.0)):
            if min(w, h) <= bs:
                continue
            corners = c.shi_tomasi(p1, 5, params=c.st_params(max_corners=300, quality_level=0.02, min_distance=md, block_size=bs))
            eig, mask, nc = c.shi_tomasi_read()
            m = np.full((h, w), 255, np.uint8)
            for x, y in np.int32(p1):
                o.circle_mask(m, (x, y), 5, 0)
            ref, aux_eig, ncr = o.good_features(frames[1], m, maxCorners=300, qualityLevel=0.02, minDistance=md, blockSize=bs, return_aux=True)
            assert np.array_equal(mask, m) and np.array_equal(eig, aux_eig) and nc == ncr, (w, h, bs)
            assert np.array_equal(corners, ref), (w, h, bs)


@pytest.mark.parametrize("w,h,win", [(34, 34, 5), (47, 33, 7), (36, 90, 5), (33, 200, 9), (64, 35, 5), (17, 17, 3), (16, 40, 3), (15, 15, 3),
                                     (23, 9, 3), (130, 12, 5), (12, 130, 5)])
def test_pyramid_and_klt_on_tiny_images(w, h, win):
    """the pyramid kernels deal interior and border (row, group) pairs to separate index ranges (vo_split_index, csrc/vo_frame.hip): sizes where a
    padded row has few or NO interior groups, with windows small enough that such images keep several levels -- levels, 4x Scharr and the tracker
    (points at the image corners included) bit-exact against the oracle"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tiny_sizes_check as t
    levels, pyramid_ok, klt_ok = t.check(w, h, win)
    assert levels >= 2 and pyramid_ok and klt_ok, (w, h, win, levels, pyramid_ok, klt_ok)
