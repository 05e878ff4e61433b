"""GPU: SIFT detector / descriptor (feature part) against the numpy oracle that defines the
float32 operation order: scale-space keypoints and 128-d descriptors EXACTLY equal; plus rotation / matching sanity."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _image(w, h, seed, kind=0):
    from vo_mi355x import synthetic as syn
    if kind == 0:
        return np.ascontiguousarray(np.asarray(syn.make_texture(2 * w, 2 * h, seed=seed))[:h, :w]).astype(np.uint8)
    rng = np.random.default_rng(seed)
    if kind == 1:                                             # blocks: hard edges, flat regions, saturated values
        return np.kron(rng.integers(0, 2, (h // 12 + 1, w // 12 + 1)) * 235 + 10, np.ones((12, 12), int))[:h, :w].astype(np.uint8)
    return rng.integers(0, 256, (h, w)).astype(np.uint8)      # white noise: tens of thousands of scale-space extrema


def _compare(img, nfeatures=1000, mask=None):
    import sift_oracle as so
    from vo_mi355x import VoContext
    h, w = img.shape
    with VoContext(w, h, max_pts=64) as c:
        kp, desc = c.sift_detect_compute(img, mask=mask, nfeatures=nfeatures)
    kp_o, desc_o = so.detect_and_compute(img, nfeatures=nfeatures, mask=mask)
    assert kp.shape == kp_o.shape, (kp.shape, kp_o.shape)
    assert np.array_equal(kp, kp_o)
    assert desc.shape == desc_o.shape and np.array_equal(desc, desc_o)
    return kp, desc


@pytest.mark.parametrize("w,h,seed,kind", [(320, 240, 3, 0), (161, 97, 4, 0), (200, 150, 5, 1), (96, 64, 6, 2), (41, 33, 7, 0)])
def test_sift_equals_oracle(w, h, seed, kind):
    kp, desc = _compare(_image(w, h, seed, kind))
    if kind == 0 and w >= 161:
        assert len(kp) >= 100
    assert ((desc >= 0) & (desc <= 255) & (desc == np.rint(desc))).all()
    if len(kp):
        nrm = np.linalg.norm(desc, axis=1)
        assert (np.abs(nrm[nrm > 0] - 512) < 40).all()
        assert (kp[:, 0] >= 0).all() and (kp[:, 0] < w).all() and (kp[:, 1] >= 0).all() and (kp[:, 1] < h).all()
        assert (kp[:, 3] >= 0).all() and (kp[:, 3] < 360).all() and (np.diff(kp[:, 0]) >= 0).all()


def test_sift_nfeatures_mask_flat_and_batch():
    import sift_oracle as so
    from vo_mi355x import VoContext, VoError
    img = _image(240, 180, 11)
    kp_all, _ = _compare(img, nfeatures=0)                              # no limit
    kp_50, _ = _compare(img, nfeatures=50)
    assert 50 <= len(kp_50) < len(kp_all) and kp_50[:, 4].min() >= np.sort(kp_all[:, 4])[-50]
    mask = np.full(img.shape, 255, np.uint8); mask[:, :120] = 0
    kp_m, _ = _compare(img, nfeatures=0, mask=mask)
    assert len(kp_m) and (kp_m[:, 0] + 0.5 >= 120).all()
    with VoContext(240, 180, max_pts=64) as c:
        kp, desc = c.sift_detect_compute(np.full((180, 240), 77, np.uint8))   # flat image: nothing
        assert kp.shape == (0, 6) and desc.shape == (0, 128)
        with pytest.raises(VoError):
            c.sift_detect_compute(img, nfeatures=0, max_out=10)               # capacity is reported, not truncated
    img2 = _image(240, 180, 12)
    with VoContext(240, 180, max_pts=64, batch=2) as c:
        outs = c.sift_detect_compute(np.stack([img, img2]), nfeatures=300)
    for (k, d), im in zip(outs, (img, img2)):
        k_o, d_o = so.detect_and_compute(im, nfeatures=300)
        assert np.array_equal(k, k_o) and np.array_equal(d, d_o)


def test_sift_rotation_and_matching_sanity():
    """the same scene rotated by 90 degrees: keypoints map onto each other and Lowe-ratio matches are geometrically right"""
    from vo_mi355x import Extractor, VoContext
    img = _image(320, 240, 3)
    rot = np.ascontiguousarray(np.rot90(img))
    with VoContext(320, 240, max_pts=64) as c, VoContext(240, 320, max_pts=64) as c2:
        kp, desc = c.sift_detect_compute(img)
        kp2, desc2 = c2.sift_detect_compute(rot)
        ext = Extractor(min_kp_dist=7, ctx=c)
        ms = ext.match(desc, desc2)
    assert len(ms) >= 0.6 * min(len(kp), len(kp2))
    good = sum(1 for m in ms if abs(kp2[m.trainIdx, 0] - kp[m.queryIdx, 1]) < 1.5
               and abs(kp2[m.trainIdx, 1] - (img.shape[1] - 1 - kp[m.queryIdx, 0])) < 1.5)
    assert good >= 0.97 * len(ms)


def test_extract_custom_dropin():
    """Extractor.extract(detector='custom', describe=True) (reference extractor.py:114-131)"""
    from vo_mi355x import Extractor
    img = _image(200, 150, 21)
    ext = Extractor(min_kp_dist=7)
    kps = ext.extract(img, 0, detector='custom', describe=True)
    assert len(kps) > 50 and all(k.des.shape == (128, 1) and k.uv.shape == (2, 1) and k.uv.dtype == np.float32 for k in kps)
    assert all(k.t_first == 0 and k.t_total == 1 and len(k.uv_history) == 1 for k in kps)
    nodes = ext.extract(img, 0, detector='custom', describe=False)
    assert len(nodes) == len(kps) and all(k.des.shape == (1, 1) for k in nodes)
    assert np.array_equal(np.array([k.uv for k in nodes]), np.array([k.uv for k in kps]))
