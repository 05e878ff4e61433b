"""Config 1: the reference's main loop -- Loader (src/loader/loader.py) feeding the Pipeline loop (src/pipeline/pipeline.py,
restated in tests/test_gpu_e2e.py) -- through the drop-in classes on a 'parking'-shaped dataset written to disk (640 x 480 PNGs, K.txt with cx = 320 and
cy = 240, poses.txt; the real set is not available offline): file decoding, GPU bilateral pre-filter, SIFT bootstrap and
the per-frame steps, trajectory against the rendered ground truth."""
import numpy as np
import pytest


def _write_parking(tmp_path, n=12):
    from PIL import Image
    from vo_mi355x import synthetic as syn
    frames, K, poses = syn.make_two_plane_sequence(n, w=640, h=480, f=500.0, seed=99)
    K = K.copy(); K[0, 2], K[1, 2] = 319.5, 239.5                    # the renderer's principal point (image centre)
    base = tmp_path / "parking"
    (base / "images").mkdir(parents=True)
    for t, im in enumerate(frames):
        Image.fromarray(im).save(str(base / "images" / ("img_%05d.png" % t)))
    with open(base / "K.txt", "w") as f:
        for row in K:
            f.write(", ".join("%.6f" % v for v in row) + ",\n")     # trailing commas like the VAMR file
    cam_to_world = np.array([np.linalg.inv(P)[:3].reshape(-1) for P in poses])
    np.savetxt(str(base / "poses.txt"), cam_to_world)
    cfg = {"parking": {"path": str(base), "init": [0, 4]}}
    return cfg, frames, K, poses


def _loop_over_loader(ld, make_ctx, n_steps):
    """the reference's Pipeline (pipeline.py:12-176: bootstrap from getInit(), then step() per frame) as restated in
    tests/test_gpu_e2e.py, fed by the drop-in Loader: images through getImage (decode + GPU pre-filter), K from getCamera"""
    from test_gpu_e2e import _run
    t0, t1 = ld.getInit()
    frames = [ld.getFrame(t) for t in range(t1 + n_steps + 1)]
    # Loader poses are camera -> world (the parking file's convention); the loop compares world -> camera
    frames = [(im, np.linalg.inv(P)) for im, P in frames]
    return _run(make_ctx, n_steps, t_init=(t0, t1), frames=frames, K=ld.getCamera())


def _check_run(r, n_steps):
    rot, tra = r["errs"][:, 0], r["errs"][:, 1]
    assert rot.max() <= 0.5 and tra.max() <= 0.3, (rot, tra)
    assert r["n_boot"] >= 150 and all(s[0] >= 100 for s in r["sizes"]) and len(r["sizes"]) == n_steps


def test_loader_reads_the_dataset_layout(tmp_path):
    """host side only: paths sorted, K.txt with trailing commas, 3x4 pose rows -> 4x4, getInit, bounds"""
    from vo_mi355x import Loader
    from vo_mi355x.loader import imread_gray
    cfg, frames, K, poses = _write_parking(tmp_path, n=3)
    ld = Loader("parking", cfg)
    assert len(ld) == 3 and str(ld) == "parking" and ld.getInit() == (0, 4)
    assert np.allclose(ld.getCamera(), K, atol=1e-6) and ld.getCamera().shape == (3, 3)
    assert np.allclose(ld.getPose(2), np.linalg.inv(poses[2]), atol=1e-9) and ld.getPose(0).shape == (4, 4)
    assert np.array_equal(imread_gray(ld.image_paths[1]), frames[1])
    with pytest.raises(AssertionError):
        ld.getPose(3)
    with pytest.raises(Exception):
        Loader("unknown", {"unknown": {"path": str(tmp_path)}})
    # the frame as decoded (no pre-filter), optionally into the caller's buffer (a page-locked one on the GPU path)
    assert np.array_equal(ld.getRawImage(1), frames[1])
    buf = np.zeros((480, 640), np.uint8)
    assert ld.getRawImage(2, out=buf) is buf and np.array_equal(buf, frames[2])
    with pytest.raises(ValueError):
        ld.getRawImage(2, out=np.zeros((480, 641), np.uint8))


@pytest.mark.gpu
def test_loader_prefilter_and_pipeline_run(tmp_path):
    import vo_oracle as o
    from test_adapters import _gpu_ctx
    from vo_mi355x import Loader
    cfg, frames, K, poses = _write_parking(tmp_path)
    ld = Loader("parking", cfg)
    im, pose = ld.getFrame(5)
    assert np.array_equal(im, o.bilateral(frames[5], 5, 1.5, 1.5))            # getImage = imread + bilateralFilter(5, 1.5, 1.5)
    _check_run(_loop_over_loader(ld, _gpu_ctx, 7), 7)                         # frames 5 .. 11
    # the same frame decoded into page-locked memory and pre-filtered while it enters the frame store of a fused host-frame step: level 0 of the
    # frame store = getImage's result (loader.py:86 applied on the device, no read-back, no staging copy)
    from vo_mi355x import VoContext
    with VoContext(640, 480, max_pts=64) as c:
        c.set_prefilter(5, 1.5, 1.5)
        pin = VoContext.host_alloc((480, 640))
        c.push_frame(ld.getRawImage(4, out=pin))
        c.points_upload(np.array([[100.0, 100.0]], np.float32))
        c.frame_step_host(ld.getRawImage(5, out=pin), 1, do_dlt=False, do_ba=False, do_st=False)
        c.frame_fetch()
        assert np.array_equal(c.pyramid_read(1, 0)[0], im)


def test_pipeline_run_cpu_twin(tmp_path):
    from oracle_context import OracleContext
    from test_adapters import _oracle_ctx
    from vo_mi355x import Loader
    cfg, frames, K, poses = _write_parking(tmp_path, n=7)
    ld = Loader("parking", cfg, ctx=OracleContext(640, 480))
    _check_run(_loop_over_loader(ld, _oracle_ctx, 2), 2)


def test_imread_gray_16_bit_and_colour(tmp_path):
    """16-bit grey PNGs scale by >> 8 (cv2.imread's IMREAD_GRAYSCALE), colour PNGs use OpenCV's fixed-point weights"""
    from PIL import Image
    from vo_mi355x.loader import imread_gray
    rng = np.random.default_rng(0)
    g16 = rng.integers(0, 65536, (9, 13), dtype=np.uint16)
    Image.fromarray(g16).save(str(tmp_path / "g16.png"))
    assert np.array_equal(imread_gray(str(tmp_path / "g16.png")), (g16 >> 8).astype(np.uint8))
    rgb = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(str(tmp_path / "c.png"))
    want = ((rgb[..., 0].astype(np.int32) * 4899 + rgb[..., 1].astype(np.int32) * 9617 + rgb[..., 2].astype(np.int32) * 1868 + 8192) >> 14)
    assert np.array_equal(imread_gray(str(tmp_path / "c.png")), want.astype(np.uint8))
