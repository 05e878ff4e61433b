"""Config 1: the reference's main loop -- Loader + Pipeline (src/loader/loader.py, src/pipeline/pipeline.py)
-- through the drop-in classes on a 'parking'-shaped dataset written to disk (640 x 480 PNGs, K.txt with cx = 320 and
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


def _check_run(pipe, poses, t1, n_steps):
    unit = np.linalg.norm(poses[t1][:3, 3])
    for k in range(1, n_steps + 2):
        Hk = pipe._state._trajectory[k]
        gt = poses[t1 + k - 1]
        cosang = (np.trace(Hk[:3, :3] @ gt[:3, :3].T) - 1) / 2
        assert np.degrees(np.arccos(np.clip(cosang, -1, 1))) <= 0.5
        assert np.linalg.norm(Hk[:3, 3] - gt[:3, 3] / unit) <= 0.3
    assert len(pipe._state._landmarks) >= 100


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


@pytest.mark.gpu
def test_loader_prefilter_and_pipeline_run(tmp_path):
    import vo_oracle as o
    from vo_mi355x import Loader, Pipeline
    cfg, frames, K, poses = _write_parking(tmp_path)
    ld = Loader("parking", cfg)
    im, pose = ld.getFrame(5)
    assert np.array_equal(im, o.bilateral(frames[5], 5, 1.5, 1.5))            # getImage = imread + bilateralFilter(5, 1.5, 1.5)
    pipe = Pipeline(ld, headless=True)
    assert pipe._t_loader == 4 and pipe._t_step == 1 and len(pipe._state._landmarks) >= 150
    pipe.full_run()                                                           # frames 5 .. 11
    assert pipe._t_loader == len(ld) - 1 and pipe._t_step == 8
    _check_run(pipe, poses, 4, 7)
    assert pipe._bundle_adjuster._ctx is pipe._extractor._ctx


def test_pipeline_run_cpu_twin(tmp_path):
    from oracle_context import OracleContext
    from vo_mi355x import Loader, Pipeline
    cfg, frames, K, poses = _write_parking(tmp_path, n=7)
    ld = Loader("parking", cfg, ctx=OracleContext(640, 480))
    pipe = Pipeline(ld, headless=True, ctx=OracleContext(640, 480))
    pipe.step(); pipe.step()
    _check_run(pipe, poses, 4, 2)
