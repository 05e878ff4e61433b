"""The OpenCV boundary against a LIVE cv2 (SURVEY.md 8a' KLT-1..3, ST-1/2, DLT-1; the only route to pin rows a4 / a5 / a6).

* `-m gpu` tests: skipped unless `import cv2` finds a real OpenCV on the GPU box (it is absent from the build image and from
  the stock GPU image; nothing can be installed).  When present they compare the HIP path with OpenCV itself at the call
  sites and parameters of the reference (extractor.py:16-24,44-45,65-66,107,111,270; loader.py:86; bundle_adjuster.py:48).
* the CPU self-test runs the same comparison code with the stub (forwarding to the C oracle) in the role of cv2 and the
  oracle-backed context in the role of the device, so the harness is exercised in every CPU run."""
import importlib.util
import os

import numpy as np
import pytest

import live_cv2 as lv


def _scene():
    from vo_mi355x import synthetic as syn
    frames, _ = syn.make_sequence(2, w=416, h=240, seed=11, margin=64)
    p0 = syn.grid_points(300, 416, 240, margin=10).astype(np.float32)
    s = syn.make_ba_scene(n_pts=200, n_slots=4, seed=2)
    K = s["K"]
    H = []
    for i in (3, 0):
        Hm = np.eye(4); Hm[:3, :3] = syn.rodrigues(s["poses_gt"][i, :3]); Hm[:3, 3] = s["poses_gt"][i, 3:]
        H.append(Hm)
    P0, P1 = np.float32(K @ H[0][:3]), np.float32(K @ H[1][:3])
    return frames, p0, P0, P1, s["obs"][3].astype(np.float32), s["obs"][0].astype(np.float32)


def _run_all(cv2, ctx, loader_ctx, vec_to_mat, mat_to_vec):
    frames, p0, P0, P1, uv0, uv1 = _scene()
    ctx.push_frame(frames[0]); ctx.push_frame(frames[1])
    rep = {}
    if hasattr(ctx, "pyramid_read") and hasattr(cv2, "buildOpticalFlowPyramid"):
        rep["pyramid"] = lv.compare_pyramid(cv2, ctx, frames[1])
    rep["klt"] = lv.compare_klt(cv2, ctx, frames[0], frames[1], p0)
    p1 = ctx.klt_track(p0)[0]
    rep["st"] = lv.compare_shi_tomasi(cv2, ctx, frames[1], p1)
    rep["dlt"] = lv.compare_dlt(cv2, ctx, P0, P1, uv0, uv1)
    rep["bilateral"] = lv.compare_bilateral(cv2, loader_ctx, frames[0])
    rep["rodrigues"] = lv.compare_rodrigues(cv2, vec_to_mat, mat_to_vec)
    return rep


def _assert_contract(rep):
    for lvl, (img_ok, der_ok) in rep.get("pyramid", {}).items():
        assert img_ok and der_ok, "KLT-1 level %d" % lvl
    k = rep["klt"]
    assert k["status_equal"] == 1.0 and k["frac_within_0p01"] >= 0.99 and k["err_rel_p99"] <= 1e-4, k      # KLT-2 / KLT-3
    s = rep["st"]
    assert s.get("eig_rel", 0.0) <= 1e-5 and s.get("mask_equal", True), s                                   # ST-1, mask
    assert s["same_set"] and s["order_mismatches"] <= max(2, s["n_cv"] // 50), s                            # ST-2 (near-tie swaps reported)
    assert rep["dlt"]["max_rel"] <= 1e-4, rep["dlt"]                                                        # DLT-1
    assert rep["bilateral"]["max_diff"] <= 1, rep["bilateral"]
    assert rep["rodrigues"]["max_abs"] <= 1e-9, rep["rodrigues"]


@pytest.mark.gpu
def test_hot_path_against_live_opencv():
    cv2 = lv.find_real_cv2()
    if cv2 is None:
        pytest.skip("no real OpenCV importable on this box: the OpenCV boundary stays 'parity unpinned' (DESIGN.md section 2)")
    from vo_mi355x import VoContext, so3
    with VoContext(416, 240, max_pts=1024) as ctx, VoContext(416, 240, max_pts=64) as lctx:
        rep = _run_all(cv2, ctx, lctx, so3.rodrigues_vec_to_mat, so3.rodrigues_mat_to_vec)
    print("live OpenCV", cv2.__version__, rep)
    _assert_contract(rep)


def test_live_cv2_is_not_shadowed_by_the_stub():
    """conftest no longer puts oracle/ref_stub on sys.path: a plain `import cv2` in a test process must find a real OpenCV
    or nothing -- never our stub"""
    import sys
    assert not any(os.path.abspath(p or ".") == lv.STUB_DIR for p in sys.path)
    mod = lv.find_real_cv2()
    assert mod is None or hasattr(mod, "getBuildInformation")


def test_live_comparison_harness_selftest():
    """the comparison code itself, with (stub cv2 -> C oracle) against the oracle-backed context: every check must come out exact"""
    import vo_oracle as o
    from helpers import ref_stub_cv2
    from oracle_context import OracleContext
    from vo_mi355x import so3
    cv2 = ref_stub_cv2()
    cv2.set_backend(o)
    rep = _run_all(cv2, OracleContext(416, 240), OracleContext(416, 240), so3.rodrigues_vec_to_mat, so3.rodrigues_mat_to_vec)
    _assert_contract(rep)
    assert rep["klt"]["frac_bit_equal"] == 1.0 and rep["st"]["order_mismatches"] == 0 and rep["bilateral"]["max_diff"] == 0
    frames, p0, P0, P1, uv0, uv1 = _scene()
    t = lv.time_reference_call_pattern(cv2, frames[0], frames[1], p0[:100], p0[100:], p0, P0, P1, uv0, uv1, repeats=1)
    assert t["frame_s"] > 0 and t["klt_x4_s"] > 0
