"""GPU: the wave-private bundle-adjustment kernels (csrc/vo_ba_wave.h: windows of <= 10 slots) -- every lane map, panel width and workgroup count
against the oracle, and the running-problem compaction of the tail launch groups.  tests/test_gpu_ba.py runs the goldens through them as well
(and through the lane-per-observation kernels)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,n_pts", [(2, 130), (4, 1001), (5, 700), (7, 333), (8, 1000), (9, 500), (10, 2000), (10, 611)])
def test_ba_wave_private_lane_maps_and_workgroup_counts(monkeypatch, W, n_pts):
    """csrc/vo_ba_wave.h: every lane map (4 lanes per landmark up to W = 4, 8 up to W = 8, 5 at W = 9 and 10 -- and 8 lanes with two slot passes,
    vo_tuning.ba_lanes = 8), every panel width (1-4 column blocks), landmark counts that leave the last chunk partly filled, and workgroup counts from ONE
    (a wave walks every fourth chunk) to one chunk per wave (vo_tuning.ba_workgroups): same LM iteration / acceptance sequence and cost as the oracle, solutions
    equal to 1e-9 between the forms (their summation orders differ)."""
    import ba_oracle as bo
    from vo_mi355x import VoContext, synthetic as syn
    monkeypatch.setattr(VoContext, "default_tuning", {"ba_kernels": 2})   # (ONE sequence with a window of 9-10 slots takes the lane-per-observation kernels by default)
    s = syn.make_ba_scene(n_pts=n_pts, n_slots=W, seed=20 + W, visibility=0.85)
    ref = bo.solve(s["K"], s["poses0"], s["points0"], s["obs"], max_iters=12)
    out = {}
    forms = [("rule", {}), ("g1", {"ba_workgroups": 1}), ("g3", {"ba_workgroups": 3}), ("g999", {"ba_workgroups": 999})]
    if W >= 9:
        forms.append(("lpp8", {"ba_lanes": 8}))
    for key, tune in forms:
        with VoContext(64, 64, max_pts=64) as c:
            c.set_tuning(**tune)
            out[key] = c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c.ba_params(max_iters=12))
    for key, (po, pt, st) in out.items():
        assert st["iters"] == ref["iters"] and st["accepted"] == ref["accepted"] and st["status"] == ref["status"], (key, st, ref["iters"])
        assert abs(st["cost"] - ref["cost"]) <= 1e-7 * ref["cost"], (key, st["cost"], ref["cost"])
        assert np.abs(po - ref["poses"]).max() <= 1e-6 and np.abs(pt - ref["points"]).max() <= 1e-5
        assert np.abs(po - out["rule"][0]).max() <= 1e-9 and np.abs(pt - out["rule"][1]).max() <= 1e-8, key


def test_ba_running_problem_compaction(monkeypatch):
    """64 problems of one batch that need between 2 and 12 LM iterations: once some have finished, the launch groups hand their workgroups to
    the ones still running (`ba2_select_work`: 8 workgroups per problem in a full launch, up to 16 in the tail) -- a problem's partial sums are
    then folded in another order, nothing else may change: every problem = the same problem solved alone (iterations, acceptance sequence,
    status; cost 1e-10; poses 1e-9, points 1e-6 of their distance), and the easy ones really did finish early."""
    from vo_mi355x import VoContext, synthetic as syn
    monkeypatch.setattr(VoContext, "default_tuning", {"ba_kernels": 2})   # (the single context below would take the lane-per-observation kernels otherwise)
    B, N, W = 64, 800, 10
    sc = []
    for b in range(B):
        kind = b % 4
        s = syn.make_ba_scene(n_pts=N, n_slots=W, seed=300 + b, visibility=(1.0, 0.9, 0.7, 0.5)[kind], obs_noise=(0.05, 0.3, 0.5, 1.0)[kind],
                              pt_noise=(0.02, 0.3, 0.6, 1.0)[kind])
        if b % 16 == 0:                                   # already at its minimum's doorstep: the poses and points it was rendered from
            s["poses0"], s["points0"] = s["poses_gt"].copy(), s["points_gt"].copy()
        sc.append(s)
    stack = lambda k: np.stack([s[k] for s in sc])
    with VoContext(64, 64, max_pts=64, batch=B) as c:
        prm = c.ba_params(max_iters=12)
        po, pt, st = c.ba_adjust(stack("K"), stack("poses0"), stack("points0"), stack("obs"), prm)
    its = [x["iters"] for x in st]
    assert min(its) + 2 <= max(its), its                 # the batch had a tail
    with VoContext(64, 64, max_pts=64) as c1:
        for b in range(0, B, 3):
            s = sc[b]
            po1, pt1, st1 = c1.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c1.ba_params(max_iters=12))
            assert (st[b]["iters"], st[b]["accepted"], st[b]["status"]) == (st1["iters"], st1["accepted"], st1["status"]), (b, st[b], st1)
            assert abs(st[b]["cost"] - st1["cost"]) <= 1e-10 * max(st1["cost"], 1e-30), (b, st[b]["cost"], st1["cost"])
            # (points: relative to their distance -- a landmark two frames saw under a small angle moves 1e-6 m along its ray for a change in
            #  the last bits of its 3 x 3 block; the median landmark agrees to 1e-11)
            dp = np.linalg.norm(pt[b] - pt1, axis=1) / np.linalg.norm(pt1, axis=1)
            assert np.abs(po[b] - po1).max() <= 1e-9 and dp.max() <= 1e-6 and np.median(dp) <= 1e-11, (b, dp.max(), np.median(dp))
