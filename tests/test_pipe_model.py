"""The table model of the device-resident Pipeline.step (oracle/pipe_oracle.py: object tables + ordered index lists, what
csrc/vo_pipeline.hip implements) against the reference's loop over Python objects (pipeline.py:92-167 restated over the drop-in
classes, tests/pipe_helpers.ObjectLoop), frame by frame and object by object, on the CPU oracle back end: candidates, landmarks,
the dead lists (incl. landmarks resurrected into the window, the entries that share a Landmark / Keypoint object with the dead
list, and the duplicates that follow from it), landmark positions after every adjust, the trajectory."""
import numpy as np
import pytest

import pipe_helpers as ph
from test_adapters import _oracle_ctx


def model_entries(m):
    return dict(cand=[m.entry(None, k) for k in m.cand], lm=[m.entry(l, k) for l, k in zip(m.lm_L, m.lm_K)],
                dead=[m.entry(l, k) for l, k in zip(m.dead_L, m.dead_K)], n_dead_total=len(m.dead_L) + m.n_dead_inert)


def model_sharing(m):
    """same signature as pipe_helpers.sharing_signature, over row indices"""
    dl, dk = {}, {}
    for i, (l, k) in enumerate(zip(m.dead_L, m.dead_K)):
        dl.setdefault(l, i); dk.setdefault(k, i)
    return [(dl.get(l, -1), dk.get(k, -1)) for l, k in zip(m.lm_L, m.lm_K)]


def run_both(make_ctx, n_steps, w=256, h=160, t1=3, ba_window=4, seed=2024, period=24.0, amp=(0.9, 0.25, -0.5), p_tol=0.0):
    import pipe_oracle as po
    sc = ph.scene(t1 + n_steps + 1, w=w, h=h, f=260.0, seed=seed, pose_fn=lambda t: ph.sway_pose(t, amp=amp, period=period))
    ctx_a, ctx_b = make_ctx(w, h), make_ctx(w, h)
    state, t_loader = ph.gt_bootstrap(ctx_a, sc, 0, t1)
    import copy
    loop = ph.ObjectLoop(ctx_a, sc["K"], copy.deepcopy(state), sc["frames"][t_loader], ba_window=ba_window)
    model = po.PipeModel(ctx_b, sc["K"], w, h, cap=4096, params=po.Params(ba_window=ba_window))
    model.seed(state, [], [], 1)
    ctx_b.push_frame(sc["frames"][t_loader])
    ph.compare_lists(loop, model_entries(model), what="seed")
    seen = dict(resurrected=0, shared_L=0, dup=0, new=0, dead=0)
    for s in range(n_steps):
        im = sc["frames"][t_loader + 1 + s]
        loop.step(im)
        model.step(im)
        assert model.status == 0
        what = "step %d" % (s + 2)
        ph.compare_lists(loop, model_entries(model), what=what, p_tol=p_tol)
        # the dead list the model keeps is the filtered one: compare the sharing structure over the state's landmark entries only by
        # whether they share at all (indices into the two dead lists differ by the dropped entries)
        sl, sm = ph.sharing_signature(loop), model_sharing(model)
        assert [(a >= 0, b >= 0) for a, b in sl] == [(a >= 0, b >= 0) for a, b in sm], what
        for t in range(model.t + 1):
            assert np.abs(model.poses[t] - loop.state._trajectory[t]).max() <= (1e-9 if p_tol else 0.0), (what, t)
        assert model.info["n_new"] == loop.info["n_new"] and model.info["n_resurrected"] == loop.info["n_resurrected"]
        seen["resurrected"] += loop.info["n_resurrected"]; seen["new"] += loop.info["n_new"]; seen["dead"] = len(loop.dead)
        seen["shared_L"] += sum(1 for a, b in sl if a >= 0 and b < 0)
        ids = [id(l) for l in loop.state._landmarks]
        seen["dup"] += len(ids) - len(set(ids))
    return seen, loop, model


@pytest.mark.parametrize("ba_window,n_steps", [(4, 9), (10, 10)])
def test_table_model_equals_object_loop_cpu(ba_window, n_steps):
    """window 4 = the reference's own setting (pipeline.py:19), window 10 = BASELINE's: there the reference's resurrection
    quirk compounds (a landmark that dies young is appended again every frame, and again for every copy that dies again)"""
    seen, loop, model = run_both(_oracle_ctx, n_steps, ba_window=ba_window)
    # the run must actually exercise the quirks: dead landmarks resurrected into the window, survivors that keep sharing their
    # Landmark object with the dead list, and (long window) the same Landmark object several times in the state's list
    assert seen["resurrected"] > 0 and seen["new"] > 0 and seen["dead"] > 0 and seen["shared_L"] > 0, seen
    if ba_window > 4:
        assert seen["dup"] > 0 and model.n_dead_inert > 0, seen
    assert len(loop.state._landmarks) >= 30
