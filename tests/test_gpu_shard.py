"""GPU: landmark-sharded bundle adjustment of ONE problem (BASELINE config 5).

One GPU is available to the tests, so the partition arithmetic is exercised with the batch dimension of a context acting
as shards (k_ba_xsum / k_ba_xstat sum them exactly where the RCCL all-reduce sums ranks), and the RCCL plumbing with a
1-rank communicator on the same stream (all-reduce / all-gather over one rank = identity, so results must be
bit-identical to the run without a communicator).  The sharded solve must reproduce the unsharded one up to summation
order: same iteration / acceptance sequence, cost to 1e-10, poses and points to 1e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve_unsharded(s, prm_kw):
    from vo_mi355x import VoContext
    with VoContext(64, 64, max_pts=64) as c:
        return c.ba_adjust(s["K"], s["poses0"], s["points0"], s["obs"], c.ba_params(**prm_kw))


def _solve_sharded(s, V, prm_kw, with_comm=False):
    from vo_mi355x import VoContext, sharding
    N = s["points0"].shape[0]
    Ks, po_s, pt_s, ob_s = sharding.shard_problem(s["K"], s["poses0"], s["points0"], s["obs"], V)
    with VoContext(64, 64, max_pts=64, batch=V) as c:
        if with_comm:
            c.comm_init(1, 0, VoContext.comm_unique_id())
        c.ba_set_sharded(True)
        po, pt, st = c.ba_adjust(Ks, po_s, pt_s, ob_s, c.ba_params(**prm_kw))
        gathered = c.ba_gather_points()
    po, pt = np.asarray(po).reshape(V, -1, 6), np.asarray(pt).reshape(V, -1, 3)
    st = st if isinstance(st, list) else [st]
    assert gathered.shape == (1, V, pt.shape[1], 3) and np.array_equal(gathered[0], pt)
    for v in range(1, V):   # every shard holds the same poses and took the same decisions
        assert np.array_equal(po[v], po[0])
        assert all(st[v][k] == st[0][k] for k in ("cost", "cost0", "iters", "accepted", "status", "lam"))
    return po[0], sharding.unshard_points(gathered, N), st[0], sum(x["n_obs"] for x in st)


@pytest.mark.parametrize("V,N,W,vis", [(2, 2000, 10, 1.0), (4, 2001, 10, 0.8), (8, 1500, 10, 0.9), (3, 700, 20, 0.85)])
def test_virtual_shards_match_unsharded(V, N, W, vis):
    from vo_mi355x import synthetic as syn
    s = syn.make_ba_scene(n_pts=N, n_slots=W, seed=7 + V, visibility=vis)
    kw = dict(max_iters=25)
    po, pt, st = _solve_unsharded(s, kw)
    po_s, pt_s, st_s, n_obs = _solve_sharded(s, V, kw)
    assert n_obs == st["n_obs"]
    assert (st_s["iters"], st_s["accepted"], st_s["status"]) == (st["iters"], st["accepted"], st["status"])
    assert abs(st_s["cost"] - st["cost"]) <= 1e-10 * st["cost"] and abs(st_s["cost0"] - st["cost0"]) <= 1e-10 * st["cost0"]
    assert np.abs(po_s - po).max() <= 1e-8 and np.abs(pt_s - pt).max() <= 1e-8
    assert st_s["cost"] < 0.1 * st_s["cost0"]


def test_sharded_cost_matches_oracle():
    import ba_oracle as bo
    from vo_mi355x import synthetic as syn
    s = syn.make_ba_scene(n_pts=1203, n_slots=10, seed=11, visibility=0.9)
    po, pt, st, _ = _solve_sharded(s, 4, dict(max_iters=20))
    assert abs(bo.cost(s["K"], po, pt, s["obs"]) - st["cost"]) <= 1e-9 * st["cost"]
    ref = bo.solve(s["K"], s["poses0"], s["points0"], s["obs"], max_iters=20)
    assert st["iters"] == ref["iters"] and st["status"] == ref["status"]
    assert abs(st["cost"] - ref["cost"]) <= 1e-7 * ref["cost"]


def test_rccl_single_rank_communicator_is_identity():
    from vo_mi355x import synthetic as syn
    s = syn.make_ba_scene(n_pts=1000, n_slots=10, seed=3, visibility=0.9)
    kw = dict(max_iters=15)
    a = _solve_sharded(s, 2, kw, with_comm=False)
    b = _solve_sharded(s, 2, kw, with_comm=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_sharded_resident_solve_and_config5_shape():
    """config 5 shape: 5000 landmarks, 20-frame window, as 8 shards (what 8 ranks would hold), resident solve"""
    from vo_mi355x import VoContext, sharding, synthetic as syn
    K = np.array([[1100.0, 0, 960.0], [0, 1100.0, 540.0], [0, 0, 1]])
    s = syn.make_ba_scene(n_pts=5000, n_slots=20, K=K, seed=1)
    po, pt, st = _solve_unsharded(s, dict(max_iters=12))
    Ks, po_s, pt_s, ob_s = sharding.shard_problem(K, s["poses0"], s["points0"], s["obs"], 8)
    with VoContext(64, 64, max_pts=64, batch=8) as c:
        c.ba_set_sharded(True)
        c.ba_upload(Ks, po_s, pt_s, ob_s)
        c.ba_solve_resident(c.ba_params(max_iters=12))
        po2, _, st2 = c.ba_fetch()
        pts2 = sharding.unshard_points(c.ba_gather_points(), 5000)
    assert (st2[0]["iters"], st2[0]["accepted"]) == (st["iters"], st["accepted"])
    assert abs(st2[0]["cost"] - st["cost"]) <= 1e-10 * st["cost"]
    assert np.abs(po2[0] - po).max() <= 1e-8 and np.abs(pts2 - pt).max() <= 1e-8
