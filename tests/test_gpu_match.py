"""GPU: brute-force 2-NN descriptor matching (matching part) against the numpy oracle:
indices exact (ties to the lower train index), float32 distances to the last bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(d1, d2, batch=1):
    import match_oracle as mo
    from vo_mi355x import VoContext
    with VoContext(64, 64, max_pts=64, batch=batch) as c:
        idx, dist = c.match_knn2(d1, d2)
    for b in range(batch):
        i_o, d_o = mo.knn2(d1[b] if batch > 1 else d1, d2[b] if batch > 1 else d2)
        i_g, d_g = (idx[b], dist[b]) if batch > 1 else (idx, dist)
        assert np.array_equal(i_g, i_o)
        assert np.array_equal(d_g, d_o)
    return idx, dist


@pytest.mark.parametrize("n1,n2,dim,seed", [(1000, 1000, 128, 0), (37, 500, 128, 1), (300, 2, 128, 2), (65, 129, 32, 3), (10, 70, 7, 4)])
def test_knn2_matches_oracle(n1, n2, dim, seed):
    rng = np.random.default_rng(seed)
    d1 = np.floor(rng.gamma(0.6, 30.0, (n1, dim))).astype(np.float32)
    d2 = np.floor(rng.gamma(0.6, 30.0, (n2, dim))).astype(np.float32)
    k = min(n1, n2) // 2
    d2[:k] = d1[:k] + rng.integers(-2, 3, (k, dim)).astype(np.float32)           # true matches
    _check(d1, d2)


def test_knn2_ties_single_train_nan_and_batch():
    rng = np.random.default_rng(5)
    d1 = rng.integers(0, 40, (50, 128)).astype(np.float32)
    d2 = rng.integers(0, 40, (90, 128)).astype(np.float32)
    d2[10] = d2[70] = d2[33] = d1[0]                      # three exact duplicates of the first query: the two lowest indices win
    idx, dist = _check(d1, d2)
    assert list(idx[0]) == [10, 33] and list(dist[0]) == [0.0, 0.0]
    idx, dist = _check(d1, d2[:1])                        # one train descriptor: second slot empty
    assert (idx[:, 0] == 0).all() and (idx[:, 1] == -1).all() and np.isinf(dist[:, 1]).all()
    d2n = d2.copy(); d2n[10, 5] = np.nan                  # a NaN train descriptor is never a neighbour
    idx, _ = _check(d1, d2n)
    assert list(idx[0]) == [33, 70] and not (idx == 10).any()
    _check(np.stack([d1, d1[::-1].copy()]), np.stack([d2, d2 * 0.5]), batch=2)
