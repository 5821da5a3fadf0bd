"""Row f3 of the scope table, oracle first: Hnsw::search_filter with a filter (src/hnsw.rs:1487-1580 and the filter
branches of search_layer, :992-1001, :1037-1050) restated in the oracle, checked with the properties the reference's
own tests assert (tests/filtertest.rs) and against brute force.  The GPU entry points are still filter = None."""
import numpy as np
import pytest

from conftest import uniform


def _l2(X, q):
    return np.sqrt(((X - q) ** 2).sum(1, dtype=np.float64))


def test_binary_heap_retain_known_answers(oracle):
    # hand-derived (std's retain = Vec::retain + rebuild_tail): pushes (2,a)(2,b)(2,c)(1,d) give [2a,2b,2c,1d]; removing
    # a leaves [2b,2c,1d], rebuild (start 0 < tail) keeps it; into_sorted_vec then yields d, c, b
    v, t = oracle.heap_retain([2, 2, 2, 1], [10, 11, 12, 13], [0, 1, 1, 1])
    assert t.tolist() == [13, 12, 11] and v.tolist() == [1, 2, 2]
    # descending pushes never sift: [8..1]; removing 2 (index 6) leaves a one-element tail -> the sift_up branch
    v, t = oracle.heap_retain([8, 7, 6, 5, 4, 3, 2, 1], list(range(8)), [1, 1, 1, 1, 1, 1, 0, 1])
    assert v.tolist() == [1, 3, 4, 5, 6, 7, 8]
    # nothing removed / everything removed
    v, t = oracle.heap_retain([3, 1, 2], [0, 1, 2], [1, 1, 1])
    assert v.tolist() == [1, 2, 3]
    v, t = oracle.heap_retain([3, 1, 2], [0, 1, 2], [0, 0, 0])
    assert len(v) == 0
    # random: retain is a filter on the multiset, whatever the internal order
    rng = np.random.default_rng(0)
    for _ in range(50):
        n = int(rng.integers(1, 200))
        vals = rng.integers(0, 6, n).astype(np.float32)
        keep = rng.integers(0, 2, n).astype(np.uint8)
        v, t = oracle.heap_retain(vals, np.arange(n), keep)
        assert v.tolist() == sorted(vals[keep == 1].tolist())
        assert sorted(t.tolist()) == np.nonzero(keep)[0].tolist()


def test_filter_l2_like_the_reference_test(oracle):
    """tests/filtertest.rs:155-219: filtered search in the full index vs plain search in an index of the allowed points."""
    n, d = 3000, 25
    X = uniform(n, d, 11)
    full = oracle.OracleHnsw(15, n, 16, 200, "DistL2")
    full.insert_batch(X)
    allowed = np.arange(300, 400, dtype=np.uint64)
    small = oracle.OracleHnsw(15, n, 16, 200, "DistL2")
    small.insert_batch(X[300:400], ids=allowed)
    hits = 0
    for qi in range(20):
        q = uniform(1, d, 100 + qi)[0]
        fid, fd, _, _ = full.search_filter(q, 10, 30, allowed)
        rid, rd, _, _ = small.search(q, 10, 30)
        assert len(fid) <= 10 and np.all(np.isin(fid, allowed))
        assert np.all(np.diff(fd) >= 0)
        for i, pid in enumerate(rid):                      # :205-213: same id => same distance
            j = np.nonzero(fid == pid)[0]
            if len(j):
                hits += 1
                assert abs(1.0 - rd[i] / fd[j[0]]) < 1e-5
        # every answer carries the true distance of its point
        assert np.allclose(fd, _l2(X[fid.astype(np.int64)], q), rtol=1e-6)
    assert hits > 0


def test_filter_grid_like_villsnow(oracle):
    """tests/filtertest.rs:225-271 (f32 instead of f64): a one-point filter yields at most one hit, an all-false one none."""
    g = 40
    ii, jj = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    X = np.stack([(ii.ravel() + 0.5) / g, (jj.ravel() + 0.5) / g], axis=1).astype(np.float32)
    o = oracle.OracleHnsw(4, g * g, 16, 100, "DistL2")
    o.insert_batch(X)
    corner = np.nonzero(_l2(X, np.array([1.0, 1.0], np.float32)) < 2e-2)[0].astype(np.uint64)
    assert len(corner) == 1
    ids, dists, _, _ = o.search_filter(np.zeros(2, np.float32), 10, 4, corner)
    assert len(ids) <= 1
    if len(ids):
        assert ids[0] == corner[0]
    ids, _, _, _ = o.search_filter(np.zeros(2, np.float32), 10, 64, np.zeros(0, np.uint64))
    assert len(ids) == 0


def test_filtered_answers_against_brute_force(oracle):
    n, d = 4000, 16
    X = uniform(n, d, 5)
    o = oracle.OracleHnsw(16, n, 16, 200, "DistL2")
    o.insert_batch(X)
    rng = np.random.default_rng(6)
    allowed = np.sort(rng.choice(n, 800, replace=False)).astype(np.uint64)
    found = total = 0
    for qi in range(30):
        q = uniform(1, d, 500 + qi)[0]
        ids, dists, _, _ = o.search_filter(q, 10, 64, allowed)
        assert np.all(np.isin(ids, allowed)) and len(set(ids.tolist())) == len(ids)
        truth = allowed[np.argsort(_l2(X[allowed.astype(np.int64)], q))[:10]]
        found += len(set(ids.tolist()) & set(truth.tolist()))
        total += 10
    assert found / total > 0.6   # a filter keeps the search going until the candidates run out: recall is high
    with pytest.raises(RuntimeError):
        o.search_filter(uniform(1, d, 1)[0], 10, 64, np.array([5, 3], np.uint64))   # unsorted id vector
