"""GPU tests of hnsw_search_pair_kernel (search_pair.inc): search_layer for two queries per wavefront as the first pass of a batch
(HNSWGPU_PAIR_SEARCH=1), the one-query kernels over what it hands back -- against the oracle and against the same call without the
first pass (src/hnsw.rs:922-1064, :1487-1580)."""
import numpy as np
import pytest

from conftest import normalized, uniform
from test_gpu_parity import build_pair
from test_gpu_round6 import _device_call_with_stats

pytestmark = pytest.mark.gpu


def _check(native, oracle, h, o, Q, k, ef, knob, min_first_pass=0.5):
    ref = o.parallel_search(Q, k, ef)
    knob("HNSWGPU_PAIR_SEARCH", "0")
    ids_s, d_s, cnt_s, st_s = _device_call_with_stats(native, h, Q, k, ef)
    knob("HNSWGPU_PAIR_SEARCH", "1")
    ids_p, d_p, cnt_p, st_p = _device_call_with_stats(native, h, Q, k, ef)
    info = h.last_call_info() if hasattr(h, "last_call_info") else None
    knob("HNSWGPU_PAIR_SEARCH", None)
    assert np.array_equal(cnt_s, ref.counts.astype(np.uint32))
    assert np.array_equal(cnt_p, ref.counts.astype(np.uint32))
    bad = [i for i in range(len(Q)) if not np.array_equal(ids_p[i, :cnt_p[i]], ref.ids[i, :cnt_p[i]].astype(np.uint64))]
    assert not bad, f"{len(bad)} queries with other ids than the oracle, first {bad[:5]}: status {st_p[bad[:5], 3]}"
    for i in range(len(Q)):
        c = int(cnt_p[i])
        assert np.array_equal(d_p[i, :c], ref.dists[i, :c].view(np.uint32)), f"distance bits differ for query {i}"
    assert np.array_equal(ids_s, ids_p) and np.array_equal(d_s, d_p)
    # the per-query counters (distances, lists scanned, ids read) are the search's, whichever kernel ran it
    for c in (0, 1, 2):
        diff = np.nonzero(st_p[:, c] != st_s[:, c])[0]
        assert diff.size == 0, f"counter {c} differs for {diff.size} queries, first {diff[:5]}: {st_p[diff[:5], c]} against {st_s[diff[:5], c]}"
    first_pass = float(np.mean(st_p[:, 3] == 0))
    assert first_pass >= min_first_pass, f"only {first_pass:.2f} of the queries were answered by the first pass"
    return first_pass, info


@pytest.mark.parametrize("n,d,m,dist,normalize,k,ef,nq", [
    (20000, 25, 24, "DistCosine", False, 10, 128, 1001),   # config 3's shape; an odd count: the last wavefront's second half idles
    (20000, 25, 24, "DistDot", True, 10, 128, 1000),       # config 3'
    (8000, 128, 16, "DistL2", False, 10, 64, 700),         # config 2's shape
    (6000, 32, 12, "DistCosine", False, 10, 100, 600),     # ef not a power of two; the separate norm array
    (6000, 10, 8, "DistL1", False, 5, 20, 513),
    (5000, 40, 32, "DistL2", False, 10, 10, 640),          # lists of 64 ids, ef == k
    (5000, 16, 4, "DistL2", False, 1, 1, 600),             # ef == 1
])
def test_pair_first_pass_matches_the_oracle(native, oracle, tmp_path, knob, n, d, m, dist, normalize, k, ef, nq):
    X, o, h = build_pair(native, oracle, tmp_path, n, d, m, 100, dist, seed=n + d + m, normalize=normalize)
    Q = normalized(nq, d, 11) if normalize else uniform(nq, d, 11)
    _check(native, oracle, h, o, Q, k, ef, knob)
    # stored points as queries: distance 0 to themselves
    _check(native, oracle, h, o, np.ascontiguousarray(X[:nq]), k, ef, knob, min_first_pass=0.0)


def test_pair_first_pass_hands_back_ties_and_full_tables(native, oracle, tmp_path, knob):
    """Integer grid data (equal distances everywhere) and a visited table far too small: the first pass hands such queries to the
    one-query kernels, the answers stay the reference's."""
    rng = np.random.default_rng(3)
    n, d = 6000, 8
    X = rng.integers(0, 4, size=(n, d)).astype(np.float32)
    o = oracle.OracleHnsw(12, n, 16, 100, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "grid")
    h = native.HnswIo(tmp_path, "grid").load_hnsw("DistL2")
    h.upload(0)
    Q = rng.integers(0, 4, size=(600, d)).astype(np.float32)
    _check(native, oracle, h, o, Q, 10, 48, knob, min_first_pass=0.0)
    X2, o2, h2 = build_pair(native, oracle, tmp_path, 20000, 25, 24, 100, "DistCosine", seed=77, tag="small")
    Q2 = uniform(800, 25, 12)
    knob("HNSWGPU_HASH_BITS", "9")
    _check(native, oracle, h2, o2, Q2, 10, 128, knob, min_first_pass=0.0)
    knob("HNSWGPU_HASH_BITS", None)


@pytest.mark.parametrize("dist,normalize", [("DistCosine", False), ("DistDot", True)])
def test_default_policy_takes_the_first_pass_for_large_strict_cosine_batches(native, oracle, tmp_path, knob, capfd, dist, normalize):
    """Without HNSWGPU_PAIR_SEARCH a strict DistCosine / DistDot batch of >= 40 000 queries on short rows goes through the pair pass
    (search_device.hip, PAIR_SEARCH_AUTO_MIN_QUERIES): same answers as with the pass switched off and as the oracle."""
    X, o, h = build_pair(native, oracle, tmp_path, 6000, 25, 24, 100, dist, seed=91, normalize=normalize, tag="auto")
    Q = normalized(40960, 25, 13) if normalize else uniform(40960, 25, 13)
    k, ef = 10, 128
    ref = o.parallel_search(Q, k, ef)
    knob("HNSWGPU_PAIR_SEARCH", None)
    knob("HNSWGPU_TRACE_LAUNCH", "1")
    capfd.readouterr()
    ids_a, d_a, cnt_a, st_a = _device_call_with_stats(native, h, Q, k, ef)
    assert "pair pass" in capfd.readouterr().err
    knob("HNSWGPU_PAIR_SEARCH", "0")
    ids_s, d_s, cnt_s, st_s = _device_call_with_stats(native, h, Q, k, ef)
    assert "pair pass" not in capfd.readouterr().err
    knob("HNSWGPU_TRACE_LAUNCH", None)
    knob("HNSWGPU_PAIR_SEARCH", None)
    assert np.array_equal(ids_a, ids_s) and np.array_equal(d_a, d_s) and np.array_equal(cnt_a, cnt_s)
    assert np.array_equal(cnt_a, ref.counts.astype(np.uint32))
    assert np.array_equal(ids_a, ref.ids.astype(np.uint64)) and np.array_equal(d_a, ref.dists.view(np.uint32))
    assert np.mean(st_a[:, 3] == 0) > 0.5 and np.mean(st_s[:, 3] == 0) > 0.5
    # (the pass leaves bit 2 of word 6 clear and never reports a literal pop: its queries carry status 0 and flags 0)
    h.set_strict_ties(False)
    ids_l, d_l, cnt_l, st_l = _device_call_with_stats(native, h, Q, k, ef)   # lean calls keep the one-query kernels by default
    h.set_strict_ties(True)
    tie_free = (st_l[:, 7] & 1) == 0
    assert np.array_equal(ids_l[tie_free], ids_a[tie_free])
