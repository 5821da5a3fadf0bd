"""GPU parity tests: the HIP search path (through the C ABI) against the CPU oracle, same graph,
same seeded inputs.  The bar: ids, p_ids, counts AND f32 distance bit patterns identical.

Shapes follow BASELINE.json's configs at sizes the oracle finishes in seconds.
"""
import numpy as np
import pytest

from conftest import normalized, uniform

pytestmark = pytest.mark.gpu


def build_pair(native, oracle, tmp_path, n, d, m, efc, dist, seed, normalize=False, scale=None, tag="g"):
    """Graph built by the ORACLE's restated insert, dumped in hnswio format, reloaded by the product."""
    X = normalized(n, d, seed) if normalize else uniform(n, d, seed)
    o = oracle.OracleHnsw(m, n, 16, efc, dist)
    if scale is not None:
        o.modify_level_scale(scale)
    o.insert_batch(X)
    o.file_dump(tmp_path, tag)
    h = native.HnswIo(tmp_path, tag).load_hnsw(dist)
    h.upload(0)
    return X, o, h


def assert_same(res, ref):
    assert np.array_equal(res.counts, ref.counts)
    for i in range(len(ref.counts)):
        c = int(ref.counts[i])
        assert np.array_equal(res.ids[i, :c], ref.ids[i, :c]), f"ids differ for query {i}"
        assert np.array_equal(res.dists[i, :c].view(np.uint32), ref.dists[i, :c].view(np.uint32)), f"distance bits differ for query {i}"
        assert np.array_equal(res.layers[i, :c], ref.layers[i, :c])
        assert np.array_equal(res.ranks[i, :c], ref.ranks[i, :c])


CASES = [
    # n, d, M, ef_c, dist, normalize, k, ef, nq      (BASELINE config it scales down)
    (10000, 25, 15, 200, "DistL2", False, 10, 24, 1000),    # config 1 random.rs shape, full size
    (5000, 128, 16, 200, "DistL2", False, 10, 64, 500),     # config 2 SIFT1M shape
    (4000, 25, 24, 400, "DistCosine", False, 10, 128, 400),  # config 3 GloVe-25 shape, cosine
    (4000, 25, 24, 400, "DistDot", True, 10, 128, 400),     # config 3, DistDot on normalised data
    (1500, 784, 32, 400, "DistL2", False, 10, 200, 200),    # config 5 MNIST-784 shape
    (3000, 10, 32, 400, "DistL1", False, 10, 20, 300),      # tests/serpar.rs shape (DistL1)
]


@pytest.mark.parametrize("n,d,m,efc,dist,normalize,k,ef,nq", CASES)
def test_search_matches_oracle(native, oracle, tmp_path, n, d, m, efc, dist, normalize, k, ef, nq):
    X, o, h = build_pair(native, oracle, tmp_path, n, d, m, efc, dist, seed=n + d, normalize=normalize)
    Q = normalized(nq, d, 7) if normalize else uniform(nq, d, 7)
    ref = o.parallel_search(Q, k, ef)
    res = h.parallel_search_flat(Q, k, ef)
    assert_same(res, ref)
    # self queries (tests/equality.rs counts how often a point finds itself; it is not always,
    # on uniform high-d data): same answers as the oracle, and a found self is at distance exactly 0
    res_self = h.parallel_search_flat(X[:200], 1, ef)
    assert_same(res_self, o.parallel_search(X[:200], 1, ef))
    if dist in ("DistL2", "DistL1"):
        found = res_self.ids[:, 0] == np.arange(200)
        assert found.mean() > 0.8
        assert np.all(res_self.dists[found, 0] == 0.0)


def test_k_larger_than_ef_and_short_answers(native, oracle, tmp_path):
    X, o, h = build_pair(native, oracle, tmp_path, 300, 8, 8, 50, "DistL2", seed=3)
    Q = uniform(50, 8, 11)
    for k, ef in ((16, 4), (100, 10), (400, 3)):  # ef = max(ef, k); fewer than k answers when the graph is small
        ref = o.parallel_search(Q, k, ef)
        res = h.parallel_search_flat(Q, k, ef)
        assert_same(res, ref)
    assert res.counts.max() <= 300


def test_sparse_single_point_index(native, oracle, tmp_path):
    """src/hnsw.rs:1870-1881 test_sparse_search: one point (possibly drawn into a layer >= 1)."""
    for seed in range(40):
        X = uniform(1 + seed % 3, 4, 100 + seed)
        o = oracle.OracleHnsw(8, 10, 16, 20, "DistL2")
        for _ in range(seed):  # advance the level stream so the point lands in different layers
            pass
        o.insert_batch(X)
        o.file_dump(tmp_path, "sp")
        h = native.HnswIo(tmp_path, "sp").load_hnsw("DistL2")
        res = h.parallel_search_flat(X[:1], 2, 10)
        ref = o.parallel_search(X[:1], 2, 10)
        assert_same(res, ref)
        assert res.dists[0, 0] == 0.0


def test_serial_equals_batched(native, oracle, tmp_path):
    X, o, h = build_pair(native, oracle, tmp_path, 2000, 16, 12, 100, "DistL2", seed=5)
    Q = uniform(64, 16, 13)
    batched = h.parallel_search(Q, 8, 32)
    for i in range(64):
        assert h.search(Q[i], 8, 32) == batched[i]


def test_device_distances_bit_exact(native, oracle):
    rng = np.random.default_rng(0)
    for dist in ("DistL2", "DistL1", "DistCosine", "DistDot"):
        for d in (1, 3, 25, 128, 784):
            a = (rng.random((256, d), dtype=np.float32) - 0.3) * 3
            b = (rng.random((256, d), dtype=np.float32) - 0.3) * 3
            if dist == "DistDot":
                a, b = a * 0.05, b * 0.05
            got = native.eval_distances(dist, a, b)
            want = np.array([oracle.dist_eval(dist, a[i], b[i]) for i in range(256)], np.float32)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (dist, d)


def test_sqrt_correctly_rounded(native):
    """L2 of a 1-d vector = sqrt(x*x')... use d=1 with b=0: dist = sqrt(a^2) path exercises v_sqrt."""
    rng = np.random.default_rng(1)
    # choose a so that a*a is exactly representable: a = m * 2^e with 12-bit m
    m = rng.integers(1, 4096, 4096).astype(np.float32)
    e = rng.integers(-40, 40, 4096)
    a = (m * np.exp2(e.astype(np.float32))).astype(np.float32)[:, None]
    got = native.eval_distances("DistL2", a, np.zeros_like(a))
    assert np.array_equal(got, np.abs(a[:, 0]))
    # generic: two-term sums, compared with a correctly rounded host sqrt of the same f32 sum
    x = rng.random((4096, 2), dtype=np.float32) * 100
    s = (x[:, 0] * x[:, 0]).astype(np.float32)
    s = (s + (x[:, 1] * x[:, 1]).astype(np.float32)).astype(np.float32)
    got = native.eval_distances("DistL2", x, np.zeros_like(x))
    assert np.array_equal(got, np.sqrt(s))


@pytest.mark.parametrize("bits", [6, 8, 10])
def test_visited_table_overflow_is_exact(native, oracle, tmp_path, bits, knob):
    """A visited table far too small for the query must not change results: a query that outgrows its
    LDS table starts over, inside the same launch, on the exact HBM bitmap (HNSWGPU_HASH_BITS is a
    test/tuning hook that forces a tiny table)."""
    X, o, h = build_pair(native, oracle, tmp_path, 4000, 32, 16, 100, "DistL2", seed=21)
    Q = uniform(300, 32, 22)
    ref = o.parallel_search(Q, 10, 100)
    knob("HNSWGPU_HASH_BITS", str(bits))
    res = h.parallel_search_flat(Q, 10, 100)
    assert_same(res, ref)
    ms, launches = h.last_kernel_ms()
    assert launches >= 1


def test_batch_scheduling_and_exact_first_do_not_change_answers(native, oracle, tmp_path, knob):
    """Batches of >= 256 queries are searched longest-first (estimate + ordering kernels); HNSWGPU_NO_SCHED turns
    that off, HNSWGPU_EXACT_FIRST=1 answers every query with the literal heaps from the start (what the library does
    by itself when most queries of the previous batch met a tie).  Same answers as the oracle in every mode."""
    X, o, h = build_pair(native, oracle, tmp_path, 6000, 48, 16, 100, "DistL2", seed=33)
    Q = uniform(700, 48, 34)
    ref = o.parallel_search(Q, 10, 64)
    assert_same(h.parallel_search_flat(Q, 10, 64), ref)
    knob("HNSWGPU_NO_SCHED", "1")
    assert_same(h.parallel_search_flat(Q, 10, 64), ref)
    knob("HNSWGPU_NO_SCHED", None)
    knob("HNSWGPU_EXACT_FIRST", "1")
    assert_same(h.parallel_search_flat(Q, 10, 64), ref)
    assert h.last_tie_count() <= 700


def _tie_heavy(kind, n, d, seed):
    rng = np.random.default_rng(seed)
    if kind == "duplicates":      # every vector appears twice (distinct ids): all distances tie pairwise
        base = rng.random((n // 2, d), dtype=np.float32)
        X = np.concatenate([base, base])[rng.permutation(n)]
    else:                         # small-integer grid (tests/filtertest.rs:229-241 style): few distinct distances
        X = rng.integers(0, 4, (n, d)).astype(np.float32)
    return np.ascontiguousarray(X)


@pytest.mark.parametrize("kind,dist,ef", [("duplicates", "DistL2", 32), ("grid", "DistL2", 32), ("grid", "DistL1", 100),
                                          ("duplicates", "DistCosine", 32), ("grid", "DistL2", 200), ("duplicates", "DistL2", 64)])
def test_strict_ties_match_the_reference_heap_order(native, oracle, tmp_path, kind, dist, ef):
    """With EQUAL f32 distances the reference's answer depends on its BinaryHeaps' internal order.  The fast
    kernel flags such queries; the strict replay emulates both heaps literally and must agree with the
    oracle (which restates Rust's BinaryHeap) on ids, not only on distances."""
    n, d = 1200, 6
    X = _tie_heavy(kind, n, d, 77)
    o = oracle.OracleHnsw(8, n, 16, 40, dist)
    o.insert_batch(X)
    o.file_dump(tmp_path, "ties")
    h = native.HnswIo(tmp_path, "ties").load_hnsw(dist)
    h.upload(0)
    Q = _tie_heavy(kind, 200, d, 78)
    ref = o.parallel_search(Q, 10, ef)  # ef <= 63 / <= 127 / larger: return_points in 1 / 2 VGPR slots / memory
    h.set_strict_ties(True)
    res = h.parallel_search_flat(Q, 10, ef)
    assert h.last_tie_count() > 20          # the data really produces ties
    assert_same(res, ref)
    # fast mode alone: same distances (as sorted lists), ids may be permuted among equals
    h.set_strict_ties(False)
    fast = h.parallel_search_flat(Q, 10, ef)
    assert np.array_equal(fast.counts, ref.counts)
    agree = np.mean([np.array_equal(fast.dists[i, :c], ref.dists[i, :c]) for i, c in enumerate(ref.counts)])
    assert agree > 0.9


def test_reference_style_c_symbols_search(native, oracle, tmp_path, monkeypatch):
    """search_neighbours_f32 / parallel_search_neighbours_f32 (src/libext.rs:728-767, :205-254): same structs,
    same answers as the oracle."""
    import ctypes as C
    lib = native.lib()
    X = uniform(1500, 12, 31)
    o = oracle.OracleHnsw(12, 1500, 16, 60, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "ffi")
    monkeypatch.chdir(tmp_path)
    io = lib.get_hnswio(3, b"ffi")
    api = lib.load_hnswdump_f32_DistL2(io)
    assert api
    Q = uniform(40, 12, 32)
    ref = o.parallel_search(Q, 6, 30)
    ptrs = (C.c_void_p * 40)(*[Q[i].ctypes.data for i in range(40)])
    res = lib.parallel_search_neighbours_f32(api, 40, 12, ptrs, 6, 30)
    assert res and res.contents.len == 40
    for i in range(40):
        nb = res.contents.ptr[i]
        assert nb.nbgh == ref.counts[i]
        for j in range(nb.nbgh):
            assert nb.neighbours[j].id == ref.ids[i, j] and nb.neighbours[j].d == ref.dists[i, j]
    one = lib.search_neighbours_f32(api, 12, Q[3].ctypes.data, 6, 30)
    assert one and one.contents.nbgh == ref.counts[3]
    assert [one.contents.neighbours[j].id for j in range(6)] == ref.ids[3].tolist()
    lib.hnswgpu_free_neighbourhood(one)
    lib.hnswgpu_free_neighbourhood_vec(res)
    lib.drop_hnsw_f32(api)
    lib.hnswgpu_free_hnswio(io)


EDGE_CASES = [
    # n, d, M, ef_c, dist, k, ef, nq     what it exercises
    (1500, 3, 8, 40, "DistL2", 5, 20, 100),        # d < 4: one padded chunk
    (1500, 33, 8, 40, "DistL2", 5, 20, 100),       # row_stride 64: two passes, tail of zeros
    (1200, 100, 8, 40, "DistCosine", 5, 20, 80),   # 4 passes, d not a multiple of 32, f64 sums
    (1200, 200, 8, 40, "DistDot", 5, 20, 80),      # 7 passes: one full group of 4 + remainder 3
    (2500, 16, 40, 100, "DistL2", 10, 50, 150),    # 2M = 80 neighbour ids per row: multi-batch id rows (> 64)
    (2500, 16, 100, 150, "DistL1", 10, 30, 100),   # 2M = 200: four id batches per expansion
    (3000, 8, 12, 60, "DistL2", 10, 300, 100),     # ef = 300: 16 result slots per lane, memory return_points in replays
    (3000, 8, 12, 60, "DistL2", 100, 1000, 60),    # ef = 1000, k = 100
    (40, 8, 6, 20, "DistL2", 10, 64, 40),          # tiny index: fewer points than ef
]


@pytest.mark.parametrize("n,d,m,efc,dist,k,ef,nq", EDGE_CASES)
def test_shape_edge_cases(native, oracle, tmp_path, n, d, m, efc, dist, k, ef, nq):
    normalize = dist == "DistDot"
    X, o, h = build_pair(native, oracle, tmp_path, n, d, m, efc, dist, seed=n + d + m, normalize=normalize)
    Q = normalized(nq, d, 3) if normalize else uniform(nq, d, 3)
    assert_same(h.parallel_search_flat(Q, k, ef), o.parallel_search(Q, k, ef))
