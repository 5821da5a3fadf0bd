"""GPU parity tests, second batch: the search's own distance routine over every dimension residue, filtered search
(Hnsw::search_filter with a sorted id vector), the multi-device / concurrent entry points of the C ABI, ef beyond the
register-resident result set, and FULL-SIZE parity (200k x 128 and BASELINE config 5 at its real size, 60k x 784) with
the product builder's graph loaded into both the product and the oracle.
The bar everywhere: ids, p_ids, counts AND f32 distance bit patterns identical to the oracle."""
import threading

import numpy as np
import pytest

from conftest import normalized, probability, uniform
from test_gpu_parity import assert_same, build_pair

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------- distances
SWEEP_D = sorted(set(list(range(1, 131)) + [159, 160, 161, 191, 192, 193, 255, 256, 257, 300, 383, 384, 385, 480, 500, 511, 512,
                                             513, 640, 700, 767, 768, 769, 783, 784, 785, 799, 800]))


@pytest.mark.parametrize("dist", ["DistL2", "DistL1", "DistDot", "DistCosine"])
def test_search_distance_routine_sweep(native, oracle, dist):
    """batch_dist -- the routine the search kernel calls (lane groups of 4 or 2 lanes per row, DPP chain in the group's
    first lane; src/hnsw.rs:1026 `dist_f.eval`) -- for every dimension residue mod 32 up to 800 and for the batch sizes
    that select its branches: 1, 16 (one G = 4 round), 17 and 32 (G = 2), 33 (G = 2 then G = 4), 64 (two G = 2 rounds).
    Bit-compared with the oracle's scalar left-to-right sums.  Tolerance: 0 ulp."""
    rng = np.random.default_rng(11)
    for d in SWEEP_D:
        scale = 0.05 if dist == "DistDot" else 3.0
        q = ((rng.random((3, d), dtype=np.float32) - 0.3) * scale).astype(np.float32)
        rows = ((rng.random((70, d), dtype=np.float32) - 0.3) * scale).astype(np.float32)
        rows[5] = q[0]                      # an exact self distance
        if d > 2:
            rows[6, : d // 2] = 0.0         # zeros inside a row
        want = oracle.dist_matrix(dist, q, rows).view(np.uint32)
        for nf in ((1, 16, 17, 32, 33, 64) if d in (1, 3, 25, 31, 32, 33, 64, 100, 128, 200, 784, 800) else (17, 64)):
            got = native.eval_distance_matrix(dist, q, rows, nf).view(np.uint32)
            assert np.array_equal(got, want), (dist, d, nf, np.argwhere(got != want)[:4].tolist())
    if dist in ("DistL2", "DistL1"):
        assert native.eval_distance_matrix(dist, q, rows, 16)[0, 5] == 0.0


@pytest.mark.parametrize("dist", ["DistHellinger", "DistJeffreys", "DistJensenShannon"])
def test_probability_distances_on_the_device(native, oracle, tmp_path, dist):
    """The f32 distances between probability vectors of the crate's FFI (src/libext.rs:334-345, :491-513): the search's own
    routine against the oracle for every dimension residue (sqrt and ln terms per element: the device carries the same
    restatement of glibc's logf as the oracle), then search parity on a graph, self queries included.  0 ulp."""
    for d in list(range(1, 70)) + [96, 100, 127, 128, 129, 200, 255, 256, 300]:
        rows = probability(70, d, 100 + d) if d > 1 else np.ones((70, 1), np.float32)
        q = probability(3, d, 200 + d) if d > 1 else np.ones((3, 1), np.float32)
        rows[5] = q[0]
        want = oracle.dist_matrix(dist, q, rows).view(np.uint32)
        for nf in ((1, 16, 17, 32, 33, 64) if d in (1, 3, 25, 32, 33, 64, 100, 128) else (17, 64)):
            got = native.eval_distance_matrix(dist, q, rows, nf).view(np.uint32)
            assert np.array_equal(got, want), (dist, d, nf, np.argwhere(got != want)[:4].tolist())
    n, d, m = 3000, 20, 12
    X = probability(n, d, 7)
    o = oracle.OracleHnsw(m, n, 16, 80, dist)
    o.insert_batch(X)
    o.file_dump(tmp_path, "p")
    h = native.HnswIo(tmp_path, "p").load_hnsw(dist)
    h.upload(0)
    Q = probability(300, d, 8)
    assert_same(h.parallel_search_flat(Q, 10, 48), o.parallel_search(Q, 10, 48))
    assert_same(h.parallel_search_flat(X[:100], 3, 30), o.parallel_search(X[:100], 3, 30))
    # GPU-assisted construction, one point per window: the oracle's serial graph byte for byte
    hb = native.Hnsw(m, n, 16, 80, dist)
    hb.set_build_options(nthreads=1, gpu_device=0, gpu_window=1)
    hb.parallel_insert(X[:800])
    hb.file_dump(tmp_path, "g")
    o2 = oracle.OracleHnsw(m, n, 16, 80, dist)
    o2.insert_batch(X[:800])
    o2.file_dump(tmp_path, "o2")
    for ext in (".hnsw.graph", ".hnsw.data"):
        assert open(tmp_path / ("g" + ext), "rb").read() == open(tmp_path / ("o2" + ext), "rb").read()


def test_reference_ffi_arms_of_the_probability_distances(native, oracle, tmp_path, monkeypatch):
    """load_hnswdump_f32_DistJensenShannon / _DistJeffreys (src/libext.rs:334-345) and the DistHellinger / DistJeffreys /
    DistJensenShannon arms of init_hnsw_f32 (:491-513): build through the FFI, dump, reload, search."""
    import ctypes as C
    lib = native.lib()
    monkeypatch.chdir(tmp_path)
    X = probability(600, 10, 3)
    Q = probability(20, 10, 4)
    for dist, loader in (("DistJensenShannon", lib.load_hnswdump_f32_DistJensenShannon), ("DistJeffreys", lib.load_hnswdump_f32_DistJeffreys),
                         ("DistHellinger", None)):
        api = lib.init_hnsw_f32(8, 40, len(dist), dist.encode())
        assert api
        for i in range(600):
            lib.insert_f32(api, 10, X[i].ctypes.data, i)
        assert lib.file_dump_f32(api, 3, b"ffi") == 1
        o = oracle.OracleHnsw(8, 600, 16, 40, dist)
        o.insert_batch(X)
        o.file_dump(tmp_path, "orc")
        assert open("ffi.hnsw.graph", "rb").read() == open("orc.hnsw.graph", "rb").read()
        if loader is not None:
            lib.drop_hnsw_f32(api)
            api = loader(lib.get_hnswio(3, b"ffi"))
            assert api
            assert not lib.load_hnswdump_f32_DistL2(lib.get_hnswio(3, b"ffi"))   # the dump is not a DistL2 one
        ref = o.parallel_search(Q, 5, 20)
        for i in range(20):
            one = lib.search_neighbours_f32(api, 10, Q[i].ctypes.data, 5, 20)
            assert one and one.contents.nbgh == ref.counts[i]
            assert [one.contents.neighbours[j].id for j in range(one.contents.nbgh)] == ref.ids[i, :ref.counts[i]].tolist()
            assert [one.contents.neighbours[j].d for j in range(one.contents.nbgh)] == ref.dists[i, :ref.counts[i]].tolist()
            lib.hnswgpu_free_neighbourhood(one)
        lib.drop_hnsw_f32(api)


# ------------------------------------------------------------------------------------------------- filtered search
def _assert_filtered(h, o, Q, k, ef, allowed):
    res = h.parallel_search_filter_flat(Q, k, ef, allowed)
    for i in range(Q.shape[0]):
        try:
            ids, dd, ll, rr = o.search_filter(Q[i], k, ef, allowed)
        except RuntimeError as e:
            assert "panics" in str(e)
            assert res.status[i] == 1 and res.counts[i] == 0, f"query {i}: the reference panics here"
            continue
        c = len(ids)
        assert res.status[i] == 0 and res.counts[i] == c, f"query {i}: count {res.counts[i]} != {c}"
        assert np.array_equal(res.ids[i, :c], ids), f"query {i}"
        assert np.array_equal(res.dists[i, :c].view(np.uint32), dd.view(np.uint32)), f"query {i}"
        assert np.array_equal(res.layers[i, :c], ll) and np.array_equal(res.ranks[i, :c], rr)
    return res


def test_filter_l2_like_the_reference_test(native, oracle, tmp_path):
    """tests/filtertest.rs:155-219 (filter_l2): 5000 x 25 uniform, M = 15, ef_c = 200, filter = ids 300..400, knbn 10,
    ef 30 -- the device's filtered search against the oracle's, and the reference test's own property: every answer is
    allowed and carries its true distance."""
    X, o, h = build_pair(native, oracle, tmp_path, 5000, 25, 15, 200, "DistL2", seed=41)
    Q = uniform(60, 25, 42)
    allowed = np.arange(300, 400, dtype=np.uint64)
    res = _assert_filtered(h, o, Q, 10, 30, allowed)
    assert res.counts.max() <= 10 and res.counts.min() >= 1
    for i in range(Q.shape[0]):
        c = int(res.counts[i])
        assert np.all((res.ids[i, :c] >= 300) & (res.ids[i, :c] < 400))
        true = np.array([oracle.dist_eval("DistL2", Q[i], X[int(j)]) for j in res.ids[i, :c]], np.float32)
        assert np.array_equal(true.view(np.uint32), res.dists[i, :c].view(np.uint32))
    # a filter vector that is not sorted is refused (the reference's binary search would give garbage)
    with pytest.raises(native.HnswError):
        h.parallel_search_filter_flat(Q[:2], 10, 30, np.array([5, 3, 9], np.uint64))


def test_filter_villsnow_grid(native, oracle, tmp_path):
    """tests/filtertest.rs:225-271 (filter_villsnow): points on a regular grid (every distance ties with many others),
    M = 4; first a filter that allows ONE far-away point, then the all-false filter (must return nothing, and must not
    abort), at several (knbn, ef) including the reference's (10, 4) and (10, 64)."""
    g = 40
    ii, jj = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    X = np.stack([(ii.ravel() + 0.5) / g, (jj.ravel() + 0.5) / g], axis=1).astype(np.float32)
    o = oracle.OracleHnsw(4, g * g, 16, 100, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "grid")
    h = native.HnswIo(tmp_path, "grid").load_hnsw("DistL2")
    h.upload(0)
    Q = np.array([[0.0, 0.0], [0.3, 0.7], [1.0, 1.0], [0.5, 0.5]], np.float32)
    corner = np.array([g * g - 1], np.uint64)          # the point nearest to (1, 1)
    for k, ef in ((10, 4), (10, 64), (1, 1), (3, 2)):
        res = _assert_filtered(h, o, Q, k, ef, corner)
        assert res.counts.max() <= 1
        res = _assert_filtered(h, o, Q, k, ef, np.zeros(0, np.uint64))   # |_id| false
        assert np.all(res.counts == 0)
    # half of the ids, random: ties everywhere, results still identical
    rng = np.random.default_rng(5)
    half = np.sort(rng.choice(g * g, g * g // 2, replace=False)).astype(np.uint64)
    _assert_filtered(h, o, uniform(40, 2, 6), 10, 30, half)


@pytest.mark.parametrize("dist,d,m,k,ef", [("DistCosine", 25, 12, 10, 40), ("DistL1", 10, 8, 5, 12), ("DistDot", 16, 10, 8, 8),
                                           ("DistL2", 8, 6, 1, 1), ("DistL2", 48, 40, 20, 200)])
def test_filtered_search_matches_oracle(native, oracle, tmp_path, dist, d, m, k, ef):
    """Random filters of several densities (1 %, 30 %, everything, nothing, ids that are not in the index), including
    k = ef = 1 where the reference can panic (reported per query, never an abort)."""
    normalize = dist == "DistDot"
    X, o, h = build_pair(native, oracle, tmp_path, 3000, d, m, 60, dist, seed=d + m, normalize=normalize)
    Q = normalized(50, d, 9) if normalize else uniform(50, d, 9)
    rng = np.random.default_rng(d)
    panics = 0
    for frac in (0.01, 0.3, 1.0, 0.0):
        allowed = np.sort(rng.choice(3000, int(3000 * frac), replace=False)).astype(np.uint64)
        allowed = np.unique(np.concatenate([allowed, np.array([3000 + 17, 10 ** 9], np.uint64)])) if frac == 0.3 else allowed
        res = _assert_filtered(h, o, Q, k, ef, allowed)
        panics += int(res.status.sum())
    if ef == 1:
        # without a status array the same call reports the panics as an error code instead of aborting
        import ctypes as C
        allowed = np.zeros(0, np.uint64)
        ids = np.zeros((50, k), np.uint64); dd = np.zeros((50, k), np.float32); cnt = np.zeros(50, np.uint32)
        rc = native.lib().hnswgpu_search_batch_filtered(h.handle, Q.ctypes.data, 50, d, k, ef, None, 0, ids.ctypes.data, dd.ctypes.data,
                                                        None, None, cnt.ctypes.data, None)
        assert rc in (0, native._native.ERR_REF_PANIC)


# ------------------------------------------------------------------------------------------------- multi-device, threads
def test_sharded_entry_point_equals_unsharded(native, oracle, tmp_path):
    """hnswgpu_search_batch_sharded (BASELINE config 4's layout: replicated graph, contiguous balanced shards, answers
    gathered into the caller's arrays) on the one GPU of this box, named two and three times: the shards run
    concurrently on the same replica.  Gathered answers == unsharded answers == oracle."""
    X, o, h = build_pair(native, oracle, tmp_path, 6000, 32, 16, 100, "DistL2", seed=51)
    Q = uniform(1001, 32, 52)        # not divisible by 2 or 3
    ref = o.parallel_search(Q, 10, 64)
    assert_same(h.parallel_search_flat(Q, 10, 64), ref)
    for devices in ([0, 0], [0, 0, 0], [0]):
        assert_same(h.parallel_search_sharded_flat(Q, 10, 64, devices), ref)
    assert_same(h.parallel_search_sharded_flat(Q[:2], 10, 64, [0, 0, 0]), o.parallel_search(Q[:2], 10, 64))  # empty shard


def test_concurrent_searches_on_one_handle(native, oracle, tmp_path):
    """The reference's search is `&self`: several host threads may search one index at the same time.  Every call takes
    its own workspace; answers do not depend on what runs next to them."""
    X, o, h = build_pair(native, oracle, tmp_path, 5000, 24, 12, 80, "DistL2", seed=61)
    batches = [uniform(300 + 37 * t, 24, 70 + t) for t in range(6)]
    refs = [o.parallel_search(b, 8, 48) for b in batches]
    out = [None] * 6
    errs = []

    def work(t):
        try:
            for _ in range(3):
                out[t] = h.parallel_search_flat(batches[t], 8, 48)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for t in range(6):
        assert_same(out[t], refs[t])


def test_ef_beyond_the_register_resident_result_set(native, oracle, tmp_path):
    """ef > 1024 (tests/equality.rs uses ef = 1024, on the edge): served by the literal-heap kernel, same answers."""
    X, o, h = build_pair(native, oracle, tmp_path, 4000, 8, 12, 60, "DistL2", seed=71)
    Q = uniform(30, 8, 72)
    for k, ef in ((10, 1024), (10, 1500), (1200, 40)):
        assert_same(h.parallel_search_flat(Q, k, ef), o.parallel_search(Q, k, ef))


def test_reloaded_index_grows_and_the_replica_follows(native, oracle, tmp_path):
    """insert after reload (src/hnswio.rs:431-524 returns an insertable Hnsw): the HBM replica is refreshed, and the
    searches see the new points exactly like the oracle's reloaded + grown index."""
    X = uniform(3000, 16, 81)
    o = oracle.OracleHnsw(10, 3000, 16, 60, "DistL2")
    o.insert_batch(X[:2000])
    o.file_dump(tmp_path, "part")
    h = native.HnswIo(tmp_path, "part").load_hnsw("DistL2")
    Q = uniform(100, 16, 82)
    assert_same(h.parallel_search_flat(Q, 10, 40), o.parallel_search(Q, 10, 40))
    o2 = oracle.OracleHnsw.load(tmp_path, "part", "DistL2")
    o2.insert_batch(X[2000:], ids=np.arange(2000, 3000))
    h.insert_serial(X[2000:], ids=np.arange(2000, 3000))
    assert h.get_nb_point() == 3000
    res = h.parallel_search_flat(Q, 10, 40)
    assert_same(res, o2.parallel_search(Q, 10, 40))
    assert (res.ids >= 2000).any()


# ------------------------------------------------------------------------------------------------- full size
def _clustered(n, d, seed):
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(0xC0FFEE).random((1000, d), dtype=np.float32)
    c = rng.integers(0, 1000, n)
    return (centres[c] + 0.1 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)


def _full_size_case(native, oracle, tmp_path, knob, n, d, m, efc, k, ef, nq, n_dup):
    X = _clustered(n, d, 0x5EED0001)
    if n_dup:  # exact duplicates (distinct ids): queries near them meet equal f32 distances for certain
        rng = np.random.default_rng(3)
        X[rng.choice(n, n_dup, replace=False)] = X[rng.choice(n, n_dup, replace=False)]
    hb = native.Hnsw(m, n, 16, efc, "DistL2")
    hb.set_build_options(nthreads=0, fast_arithmetic=True)
    hb.parallel_insert(X)          # the product builder, parallel (racy by design, like parallel_insert under Rayon)
    hb.file_dump(tmp_path, "full")
    del hb
    h = native.HnswIo(tmp_path, "full").load_hnsw("DistL2")
    h.upload(0)
    o = oracle.OracleHnsw.load(tmp_path, "full", "DistL2")
    Q = _clustered(nq, d, 0x5EED0002)
    Q[: min(200, n_dup)] = X[np.random.default_rng(4).choice(n, min(200, n_dup), replace=False)]  # some queries ARE points
    ref = o.parallel_search(Q, k, ef)
    res = h.parallel_search_flat(Q, k, ef)
    assert_same(res, ref)
    ties = h.last_tie_count()
    # every query through the literal candidate heap (hand-over at the first pop) and, where equal distances reach the
    # answer, the literal result heap: same answers
    knob("HNSWGPU_EXACT_FIRST", "1")
    assert_same(h.parallel_search_flat(Q, k, ef), ref)
    knob("HNSWGPU_EXACT_FIRST", None)
    # a visited table far too small: most queries migrate to the HBM bitmap (n / 8 bytes per slice) mid-search
    knob("HNSWGPU_HASH_BITS", "8")
    assert_same(h.parallel_search_flat(Q[:1500], k, ef), oracle.SearchResult(ref.ids[:1500], ref.dists[:1500], ref.layers[:1500], ref.ranks[:1500], ref.counts[:1500]))
    knob("HNSWGPU_HASH_BITS", None)
    return ties


def test_full_size_parity_200k_x_128(native, oracle, tmp_path, knob):
    """BASELINE config 2's shape at 200 000 points (18 id bits: the 16-bit visited cells hold 5 + 11 bits, bitmap slices
    of 25 KB), product-built graph, 4 000 queries, with duplicated points so that tie handling is exercised at scale."""
    ties = _full_size_case(native, oracle, tmp_path, knob, 200_000, 128, 16, 200, 10, 64, 4000, 2000)
    assert ties > 0


def test_full_size_parity_config5_60k_x_784(native, oracle, tmp_path, knob):
    """BASELINE config 5 at its real size: 60 000 x 784, M = 32 (64 ids per row), ef = 200 (4 result slots per lane)."""
    _full_size_case(native, oracle, tmp_path, knob, 60_000, 784, 32, 400, 10, 200, 2000, 600)


# ------------------------------------------------------------------------------------------------- construction
@pytest.mark.parametrize("dist,d,m,efc,scale", [("DistL2", 16, 12, 60, None), ("DistCosine", 25, 8, 100, None), ("DistL1", 10, 10, 40, 0.5),
                                                ("DistDot", 12, 6, 250, None)])
def test_gpu_assisted_construction_window_1_equals_the_serial_insertion(native, oracle, tmp_path, dist, d, m, efc, scale):
    """GPU-assisted construction with one point per window: every search_layer of insert_slice (ef = 1 above the point's
    level, ef_construction from its level down, src/hnsw.rs:1114-1197) runs on the device against the graph built so far,
    everything else on the host.  The graph must be the oracle's serially built graph BYTE FOR BYTE -- i.e. every
    candidate heap the device returned equals the reference's search_layer on the same partial graph (ef_construction 250
    uses 4 result slots per lane)."""
    n = 1500
    X = normalized(n, d, 91) if dist == "DistDot" else uniform(n, d, 91)
    o = oracle.OracleHnsw(m, n, 16, efc, dist)
    if scale is not None:
        o.modify_level_scale(scale)
    o.insert_batch(X)
    o.file_dump(tmp_path, "orc")
    h = native.Hnsw(m, n, 16, efc, dist)
    if scale is not None:
        h.modify_level_scale(scale)
    h.set_build_options(nthreads=1, gpu_device=0, gpu_window=1)
    h.parallel_insert(X)
    h.file_dump(tmp_path, "gpu")
    for ext in (".hnsw.graph", ".hnsw.data"):
        assert open(tmp_path / ("orc" + ext), "rb").read() == open(tmp_path / ("gpu" + ext), "rb").read()


def test_gpu_assisted_construction_windows_build_a_graph_as_good(native, oracle, tmp_path):
    """Windows of many points (they do not see each other; the reference's parallel_insert is racy too): the graph is
    not the serial one, but it must search as well -- recall@10 against brute force within a point of the host builder's
    -- and the product's search of it must still equal the oracle's search of the same dump."""
    n, d, k, ef = 60_000, 32, 10, 64
    X = _clustered(n, d, 7)
    Q = _clustered(500, d, 8)
    d2 = (Q.astype(np.float64) ** 2).sum(1)[:, None] + (X.astype(np.float64) ** 2).sum(1)[None, :] - 2.0 * Q.astype(np.float64) @ X.astype(np.float64).T
    gt = np.argsort(d2, axis=1)[:, :k]
    recalls = {}
    for name, opts in (("host", dict(nthreads=0)), ("gpu", dict(nthreads=0, gpu_device=0, gpu_window=4096))):
        h = native.Hnsw(16, n, 16, 200, "DistL2")
        h.set_build_options(fast_arithmetic=False, **opts)
        h.parallel_insert(X)
        assert h.get_nb_point() == n
        h.file_dump(tmp_path, name)
        h.upload(0)
        res = h.parallel_search_flat(Q, k, ef)
        recalls[name] = np.mean([len(set(res.ids[i].tolist()) & set(gt[i].tolist())) / k for i in range(500)])
        if name == "gpu":
            o = oracle.OracleHnsw.load(tmp_path, name, "DistL2")
            assert_same(res, o.parallel_search(Q, k, ef))
    assert recalls["gpu"] > recalls["host"] - 0.01, recalls
    assert recalls["gpu"] > 0.9, recalls


# ------------------------------------------------------------------------------------------------- result-set sizes
@pytest.mark.parametrize("kind", ["uniform", "ties"])
def test_result_set_sizes_around_the_slot_switches(native, oracle, tmp_path, kind):
    """ef = max(ef, k) around every size at which the kernel changes shape: 63 / 64 / 65 (one or two entries of
    return_points per lane; at exactly 64 every lane holds one and the whole-list accept rule has no spare lane), 127 / 128 /
    129, 255 / 256 / 257, k = ef, and an index smaller than ef.  `ties`: small-integer coordinates, so that equal
    distances sit on both sides of the cut at ef again and again (the whole-list accept rule must hand those lists
    back to the one-at-a-time path) -- src/hnsw.rs:1028-1053."""
    from test_gpu_parity import _tie_heavy
    n, d = 2500, 8
    X = uniform(n, d, 5) if kind == "uniform" else _tie_heavy("grid", n, d, 5)
    o = oracle.OracleHnsw(10, n, 16, 60, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "sizes")
    h = native.HnswIo(tmp_path, "sizes").load_hnsw("DistL2")
    h.upload(0)
    Q = uniform(120, d, 6) if kind == "uniform" else _tie_heavy("grid", 120, d, 6)
    most_ties = 0
    for k, ef in ((10, 63), (10, 64), (10, 65), (64, 64), (64, 10), (10, 127), (10, 128), (10, 129), (128, 128), (10, 255),
                  (10, 256), (10, 257), (3, 2), (1, 1)):
        assert_same(h.parallel_search_flat(Q, k, ef), o.parallel_search(Q, k, ef))
        most_ties = max(most_ties, h.last_tie_count())
    if kind == "ties":
        assert most_ties > 10  # the data really puts equal distances at decisive places


def test_one_call_with_100k_queries_and_tiny_dimensions(native, oracle, tmp_path):
    """BASELINE config 4 hands 100 000 queries to the path; here they go through ONE call on one device (the persistent
    grid takes them from its work counter), and every answer is compared.  Then d = 1 and d = 2: one padded chunk per
    row, most distances equal or nearly so."""
    X, o, h = build_pair(native, oracle, tmp_path, 5000, 16, 12, 80, "DistL2", seed=21, tag="big")
    Q = uniform(100_000, 16, 22)
    res = h.parallel_search_flat(Q, 10, 32)
    ref = o.parallel_search(Q, 10, 32)
    assert np.array_equal(res.counts, ref.counts)
    assert np.array_equal(res.ids, ref.ids)
    assert np.array_equal(res.dists.view(np.uint32), ref.dists.view(np.uint32))
    for d in (1, 2):
        X, o, h = build_pair(native, oracle, tmp_path, 1500, d, 8, 40, "DistL2", seed=30 + d, tag=f"d{d}")
        Q = uniform(200, d, 31)
        assert_same(h.parallel_search_flat(Q, 10, 40), o.parallel_search(Q, 10, 40))


def test_begin_end_keeps_two_batches_in_flight_from_one_thread(native, oracle, tmp_path):
    """hnswgpu_search_batch_device_begin / hnswgpu_search_batch_end: the device-buffer call, not waited for.  Three tickets
    in flight from this one thread (own streams and output buffers), ended in another order than begun; then a ticket
    whose call fails (wrong dimension): its status and message arrive at hnswgpu_search_batch_end.
    Device buffers and streams come straight from the HIP runtime the library itself is linked to."""
    import ctypes as C
    X, o, h = build_pair(native, oracle, tmp_path, 5000, 24, 12, 80, "DistL2", seed=91)
    lib = native.lib()
    hip = C.CDLL("libamdhip64.so")
    H2D, D2H = 1, 2

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        return p

    def fetch(p, shape, dtype):
        a = np.zeros(shape, dtype)
        assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), p, C.c_size_t(a.nbytes), D2H) == 0
        return a

    k, ef = 8, 64
    batches = [uniform(400 + 50 * t, 24, 90 + t) for t in range(3)]
    refs = [o.parallel_search(b, k, ef) for b in batches]
    bufs, tickets = [], []
    for b in batches:
        nq = len(b)
        q = dmalloc(b.nbytes)
        assert hip.hipMemcpy(q, b.ctypes.data_as(C.c_void_p), C.c_size_t(b.nbytes), H2D) == 0
        st = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(st)) == 0
        outs = dict(ids=dmalloc(nq * k * 8), d=dmalloc(nq * k * 4), layer=dmalloc(nq * k), rank=dmalloc(nq * k * 4), cnt=dmalloc(nq * 4),
                    stats=dmalloc(nq * 32))
        t = C.c_void_p()
        rc = lib.hnswgpu_search_batch_device_begin(h.handle, q, nq, 24, k, ef, outs["ids"], outs["d"], outs["layer"], outs["rank"], outs["cnt"],
                                                   outs["stats"], st, C.byref(t))
        assert rc == 0 and t.value
        bufs.append((q, st, outs, nq))
        tickets.append(t)
    for i in (2, 0, 1):
        assert lib.hnswgpu_search_batch_end(tickets[i]) == 0
    for (q, st, outs, nq), ref in zip(bufs, refs):
        assert np.array_equal(fetch(outs["cnt"], (nq,), np.uint32), ref.counts.astype(np.uint32))
        assert np.array_equal(fetch(outs["ids"], (nq, k), np.uint64), ref.ids.astype(np.uint64))
        assert np.array_equal(fetch(outs["d"], (nq, k), np.uint32), ref.dists.view(np.uint32))
    # a failing call: the error travels with the ticket
    q, st, outs, nq = bufs[0]
    t = C.c_void_p()
    assert lib.hnswgpu_search_batch_device_begin(h.handle, q, 10, 23, k, ef, outs["ids"], outs["d"], outs["layer"], outs["rank"], outs["cnt"],
                                                 outs["stats"], st, C.byref(t)) == 0
    assert lib.hnswgpu_search_batch_end(t) != 0
    assert "dimension" in native._native.last_error()
    for q, st, outs, nq in bufs:
        for p in [q] + list(outs.values()):
            hip.hipFree(p)
        hip.hipStreamDestroy(st)
