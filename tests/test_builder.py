"""Product-side construction (hnswlib-rs_amd/csrc/builder.cpp) against the oracle's
literal restatement of insert_slice / select_neighbours / reverse_update: serial insertion must give the
same graph edge for edge, stored f32 distances included (compared through the dump bytes)."""
import filecmp
import os

import numpy as np
import pytest

from conftest import normalized, probability, uniform


def dumps_equal(d, a, b):
    return (filecmp.cmp(os.path.join(d, a + ".hnsw.graph"), os.path.join(d, b + ".hnsw.graph"), shallow=False)
            and filecmp.cmp(os.path.join(d, a + ".hnsw.data"), os.path.join(d, b + ".hnsw.data"), shallow=False))


CASES = [
    # n, d, M, ef_c, dist, normalize, scale, extend, keep_pruned
    (2500, 25, 15, 100, "DistL2", False, None, False, False),   # examples/random.rs shape
    (2000, 10, 10, 25, "DistL1", False, None, False, False),    # hnsw.rs unit-test shape
    (1500, 25, 24, 200, "DistCosine", False, None, False, False),
    (1500, 25, 16, 100, "DistDot", True, None, False, False),   # tests/serpar.rs: DistDot on normalised data
    (2000, 10, 32, 128, "DistL2", False, 0.5, False, False),    # tests/equality.rs: modify_level_scale(0.5)
    (1200, 8, 6, 40, "DistL2", False, None, True, False),       # set_extend_candidates(true)
    (1200, 8, 6, 40, "DistL2", False, None, False, True),       # set_keeping_pruned(true)
    (1200, 12, 8, 40, "DistHellinger", "prob", None, False, False),  # the distances between probability vectors
    (1200, 12, 8, 40, "DistJeffreys", "prob", None, False, False),
    (1000, 9, 10, 60, "DistJensenShannon", "prob", None, True, False),
]


@pytest.mark.parametrize("n,d,m,efc,dist,normalize,scale,extend,keep", CASES)
def test_serial_insert_matches_oracle(native, oracle, tmp_path, n, d, m, efc, dist, normalize, scale, extend, keep):
    X = probability(n, d, n + m) if normalize == "prob" else normalized(n, d, n + m) if normalize else uniform(n, d, n + m)
    o = oracle.OracleHnsw(m, n, 16, efc, dist)
    h = native.Hnsw(m, n, 16, efc, dist)
    if scale is not None:
        o.modify_level_scale(scale)
        h.modify_level_scale(scale)
    o.set_extend_candidates(extend)
    h.set_extend_candidates(extend)
    o.set_keeping_pruned(keep)
    h.set_keeping_pruned(keep)
    o.insert_batch(X)
    h.insert_serial(X)
    o.file_dump(tmp_path, "orc")
    h.file_dump(tmp_path, "prod")
    assert dumps_equal(tmp_path, "orc", "prod")


def test_parallel_insert_gives_a_valid_graph(native, oracle, tmp_path):
    """parallel_insert is racy by design (src/hnsw.rs:1222-1223): check invariants, not equality."""
    n, d, m = 6000, 16, 12
    X = uniform(n, d, 9)
    h = native.Hnsw(m, n, 16, 100, "DistL2")
    h.set_build_options(nthreads=8)
    h.parallel_insert(X)
    assert h.get_nb_point() == n
    lv = oracle.levels(m, n)  # levels are drawn in input order whatever the thread interleaving
    for l in range(16):
        assert h.get_layer_nb_point(l) == int((lv == l).sum())
    ep_origin, (ep_layer, _) = h.get_entry_point()
    assert ep_layer == lv.max()
    total = 0
    for layer in range(3):
        for rank in range(0, h.get_layer_nb_point(layer), 37):
            for l in range(layer + 1):
                ids, _, _, dists = h.get_neighbours(layer, rank, l)
                assert len(ids) <= (2 * m if l == 0 else m)      # src/hnsw.rs:1272-1283
                assert len(set(ids.tolist())) == len(ids)
                assert np.all(np.diff(dists) >= 0)               # lists stay sorted by stored distance
                total += len(ids)
    assert total > 0
    # it is searchable by the oracle (through the dump) with sane recall
    h.file_dump(tmp_path, "par")
    o = oracle.OracleHnsw.load(tmp_path, "par", "DistL2")
    Q = uniform(100, d, 10)
    r = o.parallel_search(Q, 10, 64, 4)
    D = ((Q[:, None, :] - X[None]) ** 2).sum(-1)
    gt = np.argsort(D, axis=1)[:, :10]
    rec = np.mean([len(set(gt[i]) & set(r.ids[i].tolist())) / 10 for i in range(100)])
    assert rec > 0.9


def test_parallel_insert_with_lock_free_readers_keeps_edges_whole(native, oracle, tmp_path):
    """The builder's searches copy neighbour lists under a sequence counter, without a lock, while other threads rewrite
    them: an edge must never come out torn.  Many more threads than cores, then every sampled (neighbour, distance) pair
    is checked against the vectors."""
    n, d, m = 8000, 8, 8
    X = uniform(n, d, 12)
    for nthreads in (32, 13):
        h = native.Hnsw(m, n, 16, 60, "DistL2")
        h.set_build_options(nthreads=nthreads)
        h.parallel_insert(X)
        assert h.get_nb_point() == n
        checked = 0
        lv = oracle.levels(m, n)  # ranks inside a layer follow the input order
        for layer in range(2):
            members = np.nonzero(lv == layer)[0]
            for rank in range(0, h.get_layer_nb_point(layer), 11):
                ids, _, _, dists = h.get_neighbours(layer, rank, 0)
                ids = ids.astype(np.int64)
                me = int(members[rank])
                assert np.all(ids < n) and me not in set(ids.tolist())
                true = np.sqrt(((X[ids].astype(np.float64) - X[me].astype(np.float64)) ** 2).sum(-1))
                assert np.allclose(dists, true, rtol=1e-5, atol=1e-6)
                checked += len(ids)
        assert checked > 1000


def test_fast_arithmetic_build_is_searchable(native, oracle, tmp_path):
    X = uniform(3000, 32, 4)
    h = native.Hnsw(16, 3000, 16, 100, "DistL2")
    h.set_build_options(nthreads=4, fast_arithmetic=True)
    h.parallel_insert(X)
    h.file_dump(tmp_path, "fast")
    o = oracle.OracleHnsw.load(tmp_path, "fast", "DistL2")
    r = o.parallel_search(X[:200], 1, 32, 4)
    assert (r.ids[:, 0] == np.arange(200)).mean() > 0.95


def test_bad_parameters(native):
    with pytest.raises(native.HnswError):
        native.Hnsw(257, 10, 16, 20, "DistL2")  # the reference exits the process (src/hnsw.rs:784-787)
    h = native.Hnsw(8, 10, 16, 20, "DistL2")
    with pytest.raises(native.HnswError):
        h.parallel_insert(np.zeros((3,), np.float32))


@pytest.mark.parametrize("dist", ["DistL2", "DistCosine"])
def test_reloaded_index_keeps_growing_like_the_oracle(native, oracle, tmp_path, dist):
    """HnswIo::load_hnsw returns a fully insertable Hnsw (its layer generator is rebuilt from the dumped scale,
    src/hnswio.rs:773-777; extend_candidates = true, keep_pruned = false, :510-511).  The product seeds its builder from the
    loaded graph: inserting the same points serially into the reloaded index gives the oracle's bytes."""
    X = uniform(700, 12, 5)
    first, more = X[:500], X[500:]
    o = oracle.OracleHnsw(8, 700, 16, 40, dist)
    o.insert_batch(first)
    o.file_dump(tmp_path, "half")
    o2 = oracle.OracleHnsw.load(tmp_path, "half", dist)
    o2.insert_batch(more, ids=np.arange(500, 700))
    assert o2.get_nb_point() == 700
    o2.file_dump(tmp_path, "orc_full")
    h = native.HnswIo(tmp_path, "half").load_hnsw(dist)
    h.insert_serial(more, ids=np.arange(500, 700))
    assert h.get_nb_point() == 700
    h.file_dump(tmp_path, "prod_full")
    assert dumps_equal(tmp_path, "orc_full", "prod_full")
    # the reference-style FFI inserts into a reloaded handle as well (it used to drop the points silently)
    import ctypes as C
    lib = native.lib()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        api = lib.load_hnswdump_f32_DistL2(lib.get_hnswio(4, b"half")) if dist == "DistL2" else lib.load_hnswdump_f32_DistCosine(lib.get_hnswio(4, b"half"))
        assert api
        for i in range(200):
            lib.insert_f32(api, 12, more[i].ctypes.data, 500 + i)
        assert lib.hnswgpu_nb_point(lib.hnswgpu_from_api(api)) == 700
        assert lib.file_dump_f32(api, 8, b"ffi_full") == 1
        lib.drop_hnsw_f32(api)
    finally:
        os.chdir(cwd)
    assert dumps_equal(tmp_path, "orc_full", "ffi_full")


def test_gpu_insert_without_a_device_leaves_the_index_unchanged(native, tmp_path):
    """hnswgpu_insert_gpu / GPU-assisted hnswgpu_build on a box without a usable device: refused BEFORE a point is accepted
    (BuildSearchBackend::check), so nb_point, the dump and a plain retry on the host are what they would have been."""
    import torch
    if torch.cuda.is_available():
        bad_device = torch.cuda.device_count() + 3   # a GPU box: an ordinal that does not exist
    else:
        bad_device = 0
    n, d, m = 2000, 12, 8
    X, Y = uniform(n, d, 31), uniform(500, d, 32)
    h = native.Hnsw(m, n + 1000, 16, 60, "DistL2")
    h.insert_serial(X)
    h.file_dump(tmp_path, "before")
    h.set_build_options(gpu_device=bad_device, gpu_window=0)
    for _ in range(2):  # a retry must not pile up half-inserted points either
        with pytest.raises(native.HnswError):
            h.parallel_insert(Y)
        assert h.get_nb_point() == n
    h.file_dump(tmp_path, "after")
    assert dumps_equal(tmp_path, "before", "after")
    # the same batch through the host builder: the index grows by exactly the batch
    h.set_build_options(gpu_device=-1, nthreads=1)
    h.parallel_insert(Y)
    assert h.get_nb_point() == n + 500
    assert sum(h.get_layer_nb_point(l) for l in range(16)) == n + 500
    # ... and a GPU-assisted build from scratch without a device fails without producing a handle
    g = native.Hnsw(m, n, 16, 60, "DistL2")
    g.set_build_options(gpu_device=bad_device, gpu_window=0)
    with pytest.raises(native.HnswError):
        g.parallel_insert(X)
