"""Round-4 GPU tests: BASELINE config 4 at its own size (1M x 128, 100 000 queries in 8 shards), and what the round added
to the search path (see the sections below)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu

from test_gpu_parity import assert_same  # noqa: E402
from test_gpu_round2 import _clustered  # noqa: E402


# ------------------------------------------------------------------------------------------------- config 4 at full size
@pytest.fixture(scope="module")
def sift1m(native, oracle, tmp_path_factory):
    """BASELINE config 2 / 4's index at its real size, built once for the module: 1M x 128 clustered, M=16, ef_c=200,
    GPU-assisted construction (the product's) -> hnswio dump -> product reload and oracle reload of the same files."""
    tmp = tmp_path_factory.mktemp("sift1m")
    n, d = 1_000_000, 128
    X = _clustered(n, d, 0x5EED0001)
    rng = np.random.default_rng(3)
    X[rng.choice(n, 2000, replace=False)] = X[rng.choice(n, 2000, replace=False)]  # exact duplicates under distinct ids
    hb = native.Hnsw(16, n, 16, 200, "DistL2")
    hb.set_build_options(nthreads=0, gpu_device=0, gpu_window=0)
    hb.parallel_insert(X)
    assert hb.get_nb_point() == n
    hb.file_dump(tmp, "sift1m")
    del hb
    h = native.HnswIo(tmp, "sift1m").load_hnsw("DistL2")
    o = oracle.OracleHnsw.load(tmp, "sift1m", "DistL2")
    stored = X[np.random.default_rng(4).choice(n, 500, replace=False)].copy()
    del X
    return h, o, stored


def test_config4_100k_queries_in_8_shards_on_1m_x_128(native, oracle, sift1m):
    """BASELINE config 4 itself (src/hnsw.rs:1612-1635 fanned out): 100 000 clustered queries against the 1M x 128
    index, split into 8 contiguous shards of 12 500, searched through hnswgpu_search_batch_sharded_device with one replica per
    device -- the shards go round-robin over every HIP device of the box (8 real GPUs on an 8-GPU node, eight concurrent
    shards on device 0 here) -- and every answer (ids, f32 distance bits, p_ids, counts) equals the oracle's.  The same
    100 000 queries in ONE call on one device give the same answers again."""
    import torch
    h, o, stored = sift1m
    lib = native.lib()
    k, ef, d, n_sh = 10, 64, 128, 8
    nq = 100_000
    Q = _clustered(nq, d, 0x5EED0004)
    Q[::200] = stored  # 500 queries ARE stored points (distance exactly 0, and the ties of their duplicates)
    ndev = lib.hnswgpu_device_count()
    assert ndev >= 1
    devs = [s % ndev for s in range(n_sh)]
    sizes = [nq // n_sh] * n_sh
    bounds = np.cumsum([0] + sizes)
    tdev = [torch.device("cuda", dv) for dv in devs]
    qs = [torch.from_numpy(Q[bounds[s]:bounds[s + 1]]).to(tdev[s]) for s in range(n_sh)]
    ids = [torch.zeros((sizes[s], k), dtype=torch.int64, device=tdev[s]) for s in range(n_sh)]
    dists = [torch.zeros((sizes[s], k), dtype=torch.float32, device=tdev[s]) for s in range(n_sh)]
    layers = [torch.zeros((sizes[s], k), dtype=torch.uint8, device=tdev[s]) for s in range(n_sh)]
    ranks = [torch.zeros((sizes[s], k), dtype=torch.int32, device=tdev[s]) for s in range(n_sh)]
    counts = [torch.zeros((sizes[s],), dtype=torch.int32, device=tdev[s]) for s in range(n_sh)]
    streams = [torch.cuda.Stream(tdev[s]) for s in range(n_sh)]
    for dv in set(devs):
        torch.cuda.synchronize(dv)

    def ptrs(ts):
        return (C.c_void_p * n_sh)(*[t.data_ptr() for t in ts])

    rc = lib.hnswgpu_search_batch_sharded_device(h.handle, (C.c_int * n_sh)(*devs), n_sh, ptrs(qs), (C.c_uint64 * n_sh)(*sizes), d, k, ef,
                                                 ptrs(ids), ptrs(dists), ptrs(layers), ptrs(ranks), ptrs(counts),
                                                 (C.c_void_p * n_sh)(*[s.cuda_stream for s in streams]))
    assert rc == 0, native._native.last_error()
    for dv in set(devs):
        torch.cuda.synchronize(dv)

    def cat(ts, dtype=None):
        a = np.concatenate([t.cpu().numpy() for t in ts])
        return a.astype(dtype) if dtype is not None else a

    got = oracle.SearchResult(cat(ids, np.uint64), cat(dists), cat(layers), cat(ranks), cat(counts, np.uint32))
    ref = o.parallel_search(Q, k, ef)
    assert_same(got, ref)
    print(f"config 4: 100 000 queries in 8 shards over {ndev} device(s) {sorted(set(devs))}: all answers == oracle")
    # the gather without a collective library (a Rust host driving the node's GPUs from one process): peer copies into device 0
    root = torch.device("cuda", 0)
    g_ids = torch.zeros((nq, k), dtype=torch.int64, device=root)
    g_d = torch.zeros((nq, k), dtype=torch.float32, device=root)
    g_l = torch.zeros((nq, k), dtype=torch.uint8, device=root)
    g_r = torch.zeros((nq, k), dtype=torch.int32, device=root)
    g_c = torch.zeros((nq,), dtype=torch.int32, device=root)
    torch.cuda.synchronize(0)
    rc = lib.hnswgpu_gather_sharded_answers((C.c_int * n_sh)(*devs), n_sh, (C.c_uint64 * n_sh)(*sizes), k, ptrs(ids), ptrs(dists), ptrs(layers),
                                            ptrs(ranks), ptrs(counts), 0, g_ids.data_ptr(), g_d.data_ptr(), g_l.data_ptr(), g_r.data_ptr(),
                                            g_c.data_ptr(), None)
    assert rc == 0, native._native.last_error()
    assert_same(oracle.SearchResult(g_ids.cpu().numpy().astype(np.uint64), g_d.cpu().numpy(), g_l.cpu().numpy(), g_r.cpu().numpy(),
                                    g_c.cpu().numpy().astype(np.uint32)), ref)
    # one call, one device, the whole batch (the large-batch end of hnswgpu_search_batch)
    assert_same(h.parallel_search_flat(Q, k, ef), ref)
    assert h.last_tie_count() > 0


def test_filtered_search_at_full_size(native, oracle, sift1m):
    """Hnsw::search_filter with a sorted id vector on the 1M x 128 index, 1 % and 30 % of the points allowed: clustered queries,
    stored points, and uniform queries that wander far from every cluster (thousands of candidates accepted per query: candidate
    heaps beyond their LDS part, runs of pushes across powers of two -- the case the round-4 soak caught) == the oracle's
    search_filter: ids, distance bits, p_ids, counts, panic flags (src/hnsw.rs:1487-1580)."""
    h, o, stored = sift1m
    h.upload(0)
    k, ef, n = 10, 64, 1_000_000
    Q = np.concatenate([_clustered(384, 128, 0x5EED0007), stored[:128], np.random.default_rng(9).random((256, 128), dtype=np.float32)])
    for pct, seed in ((1, 21), (30, 22)):
        allowed = np.sort(np.random.default_rng(seed).choice(n, n * pct // 100, replace=False)).astype(np.uint64)
        got = h.parallel_search_filter_flat(Q, k, ef, allowed)
        ref = o.parallel_search_filter(Q, k, ef, allowed)
        assert np.array_equal(got.status, ref.status)
        assert_same(got, ref)


# ------------------------------------------------------------------------------------------------- SIMD-order arithmetic (opt-in)
SIMD8_DIMS = list(range(1, 131)) + [136, 159, 160, 161, 200, 255, 256, 257, 300, 383, 384, 500, 511, 512, 640, 767, 768, 783, 784, 785, 800]


@pytest.mark.parametrize("dist", ["DistL2", "DistL1", "DistDot", "DistCosine"])
def test_simd8_distance_routine_sweep(native, oracle, dist):
    """hnswgpu_set_arithmetic(HNSWGPU_ARITH_SIMD8): the search's distance routine in the summation order of the crate's
    simdeez_f build (Cargo.toml:104-111: 8 vertical f32 accumulators over the full blocks of 8, horizontal sum, d % 8 tail;
    DistCosine with three such sums) equals the oracle's dist_simd8 BIT FOR BIT for every dimension 1..130 and 21 larger ones
    (every residue mod 8 and mod 32), rows taken in batches of 1, 31, 32, 33 and 64 (one and two rounds of the 2-lanes-per-row
    routine).  Tolerance: 0 ulp."""
    for d in SIMD8_DIMS:
        rng = np.random.default_rng(1000 + d)
        X = rng.random((70, d), dtype=np.float32) - np.float32(0.3 if dist != "DistCosine" else 0.0)
        Q = rng.random((3, d), dtype=np.float32)
        if dist == "DistDot":
            X /= np.linalg.norm(X, axis=1, keepdims=True)
            Q /= np.linalg.norm(Q, axis=1, keepdims=True)
        if dist == "DistCosine" and d > 2:
            X[3] = 0.0  # the zero-norm rule
        want = oracle.dist_matrix(dist, Q, X, simd8=True)
        for batch in ((1, 31, 32, 33, 64) if d in (1, 7, 8, 9, 25, 31, 32, 33, 128, 784) else (33,)):
            got = native.eval_distance_matrix(dist, Q, X, batch=batch, arithmetic="simd8")
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (dist, d, batch)
    # and it is a different arithmetic: somewhere the scalar order gives other bits
    X = np.random.default_rng(5).random((64, 128), dtype=np.float32)
    Q = np.random.default_rng(6).random((4, 128), dtype=np.float32)
    if dist == "DistDot":
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    if dist != "DistL1":
        assert not np.array_equal(native.eval_distance_matrix(dist, Q, X, batch=33, arithmetic="simd8").view(np.uint32),
                                  native.eval_distance_matrix(dist, Q, X, batch=33).view(np.uint32))


@pytest.mark.parametrize("dist,d,m,efc,k,ef,normalize", [("DistL2", 128, 16, 200, 10, 64, False), ("DistCosine", 25, 24, 200, 10, 128, False),
                                                       ("DistDot", 25, 24, 200, 10, 128, True), ("DistL1", 10, 16, 100, 10, 20, False),
                                                       ("DistL2", 29, 12, 100, 5, 200, False)])
def test_simd8_search_matches_the_oracle_in_simd_order(native, oracle, tmp_path, dist, d, m, efc, k, ef, normalize):
    """The whole search in SIMD-order arithmetic == the oracle searching the same dump with its distances in the same order
    (set_simd_order): ids, distance bits, p_ids, counts; strict ties included.  Switching back restores the scalar answers."""
    from conftest import normalized, uniform
    n, nq = 6000, 600
    X = normalized(n, d, 41) if normalize else uniform(n, d, 41)
    Q = normalized(nq, d, 42) if normalize else uniform(nq, d, 42)
    Q[:40] = X[:40]
    o = oracle.OracleHnsw(m, n, 16, efc, dist)
    o.insert_batch(X)
    o.file_dump(tmp_path, "s8")
    h = native.HnswIo(tmp_path, "s8").load_hnsw(dist)
    h.upload(0)
    scalar = o.parallel_search(Q, k, ef)
    assert_same(h.parallel_search_flat(Q, k, ef), scalar)
    h.set_arithmetic("simd8")
    o.set_simd_order(True)
    assert_same(h.parallel_search_flat(Q, k, ef), o.parallel_search(Q, k, ef))
    h.set_arithmetic("scalar")
    o.set_simd_order(False)
    assert_same(h.parallel_search_flat(Q, k, ef), scalar)


def test_simd8_full_size_config2(native, oracle, sift1m):
    """BASELINE config 2 in the arithmetic of the reference's published numbers: 1M x 128, 10 000 queries, SIMD-order distances
    on both sides (the device's opt-in mode, the oracle's set_simd_order) -> identical answers."""
    h, o, stored = sift1m
    Q = _clustered(10_000, 128, 0x5EED0002)
    Q[::20] = stored
    h.upload(0)
    h.set_arithmetic("simd8")
    o.set_simd_order(True)
    try:
        assert_same(h.parallel_search_flat(Q, 10, 64), o.parallel_search(Q, 10, 64))
    finally:
        h.set_arithmetic("scalar")
        o.set_simd_order(False)


# ------------------------------------------------------------------------------------------------- DistCosine on hostile exponents
@pytest.mark.parametrize("d", [1, 3, 5, 25, 29, 30, 45, 61, 62, 100, 125, 126])
def test_cosine_f64_chain_on_hostile_exponent_ranges(native, oracle, d):
    """DistCosine's three sums (anndists: f32 products widened to f64, added left to right) on rows where the ORDER of the
    additions decides the bits: components spread over 2^-40..2^40, subnormal components, products that underflow, values near
    2^61, exact cancellation, signed zeros, all-zero rows, exact powers of two -- in batches of 1 to 64 rows (4 and 2 lanes per
    row, one and two rounds), with the norm inside the row (d + 2 <= row_stride) or beside it.  Written while measuring a sum
    that leaves the chain when the exponent windows prove no addition can round (it lost to the chain on the
    device and was dropped); the inputs stay as the sharpest test of the chain's order.  Tolerance: 0 ulp."""
    rng = np.random.default_rng(7000 + d)
    n = 128
    X = (rng.random((n, d), dtype=np.float32) - np.float32(0.5))
    X[8:16] *= np.exp2(rng.integers(-40, 41, (8, d))).astype(np.float32)          # wide windows
    X[16:20] *= np.float32(1e-30)                                                  # tiny but normal: products underflow against tiny queries
    X[20:24] *= np.float32(2.0 ** 61)                                              # large: the overflow bound of the test, not crossed
    X[24, :: max(1, d // 3)] = np.float32(1e-42)                                   # subnormal components
    X[25] = 0.0
    X[26, ::2] = -0.0
    X[27] = np.float32(2.0) ** rng.integers(-3, 4, d)                              # exact powers of two
    X[64 + 5] *= np.exp2(rng.integers(-60, 61, d)).astype(np.float32)              # one failing row in an otherwise plain batch
    Q = (rng.random((12, d), dtype=np.float32) - np.float32(0.5))
    Q[1] = X[2]
    Q[2] = -X[3]                                                                    # cosine -1 ... +1, cancellation inside the sum
    Q[3] *= np.exp2(rng.integers(-30, 31, d)).astype(np.float32)
    Q[4] *= np.float32(1e-30)
    Q[5] *= np.float32(2.0 ** 61)
    Q[6, 0] = np.float32(1e-41)
    Q[7] = 0.0
    Q[8] = np.float32(2.0) ** rng.integers(-3, 4, d)
    if d > 1:
        Q[9, 1::2] = -Q[9, 1::2]
    want = oracle.dist_matrix("DistCosine", Q, X).view(np.uint32)
    for batch in (1, 16, 17, 32, 33, 64):
        got = native.eval_distance_matrix("DistCosine", Q, X, batch=batch).view(np.uint32)
        assert np.array_equal(got, want), (d, batch, np.argwhere(got != want)[:5])


def test_cosine_products_in_the_subnormal_range(native, oracle):
    """Query and rows scaled by 2^-70: every product lands in f32's subnormal range (or at zero); the device keeps f32
    subnormals like the host does, and the sums equal the oracle's."""
    rng = np.random.default_rng(7999)
    for d in (7, 25, 61, 125):
        X = (rng.random((40, d), dtype=np.float32) - np.float32(0.5)) * np.float32(2.0 ** -70)
        Q = (rng.random((4, d), dtype=np.float32) - np.float32(0.5)) * np.float32(2.0 ** -70)
        Q[3] *= np.float32(2.0 ** 70)
        want = oracle.dist_matrix("DistCosine", Q, X).view(np.uint32)
        for batch in (16, 33):
            got = native.eval_distance_matrix("DistCosine", Q, X, batch=batch).view(np.uint32)
            assert np.array_equal(got, want), (d, batch)


# ------------------------------------------------------------------------------------------------- the FFI's answers written in place
def _ffi_answers(lib, res, nq):
    out = []
    for i in range(nq):
        nb = res.contents.ptr[i]
        out.append([(nb.neighbours[j].id, nb.neighbours[j].d) for j in range(nb.nbgh)])
    return out


def test_ffi_answers_written_in_place_equal_the_unpacked_ones(native, oracle, tmp_path, knob, monkeypatch):
    """parallel_search_neighbours_f32 (src/libext.rs:205-254): the search kernels write ids, distances and counts straight into
    the Neighbour_api / Neighbourhood_api records of a page-locked slab (no unpacking pass).  The records equal the oracle's
    answers and those of the unpacking path (HNSWGPU_FFI_UNPACK=1: ordinary memory filled from the pinned arena); a freed slab
    comes back for the next call of the same shape and is laid out again for another shape; answers held by the caller stay
    valid while later calls run; ef > 1024 goes through the literal-heap kernel's writer; 300 answers never freed (the
    reference leaks them all) pass the page-locked limit and are served from ordinary memory."""
    from conftest import uniform
    lib = native.lib()
    n, d = 6000, 20
    X = uniform(n, d, 51)
    X[100:140] = X[200:240]  # exact ties
    o = oracle.OracleHnsw(12, n, 16, 80, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "ffi")
    monkeypatch.chdir(tmp_path)
    io = lib.get_hnswio(3, b"ffi")
    api = lib.load_hnswdump_f32_DistL2(io)
    assert api
    nq = 700
    Q = uniform(nq, d, 52)
    Q[:40] = X[100:140]
    ptrs = (C.c_void_p * nq)(*[Q[i].ctypes.data for i in range(nq)])

    def expect(k, ef, m=nq):
        ref = o.parallel_search(Q[:m], k, ef)
        return [[(int(ref.ids[i, j]), float(ref.dists[i, j])) for j in range(ref.counts[i])] for i in range(m)]

    want = expect(7, 40)
    res = lib.parallel_search_neighbours_f32(api, nq, d, ptrs, 7, 40)
    assert res and res.contents.len == nq
    assert _ffi_answers(lib, res, nq) == want
    first_addr = C.addressof(res.contents)
    # a second answer while the first is held: another slab, the first untouched
    res2 = lib.parallel_search_neighbours_f32(api, nq, d, ptrs, 7, 40)
    assert C.addressof(res2.contents) != first_addr
    assert _ffi_answers(lib, res2, nq) == want and _ffi_answers(lib, res, nq) == want
    lib.hnswgpu_free_neighbourhood_vec(res)
    lib.hnswgpu_free_neighbourhood_vec(res2)
    # freed slabs come back; another shape of the same size in bytes lays the records out again
    res = lib.parallel_search_neighbours_f32(api, nq, d, ptrs, 7, 40)
    assert C.addressof(res.contents) in (first_addr, C.addressof(res2.contents))
    assert _ffi_answers(lib, res, nq) == want
    lib.hnswgpu_free_neighbourhood_vec(res)
    res = lib.parallel_search_neighbours_f32(api, nq // 2, d, ptrs, 15, 40)   # 350 x (16 + 15 x 16) == 700 x (16 + 7 x 16)
    assert _ffi_answers(lib, res, nq // 2) == expect(15, 40, nq // 2)
    lib.hnswgpu_free_neighbourhood_vec(res)
    # the literal-heap kernel's writer (ef > 1024)
    res = lib.parallel_search_neighbours_f32(api, 64, d, ptrs, 12, 1100)
    assert _ffi_answers(lib, res, 64) == expect(12, 1100, 64)
    lib.hnswgpu_free_neighbourhood_vec(res)
    # the unpacking path gives the same records
    knob("HNSWGPU_FFI_UNPACK", "1")
    res = lib.parallel_search_neighbours_f32(api, nq, d, ptrs, 7, 40)
    assert _ffi_answers(lib, res, nq) == want
    lib.hnswgpu_free_neighbourhood_vec(res)
    knob("HNSWGPU_FFI_UNPACK", None)
    # answers that are never freed: beyond 256 MB of page-locked slabs the library hands out ordinary memory, still correct
    big_q = 4000
    Qb = np.tile(Q, (6, 1))[:big_q].copy()
    pb = (C.c_void_p * big_q)(*[Qb[i].ctypes.data for i in range(big_q)])
    refb = o.parallel_search(Qb[:50], 60, 64)
    held = []
    for it in range(80):  # 80 x 4000 x (16 + 60 x 16) bytes = 312 MB
        r = lib.parallel_search_neighbours_f32(api, big_q, d, pb, 60, 64)
        assert r
        held.append(r)
        if it in (0, 40, 79):
            got = _ffi_answers(lib, r, 50)
            assert got == [[(int(refb.ids[i, j]), float(refb.dists[i, j])) for j in range(refb.counts[i])] for i in range(50)]
    for r in held:
        lib.hnswgpu_free_neighbourhood_vec(r)
    lib.drop_hnsw_f32(api)
    lib.hnswgpu_free_hnswio(io)
