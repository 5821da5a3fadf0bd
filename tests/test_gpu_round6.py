"""Round-6 GPU tests: BASELINE config 5's shape beyond the Infinity Cache at full size against the oracle."""
import pytest

from test_gpu_round3 import _full_size

pytestmark = pytest.mark.gpu


def test_full_size_parity_mnist784_hbm_240k_x_784(native, oracle, tmp_path, knob):
    """bench.py --config mnist784_hbm: 240 000 x 784 (753 MB of vectors: four times BASELINE config 5, out of the 256 MiB Infinity
    Cache), M = 32 (64 ids per row), ef = 200 (4 result slots per lane): GPU-assisted build -> dump -> product and oracle reload the
    same files -> ids, f32 distance bits, p_ids and counts identical, also with every pop from the literal heap and with a visited
    table far too small (src/hnsw.rs:922-1064, :1487-1580)."""
    _full_size(native, oracle, tmp_path, knob, 240_000, 784, 32, 400, "DistL2", 10, 200, 1500, 600, 300)


def _device_call_with_stats(native, h, Q, k, ef):
    """hnswgpu_search_batch_device on buffers from the HIP runtime: (ids, distance bits, counts, the per-query counters)."""
    import ctypes as C
    import numpy as np
    lib = native.lib()
    hip = C.CDLL("libamdhip64.so")
    nq, d = Q.shape

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(max(nbytes, 4))) == 0
        return p

    def fetch(p, shape, dtype):
        a = np.zeros(shape, dtype)
        assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), p, C.c_size_t(a.nbytes), 2) == 0
        return a

    q = dmalloc(Q.nbytes)
    assert hip.hipMemcpy(q, Q.ctypes.data_as(C.c_void_p), C.c_size_t(Q.nbytes), 1) == 0
    outs = dict(ids=dmalloc(nq * k * 8), d=dmalloc(nq * k * 4), layer=dmalloc(nq * k), rank=dmalloc(nq * k * 4), cnt=dmalloc(nq * 4), stats=dmalloc(nq * 32))
    rc = lib.hnswgpu_search_batch_device(h.handle, q, nq, d, k, ef, outs["ids"], outs["d"], outs["layer"], outs["rank"], outs["cnt"], outs["stats"], None)
    assert rc == 0, native._native.last_error()
    got = (fetch(outs["ids"], (nq, k), np.uint64), fetch(outs["d"], (nq, k), np.uint32), fetch(outs["cnt"], (nq,), np.uint32),
           fetch(outs["stats"], (nq, 8), np.uint32))
    for p in [q] + list(outs.values()):
        hip.hipFree(p)
    return got


@pytest.mark.parametrize("n,d,m,dist,normalize,nq", [
    (6000, 128, 16, "DistL2", False, 701),      # config 2's shape; an odd count: the last wavefront holds one query
    (5000, 25, 16, "DistCosine", False, 512),   # the norm rides in the row's padding
    (5000, 32, 12, "DistCosine", False, 300),   # no room in the row: the separate norm array
    (4000, 10, 8, "DistL1", False, 257),
    (4000, 30, 16, "DistDot", True, 2),
    (4000, 30, 16, "DistDot", True, 1),
    (3000, 48, 24, "DistL2", False, 400),       # lists of up to 24 ids above the search layer: one query per wavefront either way
])
def test_descent_two_queries_per_wavefront(native, oracle, tmp_path, knob, n, d, m, dist, normalize, nq):
    """hnsw_descend_pair_kernel (two queries per wavefront where no list above the search layer holds more than 16 ids) against
    hnsw_descend_kernel (HNSWGPU_NO_PAIR_DESCENT) and the oracle: the same entry point of the search layer for every query -- seen
    through identical answers AND identical per-query counters (distances, lists, ids read; word 7: the descent's own share) --
    src/hnsw.rs:1506-1529."""
    import numpy as np
    from conftest import normalized, uniform
    from test_gpu_parity import build_pair
    X, o, h = build_pair(native, oracle, tmp_path, n, d, m, 100, dist, seed=n + d + m, normalize=normalize)
    Q = normalized(nq, d, 5) if normalize else uniform(nq, d, 5)
    k, ef = 10, 48
    ref = o.parallel_search(Q, k, ef)
    ids_p, d_p, cnt_p, st_p = _device_call_with_stats(native, h, Q, k, ef)
    knob("HNSWGPU_NO_PAIR_DESCENT", "1")
    ids_s, d_s, cnt_s, st_s = _device_call_with_stats(native, h, Q, k, ef)
    knob("HNSWGPU_NO_PAIR_DESCENT", None)
    assert np.array_equal(cnt_p, ref.counts.astype(np.uint32)) and np.array_equal(cnt_s, cnt_p)
    assert np.array_equal(ids_p, ref.ids.astype(np.uint64)) and np.array_equal(ids_s, ids_p)
    assert np.array_equal(d_p, ref.dists.view(np.uint32)) and np.array_equal(d_s, d_p)
    for c in (0, 1, 2, 7):
        assert np.array_equal(st_p[:, c], st_s[:, c]), f"counter {c} differs between the two descent kernels"
    assert (st_p[:, 7] >> 16).min() >= 1   # every query measured at least the distance to the entry point
