"""ctypes binding of the CPU oracle (oracle/liboracle_hnsw.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_hnsw.so")

DIST_L2, DIST_COSINE, DIST_DOT, DIST_L1, DIST_HELLINGER, DIST_JEFFREYS, DIST_JENSENSHANNON = 0, 1, 2, 3, 4, 5, 6
DIST_BY_NAME = {"DistL2": DIST_L2, "DistCosine": DIST_COSINE, "DistDot": DIST_DOT, "DistL1": DIST_L1,
                "DistHellinger": DIST_HELLINGER, "DistJeffreys": DIST_JEFFREYS, "DistJensenShannon": DIST_JENSENSHANNON}


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle_capi.cpp", "hnsw_oracle.hpp", "hnswio_oracle.hpp", "flat_baseline.hpp", "ref_logf.hpp")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_new.restype = C.c_void_p
        L.orc_new.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_modify_level_scale.argtypes = [C.c_void_p, C.c_double]
        L.orc_set_extend_candidates.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_keeping_pruned.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_nb_point.restype = C.c_size_t
        L.orc_get_nb_point.argtypes = [C.c_void_p]
        L.orc_get_layer_nb_point.restype = C.c_size_t
        L.orc_get_layer_nb_point.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_get_max_level_observed.argtypes = [C.c_void_p]
        L.orc_get_dimension.restype = C.c_size_t
        L.orc_get_dimension.argtypes = [C.c_void_p]
        L.orc_insert_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_parallel_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        L.orc_file_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_load.restype = C.c_void_p
        L.orc_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_dist.restype = C.c_float
        L.orc_dist.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_l2_normalize.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_flat_new.restype = C.c_void_p
        L.orc_flat_new.argtypes = [C.c_void_p]
        L.orc_flat_free.argtypes = [C.c_void_p]
        L.orc_flat_parallel_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ref_logf.restype = C.c_float
        L.orc_ref_logf.argtypes = [C.c_float]
        L.orc_ref_logf_mismatches.restype = C.c_uint64
        L.orc_ref_logf_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_dist_matrix.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_dist_matrix_simd8.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_levels.argtypes = [C.c_size_t, C.c_double, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_heap_exercise.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p,
                                        C.c_void_p]
        L.orc_search_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_parallel_search_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                                 C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_heap_retain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_heap_script.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _err():
    return lib().orc_last_error().decode()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class SearchResult:
    """Batched result: arrays padded to k; counts[i] entries are valid in row i."""

    def __init__(self, ids, dists, layers, ranks, counts):
        self.ids, self.dists, self.layers, self.ranks, self.counts = ids, dists, layers, ranks, counts


class OracleHnsw:
    """Mirror of hnsw_rs::Hnsw<f32, D> backed by the oracle restatement."""

    def __init__(self, max_nb_conn=None, max_elements=0, max_layer=16, ef_c=None, dist="DistL2", _handle=None):
        self.dist = dist
        if _handle is not None:
            self.h = _handle
        else:
            self.h = lib().orc_new(max_nb_conn, max_elements, max_layer, ef_c, DIST_BY_NAME[dist])
            if not self.h:
                raise RuntimeError(_err())

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:  # (module globals may be gone at interpreter shutdown)
            try:
                _lib.orc_free(self.h)
            except Exception:  # noqa: BLE001
                pass
            self.h = None

    def modify_level_scale(self, f):
        lib().orc_modify_level_scale(self.h, f)

    def set_extend_candidates(self, flag):
        lib().orc_set_extend_candidates(self.h, int(flag))

    def set_keeping_pruned(self, flag):
        lib().orc_set_keeping_pruned(self.h, int(flag))

    def get_nb_point(self):
        return lib().orc_get_nb_point(self.h)

    def get_layer_nb_point(self, l):
        return lib().orc_get_layer_nb_point(self.h, l)

    def get_max_level_observed(self):
        return lib().orc_get_max_level_observed(self.h)

    def insert_batch(self, data, ids=None):
        data = np.ascontiguousarray(data, dtype=np.float32)
        n, d = data.shape
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint64)
            idp = _p(ids)
        if lib().orc_insert_batch(self.h, _p(data), n, d, idp) != 0:
            raise RuntimeError(_err())

    def search(self, q, k, ef):
        q = np.ascontiguousarray(q, dtype=np.float32)
        ids = np.zeros(k, np.uint64)
        dists = np.zeros(k, np.float32)
        layers = np.zeros(k, np.uint8)
        ranks = np.zeros(k, np.int32)
        cnt = np.zeros(1, np.uint32)
        if lib().orc_search(self.h, _p(q), q.shape[0], k, ef, _p(ids), _p(dists), _p(layers), _p(ranks),
                            _p(cnt)) != 0:
            raise RuntimeError(_err())
        n = int(cnt[0])
        return ids[:n], dists[:n], layers[:n], ranks[:n]

    def parallel_search(self, queries, k, ef, nthreads=0, want_counters=False):
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        nq, d = queries.shape
        ids = np.zeros((nq, k), np.uint64)
        dists = np.zeros((nq, k), np.float32)
        layers = np.zeros((nq, k), np.uint8)
        ranks = np.zeros((nq, k), np.int32)
        counts = np.zeros(nq, np.uint32)
        counters = np.zeros(3, np.uint64)
        elapsed = C.c_double(0.0)
        rc = lib().orc_parallel_search(self.h, _p(queries), nq, d, k, ef, nthreads, _p(ids), _p(dists), _p(layers),
                                       _p(ranks), _p(counts), _p(counters) if want_counters else None,
                                       C.byref(elapsed))
        if rc != 0:
            raise RuntimeError(_err())
        res = SearchResult(ids, dists, layers, ranks, counts)
        res.elapsed_s = elapsed.value
        res.counters = dict(n_dist=int(counters[0]), n_expand=int(counters[1]), n_ids_read=int(counters[2]))
        return res

    def search_filter(self, q, k, ef, allowed_ids):
        """Hnsw::search_filter with a sorted Vec<usize> filter (src/hnsw.rs:1487, src/filter.rs:11-15)."""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
        allowed = np.ascontiguousarray(allowed_ids, dtype=np.uint64)
        ids = np.zeros(k, np.uint64)
        dists = np.zeros(k, np.float32)
        layers = np.zeros(k, np.uint8)
        ranks = np.zeros(k, np.int32)
        cnt = C.c_uint32(0)
        rc = lib().orc_search_filter(C.c_void_p(self.h), _p(q), len(q), k, ef, _p(allowed), len(allowed), _p(ids), _p(dists),
                                     _p(layers), _p(ranks), C.byref(cnt))
        if rc != 0:
            raise RuntimeError(_err())
        c = cnt.value
        return ids[:c], dists[:c], layers[:c], ranks[:c]

    def parallel_search_filter(self, queries, k, ef, allowed_ids, nthreads=0, want_counters=False):
        """search_filter for every row of `queries` on worker threads; .status[i] == 1 where the reference panics."""
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        allowed = np.ascontiguousarray(allowed_ids, dtype=np.uint64)
        nq, d = queries.shape
        ids = np.zeros((nq, k), np.uint64)
        dists = np.zeros((nq, k), np.float32)
        layers = np.zeros((nq, k), np.uint8)
        ranks = np.zeros((nq, k), np.int32)
        counts = np.zeros(nq, np.uint32)
        status = np.zeros(nq, np.uint8)
        counters = np.zeros(3, np.uint64)
        elapsed = C.c_double(0.0)
        rc = lib().orc_parallel_search_filter(C.c_void_p(self.h), _p(queries), nq, d, k, ef, _p(allowed), len(allowed), nthreads,
                                              _p(ids), _p(dists), _p(layers), _p(ranks), _p(counts), _p(status),
                                              _p(counters) if want_counters else None, C.byref(elapsed))
        if rc != 0:
            raise RuntimeError(_err())
        res = SearchResult(ids, dists, layers, ranks, counts)
        res.status = status
        res.elapsed_s = elapsed.value
        res.counters = counters if want_counters else None
        return res

    def flat_baseline(self):
        """The optimised flat-array CPU searcher built from this index (timing only: see oracle/flat_baseline.hpp)."""
        return FlatBaseline(self)

    @staticmethod
    def set_thread_pinning(on):
        """Timing-only: worker t of every parallel_search runs on the t-th logical CPU, NUMA node by NUMA node (oracle/pinning.hpp)."""
        lib().orc_set_thread_pinning(int(bool(on)))

    @staticmethod
    def pinning_cpu(t):
        lib().orc_pinning_cpu.restype = C.c_int
        return int(lib().orc_pinning_cpu(int(t)))

    def set_simd_order(self, on):
        """Timing-only: distances summed in the crate's SIMD (8-lane) order; results differ in the last bits."""
        lib().orc_set_simd_order(C.c_void_p(self.h), int(bool(on)))

    def file_dump(self, directory, basename):
        if lib().orc_file_dump(self.h, str(directory).encode(), basename.encode()) != 1:
            raise RuntimeError(_err())
        return basename

    @staticmethod
    def load(directory, basename, dist="DistL2"):
        h = lib().orc_load(str(directory).encode(), basename.encode(), DIST_BY_NAME[dist])
        if not h:
            raise RuntimeError(_err())
        return OracleHnsw(dist=dist, _handle=h)


class FlatBaseline:
    def __init__(self, orc):
        self.f = lib().orc_flat_new(C.c_void_p(orc.h))
        if not self.f:
            raise RuntimeError(_err())

    def __del__(self):
        if getattr(self, "f", None):
            lib().orc_flat_free(C.c_void_p(self.f))
            self.f = None

    def parallel_search(self, queries, k, ef, nthreads=0):
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        nq, d = queries.shape
        ids = np.zeros((nq, k), np.uint64)
        dists = np.zeros((nq, k), np.float32)
        counts = np.zeros(nq, np.uint32)
        elapsed = C.c_double(0.0)
        if lib().orc_flat_parallel_search(C.c_void_p(self.f), _p(queries), nq, d, k, ef, nthreads, _p(ids), _p(dists), _p(counts),
                                          C.byref(elapsed)) != 0:
            raise RuntimeError(_err())
        res = SearchResult(ids, dists, None, None, counts)
        res.elapsed_s = elapsed.value
        return res


def dist_eval(kind, a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().orc_dist(DIST_BY_NAME[kind], _p(a), _p(b), a.shape[0]))


def dist_matrix(kind, queries, rows, simd8=False):
    """out[q][r] = eval(queries[q], rows[r]) in the scalar order, or (simd8) in the crate's SIMD summation order (dist_simd8)."""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    r = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.zeros((q.shape[0], r.shape[0]), np.float32)
    fn = lib().orc_dist_matrix_simd8 if simd8 else lib().orc_dist_matrix
    fn(DIST_BY_NAME[kind], _p(q), q.shape[0], _p(r), r.shape[0], q.shape[1], _p(out))
    return out


def levels(max_nb_conn, n, scale_factor=1.0, maxlevel=16):
    out = np.zeros(n, np.uint8)
    lib().orc_levels(max_nb_conn, scale_factor, maxlevel, n, _p(out))
    return out


def heap_exercise(vals, tags, mode, npop=0):
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    tags = np.ascontiguousarray(tags, dtype=np.int32)
    n = len(vals)
    ov = np.zeros(n, np.float32)
    ot = np.zeros(n, np.int32)
    r = lib().orc_heap_exercise(_p(vals), _p(tags), n, mode, npop, _p(ov), _p(ot))
    if r < 0:
        raise RuntimeError(_err())
    return ov[:r], ot[:r]


def heap_script(vals, tags, is_pop):
    """Interleaved pushes / pops on the restated BinaryHeap; returns (popped_vals, popped_tags, sorted_vals, sorted_tags)."""
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    tags = np.ascontiguousarray(tags, dtype=np.int32)
    is_pop = np.ascontiguousarray(is_pop, dtype=np.uint8)
    n = len(vals)
    ov, ot = np.zeros(n, np.float32), np.zeros(n, np.int32)
    sv, st = np.zeros(n, np.float32), np.zeros(n, np.int32)
    npop = C.c_size_t(0)
    r = lib().orc_heap_script(_p(vals), _p(tags), _p(is_pop), n, _p(ov), _p(ot), C.byref(npop), _p(sv), _p(st))
    if r < 0:
        raise RuntimeError(_err())
    return ov[:npop.value], ot[:npop.value], sv[:r], st[:r]


def heap_retain(vals, tags, keep):
    """push all, BinaryHeap::retain(keep[push index]), into_sorted_vec -> (vals, tags)."""
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    tags = np.ascontiguousarray(tags, dtype=np.int32)
    keep = np.ascontiguousarray(keep, dtype=np.uint8)
    n = len(vals)
    ov, ot = np.zeros(n, np.float32), np.zeros(n, np.int32)
    r = lib().orc_heap_retain(_p(vals), _p(tags), _p(keep), n, _p(ov), _p(ot))
    if r < 0:
        raise RuntimeError(_err())
    return ov[:r], ot[:r]
