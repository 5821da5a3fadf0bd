"""Host-side mirror of the reference's interface for the search path, on top of the C ABI.

Same names and argument meaning as the Rust crate (so the parity tests read like the crate's tests):

    hnsw_rs::hnsw::Hnsw<f32, D>        -> Hnsw(max_nb_connection, max_elements, max_layer, ef_construction, "DistL2")
        .insert / .parallel_insert                      src/hnsw.rs:1069, :1224
        .search(data, knbn, ef)                         src/hnsw.rs:1597
        .parallel_search(datas, knbn, ef)               src/hnsw.rs:1612
    hnsw_rs::api::AnnT                 -> .search_neighbours / .parallel_search_neighbours / .file_dump
                                                        src/api.rs:13-38
    hnsw_rs::hnswio::HnswIo            -> HnswIo(directory, basename).load_hnsw("DistL2")
                                                        src/hnswio.rs:317, :431
    hnsw_rs::hnsw::Neighbour           -> Neighbour(d_id, distance, p_id=(layer, rank))   src/hnsw.rs:98-107

All searches run on the MI355X through libhnsw_mi355x.so; nothing here computes a distance.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _native as N

Neighbour = namedtuple("Neighbour", ["d_id", "distance", "p_id"])  # p_id = (layer, rank)


class HnswError(RuntimeError):
    """anyhow::Error analogue: carries the status code of the failing C-ABI call."""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def _check(rc):
    if rc != N.OK:
        raise HnswError(rc, N.last_error())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class BatchResult:
    """Flat form of Vec<Vec<Neighbour>>: row i holds counts[i] valid entries, ascending distance.
    status (filtered search only): 1 where the reference panics on the query (src/hnsw.rs:973), else 0."""

    def __init__(self, ids, dists, layers, ranks, counts, status=None):
        self.ids, self.dists, self.layers, self.ranks, self.counts = ids, dists, layers, ranks, counts
        self.status = status

    def to_neighbours(self):
        out = []
        for i in range(len(self.counts)):
            c = int(self.counts[i])
            out.append([Neighbour(int(self.ids[i, j]), float(self.dists[i, j]),
                                  (int(self.layers[i, j]), int(self.ranks[i, j]))) for j in range(c)])
        return out


class Hnsw:
    """Hnsw<f32, D>: owns a hnswgpu_index handle (flat host graph + its HBM replica)."""

    def __init__(self, max_nb_connection=16, max_elements=0, max_layer=16, ef_construction=200, dist="DistL2",
                 _handle=None):
        self._lib = N.lib()
        self._h = _handle
        self._pending = None
        if _handle is None:
            if max_nb_connection > 256:
                # the reference prints and calls process::exit(1) (src/hnsw.rs:784-787); we raise instead
                raise HnswError(N.ERR_ARG, "error max_nb_connection must be less equal than 256")
            self._params = N.BuildParams(max_nb_connection, ef_construction, max_layer, N.DIST[dist], 1.0, 0, 0, 0, 0, 0, 0, 0)
            self._dist = dist
        else:
            self._dist = N.DIST_NAME[self._lib.hnswgpu_dist(self._h)]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.hnswgpu_free_index(h)

    # ---- construction-time setters (src/hnsw.rs:845-905) --------------------------------
    def set_extend_candidates(self, flag):
        self._params.extend_candidates = int(bool(flag))

    def set_keeping_pruned(self, flag):
        self._params.keep_pruned = int(bool(flag))

    def modify_level_scale(self, scale_modification):
        self._params.level_scale_factor = min(1.0, max(0.2, float(scale_modification)))

    def set_build_options(self, nthreads=None, fast_arithmetic=None, gpu_device=None, gpu_window=None):
        """Extension: 1 thread = deterministic serial insert; fast_arithmetic = SIMD-order sums; gpu_device >= 0 =
        GPU-assisted construction (the insertions' searches on the device, window by window; gpu_window = 1: one point
        at a time, the serial insertion exactly)."""
        if nthreads is not None:
            self._params.nthreads = int(nthreads)
        if fast_arithmetic is not None:
            self._params.fast_arithmetic = int(bool(fast_arithmetic))
        if gpu_device is not None:
            self._params.gpu_assist = int(gpu_device >= 0)
            self._params.gpu_device = max(0, int(gpu_device))
        if gpu_window is not None:
            self._params.gpu_window = int(gpu_window)

    # ---- insertion (host construction; src/hnsw.rs:1069-1238) ---------------------------
    def parallel_insert(self, data, ids=None):
        """data: (n, d) f32 matrix; ids: origin ids (default: continue from the number of points).  The first call
        builds the index; later calls -- also on an index reloaded from a dump -- keep inserting into it."""
        data = np.ascontiguousarray(data, dtype=np.float32)
        if data.ndim != 2:
            raise HnswError(N.ERR_ARG, "data must be a (n, d) matrix")
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint64)
            idp = _p(ids)
        if self._h is not None:
            nthreads = self._params.nthreads if hasattr(self, "_params") else 0
            gpu = self._params.gpu_device if hasattr(self, "_params") and self._params.gpu_assist else -1
            if gpu >= 0:
                _check(self._lib.hnswgpu_insert_gpu(self._h, _p(data), data.shape[0], data.shape[1], idp, nthreads, gpu,
                                                    self._params.gpu_window))
            else:
                _check(self._lib.hnswgpu_insert(self._h, _p(data), data.shape[0], data.shape[1], idp, nthreads))
            return
        h = C.c_void_p()
        _check(self._lib.hnswgpu_build(_p(data), data.shape[0], data.shape[1], idp, C.byref(self._params),
                                       C.byref(h)))
        self._h = h.value

    def insert_serial(self, data, ids=None):
        """for (v, id) in data: hnsw.insert((v, id)) -- deterministic serial insertion."""
        if not hasattr(self, "_params"):  # reloaded index
            data = np.ascontiguousarray(data, dtype=np.float32)
            idp = _p(np.ascontiguousarray(ids, dtype=np.uint64)) if ids is not None else None
            _check(self._lib.hnswgpu_insert(self._h, _p(data), data.shape[0], data.shape[1], idp, 1))
            return
        saved = self._params.nthreads
        self._params.nthreads = 1
        try:
            self.parallel_insert(data, ids)
        finally:
            self._params.nthreads = saved

    # ---- getters -------------------------------------------------------------------------
    def get_nb_point(self):
        return self._lib.hnswgpu_nb_point(self._h) if self._h else 0

    def get_max_level_observed(self):
        return self._lib.hnswgpu_max_level_observed(self._h) if self._h else 0

    def get_layer_nb_point(self, layer):
        return self._lib.hnswgpu_layer_nb_point(self._h, layer) if self._h else 0

    def get_distance_name(self):
        return self._dist

    def get_description(self):
        d = N.Description()
        _check(self._lib.hnswgpu_get_description(self._h, C.byref(d)))
        return d

    def get_entry_point(self):
        o, l, r = C.c_uint64(), C.c_uint8(), C.c_int32()
        _check(self._lib.hnswgpu_entry_point(self._h, C.byref(o), C.byref(l), C.byref(r)))
        return o.value, (l.value, r.value)

    def get_neighbours(self, layer, rank, l, cap=512):
        ids = np.zeros(cap, np.uint64)
        layers = np.zeros(cap, np.uint8)
        ranks = np.zeros(cap, np.int32)
        dists = np.zeros(cap, np.float32)
        n = self._lib.hnswgpu_neighbours(self._h, layer, rank, l, cap, _p(ids), _p(layers), _p(ranks), _p(dists))
        if n < 0:
            raise HnswError(N.ERR_ARG, N.last_error())
        n = min(n, cap)
        return ids[:n], layers[:n], ranks[:n], dists[:n]

    # ---- device residency ------------------------------------------------------------------
    def upload(self, device=0):
        """Replicate vectors + neighbour lists into the HBM of HIP device `device`."""
        if self._h is None:
            raise HnswError(N.ERR_EMPTY, "index is empty")
        _check(self._lib.hnswgpu_upload(self._h, device))

    # ---- search -------------------------------------------------------------------------
    def parallel_search_flat(self, datas, knbn, ef, out=None):
        """Hnsw::parallel_search on a (nq, d) matrix; returns a BatchResult (flat arrays).  `out`: a BatchResult of an earlier
        call of the same shape whose arrays are written again (a caller in steady state: fresh arrays cost a page fault per
        4 KB while the answers are unpacked)."""
        datas = np.ascontiguousarray(datas, dtype=np.float32)
        if datas.ndim != 2:
            raise HnswError(N.ERR_ARG, "datas must be a (nq, d) matrix")
        nq, d = datas.shape
        if out is not None:
            ids, dists, layers, ranks, counts = out.ids, out.dists, out.layers, out.ranks, out.counts
            if ids.shape != (nq, knbn) or counts.shape != (nq,):
                raise HnswError(N.ERR_ARG, "out was made for another shape")
        else:
            ids = np.zeros((nq, knbn), np.uint64)
            dists = np.zeros((nq, knbn), np.float32)
            layers = np.zeros((nq, knbn), np.uint8)
            ranks = np.zeros((nq, knbn), np.int32)
            counts = np.zeros(nq, np.uint32)
        if self._h is None:  # empty index => empty answers (src/hnsw.rs:1498-1503)
            return BatchResult(ids, dists, layers, ranks, counts)
        _check(self._lib.hnswgpu_search_batch(self._h, _p(datas), nq, d, knbn, ef, _p(ids), _p(dists), _p(layers),
                                              _p(ranks), _p(counts)))
        return BatchResult(ids, dists, layers, ranks, counts)

    def parallel_search_sharded_flat(self, datas, knbn, ef, devices):
        """Hnsw::parallel_search with the batch sharded over the GPUs `devices` of this process (graph replicated,
        contiguous balanced blocks, answers gathered into one result; a device may be named more than once)."""
        datas = np.ascontiguousarray(datas, dtype=np.float32)
        nq, d = datas.shape
        ids = np.zeros((nq, knbn), np.uint64)
        dists = np.zeros((nq, knbn), np.float32)
        layers = np.zeros((nq, knbn), np.uint8)
        ranks = np.zeros((nq, knbn), np.int32)
        counts = np.zeros(nq, np.uint32)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        _check(self._lib.hnswgpu_search_batch_sharded(self._h, _p(dev), len(dev), _p(datas), nq, d, knbn, ef, _p(ids), _p(dists),
                                                      _p(layers), _p(ranks), _p(counts)))
        return BatchResult(ids, dists, layers, ranks, counts)

    def parallel_search_filter_flat(self, datas, knbn, ef, allowed_ids):
        """Hnsw::search_filter(data, knbn, ef, Some(&allowed_ids)) for every row of `datas` (src/hnsw.rs:1487-1580);
        allowed_ids = the SORTED id vector of `impl FilterT for Vec<usize>` (src/filter.rs:11-15).  Rows on which the
        reference panics come back with count 0 and status 1."""
        datas = np.ascontiguousarray(datas, dtype=np.float32)
        if datas.ndim != 2:
            raise HnswError(N.ERR_ARG, "datas must be a (nq, d) matrix")
        allowed = np.ascontiguousarray(allowed_ids, dtype=np.uint64)
        nq, d = datas.shape
        ids = np.zeros((nq, knbn), np.uint64)
        dists = np.zeros((nq, knbn), np.float32)
        layers = np.zeros((nq, knbn), np.uint8)
        ranks = np.zeros((nq, knbn), np.int32)
        counts = np.zeros(nq, np.uint32)
        status = np.zeros(nq, np.uint8)
        if self._h is None:
            return BatchResult(ids, dists, layers, ranks, counts, status)
        _check(self._lib.hnswgpu_search_batch_filtered(self._h, _p(datas), nq, d, knbn, ef, _p(allowed), len(allowed), _p(ids),
                                                       _p(dists), _p(layers), _p(ranks), _p(counts), _p(status)))
        return BatchResult(ids, dists, layers, ranks, counts, status)

    def search_filter(self, data, knbn, ef, allowed_ids):
        """Vec<Neighbour> of Hnsw::search_filter with a sorted id vector; raises where the reference panics."""
        data = np.ascontiguousarray(data, dtype=np.float32).reshape(1, -1)
        r = self.parallel_search_filter_flat(data, knbn, ef, allowed_ids)
        if r.status[0]:
            raise HnswError(N.ERR_REF_PANIC, "the reference panics on this query (return_points emptied by the filter, src/hnsw.rs:973)")
        return r.to_neighbours()[0]

    def parallel_search(self, datas, knbn, ef):
        """Vec<Vec<Neighbour>> in input order (src/hnsw.rs:1612-1635)."""
        return self.parallel_search_flat(datas, knbn, ef).to_neighbours()

    def search(self, data, knbn, ef):
        """Vec<Neighbour> (src/hnsw.rs:1597)."""
        data = np.ascontiguousarray(data, dtype=np.float32).reshape(1, -1)
        return self.parallel_search(data, knbn, ef)[0]

    # AnnT (src/api.rs:13-38)
    def search_neighbours(self, data, knbn, ef_s):
        return self.search(data, knbn, ef_s)

    def parallel_search_neighbours(self, data, knbn, ef_s):
        return self.parallel_search(data, knbn, ef_s)

    def file_dump(self, path, file_basename):
        """Writes <basename>.hnsw.graph / .hnsw.data into directory `path`; returns the basename."""
        if self._h is None:
            raise HnswError(N.ERR_EMPTY, "entry point not initialized")
        _check(self._lib.hnswgpu_file_dump(self._h, str(path).encode(), file_basename.encode()))
        return file_basename

    def last_search_kernel_ms(self):
        ms = C.c_double()
        _check(self._lib.hnswgpu_last_search_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def set_strict_ties(self, on=True):
        """Extension: replay tie-affected queries with a literal emulation of the reference's heaps."""
        _check(self._lib.hnswgpu_set_strict_ties(self._h, int(bool(on))))

    def set_arithmetic(self, arithmetic="scalar"):
        """Extension: "simd8" = distances of the following searches summed in the order of the crate's simdeez_f build."""
        _check(self._lib.hnswgpu_set_arithmetic(self._h, ARITH[arithmetic]))

    def last_tie_count(self):
        n = C.c_uint32()
        _check(self._lib.hnswgpu_last_tie_count(self._h, C.byref(n)))
        return n.value

    def last_kernel_ms(self):
        ms, n = C.c_double(), C.c_uint32()
        _check(self._lib.hnswgpu_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @property
    def handle(self):
        return self._h


class HnswIo:
    """HnswIo::new(directory, basename) (src/hnswio.rs:317)."""

    def __init__(self, directory, basename):
        self.dir, self.basename = str(directory), basename

    def get_basename(self):
        return self.basename

    def load_hnsw(self, dist=None):
        """load_hnsw::<f32, D>() (src/hnswio.rs:431-524).  dist=None accepts the dump's own distance."""
        h = C.c_void_p()
        code = N.DIST[dist] if dist is not None else -1
        rc = N.lib().hnswgpu_load_dump(self.dir.encode(), self.basename.encode(), code, C.byref(h))
        if rc != N.OK:
            raise HnswError(rc, N.last_error())
        return Hnsw(_handle=h.value)


class DataMap:
    """hnsw_rs::datamap::DataMap (src/datamap.rs): the vectors of a dump by DataId, memory-mapped, graph not loaded."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def from_hnswdump(directory, file_name):
        h = C.c_void_p()
        _check(N.lib().hnswgpu_datamap_open(str(directory).encode(), file_name.encode(), C.byref(h)))
        return DataMap(h.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().hnswgpu_datamap_close(h)

    def get_data(self, dataid):
        """Option<&[f32]>: a read-only view into the mapping, or None.  The view keeps this DataMap (and with it the mapping)
        alive: it may outlive every other reference to the object."""
        p = N.lib().hnswgpu_datamap_get_data(self._h, int(dataid))
        if not p:
            return None
        d = self.get_dimension()
        buf = (C.c_float * d).from_address(p)
        buf._datamap = self  # the ctypes object is the array's base: the mapping cannot be unmapped under the view
        a = np.ctypeslib.as_array(buf)
        a.flags.writeable = False
        return a

    def get_nb_data(self):
        return N.lib().hnswgpu_datamap_nb_data(self._h)

    def get_dimension(self):
        return N.lib().hnswgpu_datamap_dimension(self._h)

    def get_distname(self):
        return N.lib().hnswgpu_datamap_distname(self._h).decode()

    def get_data_typename(self):
        return N.lib().hnswgpu_datamap_typename(self._h).decode()

    def check_data_type(self, type_name):
        return type_name.rsplit("::", 1)[-1] == self.get_data_typename().rsplit("::", 1)[-1]

    def get_dataid_iter(self):
        n = self.get_nb_data()
        out = np.zeros(n, np.uint64)
        N.lib().hnswgpu_datamap_ids(self._h, _p(out), n)
        return out.tolist()


def load_description(graph_file_path):
    """load_description (src/hnswio.rs:937-1042) of a <basename>.hnsw.graph file."""
    d = N.Description()
    _check(N.lib().hnswgpu_load_description(str(graph_file_path).encode(), C.byref(d)))
    return d


ARITH = {"scalar": N.HEADER.constants["HNSWGPU_ARITH_SCALAR"], "simd8": N.HEADER.constants["HNSWGPU_ARITH_SIMD8"]}


def eval_distance_matrix(dist, queries, rows, batch, arithmetic="scalar"):
    """out[q][r] = Distance<f32>::eval(queries[q], rows[r]) on the device, by the search kernel's own distance routine with
    the rows taken in batches of `batch` (1..64) -- the lane-group branches the search takes for that many neighbours.
    arithmetic: "scalar" (the crate's default build) or "simd8" (the summation order of its simdeez_f build)."""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    r = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.zeros((q.shape[0], r.shape[0]), np.float32)
    _check(N.lib().hnswgpu_eval_distance_matrix_arith(N.DIST[dist], ARITH[arithmetic], _p(q), q.shape[0], _p(r), r.shape[0], q.shape[1],
                                                      batch, _p(out)))
    return out


def eval_distances(dist, a, b):
    """Distance<f32>::eval for the pairs (a[i], b[i]) on the device, by the search kernel's own distance routine."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.zeros(a.shape[0], np.float32)
    _check(N.lib().hnswgpu_eval_distances(N.DIST[dist], _p(a), _p(b), a.shape[0], a.shape[1], _p(out)))
    return out


def reload_env():
    """The HNSWGPU_* tuning / test hooks are read from the environment once per process; call this after changing them."""
    _check(N.lib().hnswgpu_reload_env())
